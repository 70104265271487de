#!/bin/bash
# Round 3, GPU call L: exhaustive Welford-division sweep, new tests, bench with the thread-probed CPU baseline.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03l; mkdir -p $O
{
  echo "=== $(date) welford division sweep"; timeout 600 python tools/welford_division_sweep.py --out $O/welford_division_sweep.json 2>&1 | tail -1
  echo "=== $(date) pytest (new)"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "welford or first_use or statistics or reentrant" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -12
  echo "=== $(date) bench"; timeout 900 python bench.py --no-host-fed 2>$O/bench.err | tee $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['cpu_baseline'], d['roofline']['issue'].get('weighted'))"
} > $O/round.log 2>&1
cat $O/round.log
