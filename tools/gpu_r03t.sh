#!/bin/bash
# Round 3, GPU call T: colour match alone (BASELINE configs[3]): prefetching Lab pass and 4-pixel apply, A/B.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03t; mkdir -p $O
{
  echo "=== $(date) pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "colour or color or colormatch or lab or stats or chain" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -12
  for rep in 1 2; do for lib in default oldcm; do
    echo "=== $(date) bench colormatch_4k $lib"
    if [ $lib = default ]; then timeout 600 python bench.py --workload colormatch_4k --frames 256 --no-cpu-baseline --no-host-fed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['roofline']['passes_ms'], d['fast_variant']['value'])"
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 600 python bench.py --workload colormatch_4k --frames 256 --no-cpu-baseline --no-host-fed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['roofline']['passes_ms'], d['fast_variant']['value'])"; fi
  done; done
} > $O/round.log 2>&1
cat $O/round.log
