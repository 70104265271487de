#!/bin/bash
# Round 3, GPU call D: apply march (pass 2) vs LDS tile.   bash tools/gpu_r03d.sh <tag>
TAG=${1:-d}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03${TAG}; mkdir -p $O
{
  echo "=== $(date) pytest (chains, colour match, geometry)"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "chain or colour or color or geometry or pipelined or bench or lanes or reentrant or surface" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -25
  for r in 1 2; do for V in 0 1; do for F in 32 256; do
    echo "=== variant $V frames $F"; VRGDG_VARIANT=$V timeout 300 python tools/ab_pass_times.py chain4 $F 5 2>&1 | tail -1
  done; done; done
  echo "=== $(date) bench"; timeout 900 python bench.py --no-host-fed 2>$O/bench.err | tee $O/bench.json | cut -c1-400
  echo "=== $(date) bench colormatch_4k"; timeout 900 python bench.py --workload colormatch_4k --no-cpu-baseline --no-host-fed 2>>$O/bench.err | tee $O/bench_cm.json | cut -c1-300
  echo "=== $(date) done"
} > $O/round.log 2>&1
cat $O/round.log
