#!/bin/bash
# Round 3, GPU call F: statistics reductions overlapped with pass 1 (high-priority side stream) -- sweep and bench.   bash tools/gpu_r03f.sh <tag>
TAG=${1:-f}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03${TAG}; mkdir -p $O
{
  echo "=== $(date) pytest (pipelined)"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "pipelined or geometry or bench_verifies" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -25
  for F in 64 128 256; do for P in 1 2 4 8 16; do
    echo "=== stats pieces $P frames $F"; VRGDG_CM_STATS_PIECES=$P timeout 300 python tools/ab_pass_times.py chain4 $F 5 2>&1 | tail -1
  done; done
  echo "=== $(date) bench"; timeout 900 python bench.py --no-host-fed 2>$O/bench.err | tee $O/bench.json | cut -c1-400
  echo "=== $(date) bench colormatch_4k"; timeout 900 python bench.py --workload colormatch_4k --no-cpu-baseline --no-host-fed 2>>$O/bench.err | tee $O/bench_cm.json | cut -c1-300
  echo "=== $(date) frames table"; timeout 1200 python tools/frames_table.py --out $O/frames_table.json --frames 32,64,128,256 2>&1 | grep "\[frames\]" | cut -c1-300
  echo "=== $(date) done"
} > $O/round.log 2>&1
cat $O/round.log
