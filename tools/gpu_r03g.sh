#!/bin/bash
# Round 3, GPU call G: persistent high-priority pass 2 next to pass 1 (experiment) + bench with the defaults.   bash tools/gpu_r03g.sh <tag>
TAG=${1:-g}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03${TAG}; mkdir -p $O
{
  echo "=== baseline"; timeout 300 python tools/ab_pass_times.py chain4 256 4 2>&1 | tail -1
  for P in 2 4 8; do for C in 1 2 3; do
    echo "=== p2 overlap: ranges $P, persistent workgroups per CU $C"; VRGDG_CM_STATS_PIECES=$P VRGDG_CM_P2_OVERLAP=$C timeout 300 python tools/ab_pass_times.py chain4 256 4 2>&1 | tail -1
  done; done
  echo "=== $(date) pytest (pipelined, persistent walk)"; VRGDG_APPLY_PERSISTENT=2 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "pipelined or fused_chain_with_colour or geometry" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -25
  echo "=== $(date) bench"; timeout 900 python bench.py --no-host-fed 2>$O/bench.err | tee $O/bench.json | cut -c1-400
  echo "=== $(date) done"
} > $O/round.log 2>&1
cat $O/round.log
