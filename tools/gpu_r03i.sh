#!/bin/bash
# Round 3, GPU call I: the staged pipeline (one launch per stage: pass 1 / statistics / pass 2 as workgroup roles).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
{
  echo "=== $(date) pytest (staged, pipelined, geometry, stats)"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "staged or pipelined or geometry or statistics or bench_verifies or fused_chain_with_colour" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -25
  for F in 96 256; do for S in 0 16 32 64; do
    echo "=== stage frames $S, frames $F"; VRGDG_CM_STAGE_FRAMES=$S timeout 300 python tools/ab_pass_times.py chain4 $F 5 2>&1 | tail -1
  done; done
  echo "=== $(date) bench"; timeout 900 python bench.py --no-host-fed 2>$O/bench.err | tee $O/bench.json | cut -c1-400
  echo "=== $(date) done"
} > $O/round.log 2>&1
cat $O/round.log
