#!/bin/bash
# Round 3, GPU call N: fused sharpen -> seeded grain: parity, timing, HBM traffic.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
{
  echo "=== $(date) pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "sharpen_then or sharpen_grain or seeded or enhancer" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -12
  echo "=== $(date) kernels"; timeout 600 python tools/ab_pass_times.py kernels 64 6 2>&1 | tail -1
  echo "=== $(date) kernels 256"; timeout 600 python tools/ab_pass_times.py kernels 256 4 2>&1 | tail -1
  echo "=== $(date) traffic"; bash tools/gpu_traffic.sh r03n 2>&1 | grep -A3 -E "k_sharpen_grain|k_stencil_flat|k_grain" | head -40
} > $O/round.log 2>&1
cat $O/round.log
