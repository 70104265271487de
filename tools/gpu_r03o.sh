#!/bin/bash
# Round 3, GPU call O: fused sharpen -> seeded grain, pipelined run loop A/B + issue counters.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
{
  echo "=== $(date) pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "sharpen_then or sharpen_grain or seeded or enhancer" 2>&1 | grep -E "passed|failed|error|^FAILED|^E  " | tail -12
  for lib in default sgnopipe default sgnopipe; do
    echo "=== $(date) kernels $lib"
    if [ $lib = default ]; then timeout 300 python tools/ab_pass_times.py kernels 128 6 2>&1 | tail -1
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 300 python tools/ab_pass_times.py kernels 128 6 2>&1 | tail -1; fi
  done
  echo "=== $(date) issue"; bash tools/gpu_issue.sh r03o 2>&1 | tail -40
} > $O/round.log 2>&1
cat $O/round.log
