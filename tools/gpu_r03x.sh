#!/bin/bash
# Round 3, GPU call X: apply march at 6 waves per SIMD (80 VGPRs, 5 dwords spilled) against 5 (88 VGPRs).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
{
  for rep in 1 2 3; do for lib in default amw6; do
    echo "=== $(date) A/B $lib"
    if [ $lib = default ]; then timeout 300 python tools/ab_pass_times.py chain4 64 6 2>&1 | tail -1
    else VRGDG_HIP_LIB=tools/ab/lib_$lib.so timeout 300 python tools/ab_pass_times.py chain4 64 6 2>&1 | tail -1; fi
  done; done
} > $O/round.log 2>&1
cat $O/round.log
