#!/bin/bash
# Round 3, GPU call H: how much foreign vector-ALU work fits into pass 1's idle issue slots (ballast workgroups inside k_produce_lab).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
{
  for r in 1 2; do for L in default w6 w5 w4 chain100 chain300; do
    if [ $L = default ]; then unset VRGDG_HIP_LIB; else export VRGDG_HIP_LIB=$PWD/tools/ab/lib_$L.so; fi
    echo "=== $L"; timeout 300 python tools/ab_pass_times.py chain4 64 5 2>&1 | tail -1
  done; done
} > $O/round.log 2>&1
cat $O/round.log
