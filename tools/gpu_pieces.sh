#!/bin/bash
# How many pieces should the device-statistics flow cut a batch into?  (pass 1 of piece i+1 next to the reductions of piece i)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for F in 64 256; do
  for P in 1 2 4 8 16; do
    VRGDG_CM_PIECES=$P timeout 300 python tools/ab_pass_times.py chain4 $F 5 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} > gpurun_out/pieces.log 2>&1
cat gpurun_out/pieces.log
