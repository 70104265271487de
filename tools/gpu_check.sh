#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/check; mkdir -p $O
{
  echo "=== $(date) A/B"; timeout 900 python tools/ab_interleaved.py --libs base=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so,rows120=tools/ab/lib_rows120.so,rows180=tools/ab/lib_rows180.so,rows360=tools/ab/lib_rows360.so --cases chain4 --frames 64 --rounds 7 --json $O/ab_apply_rows.json 2>&1 | grep "^\[ab\] chain4.apply\|bit" | cut -c1-1800
  echo "=== $(date) done"
} > $O/check.log 2>&1
cat $O/check.log | cut -c1-1900
