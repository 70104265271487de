#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/check; mkdir -p $O
{
  echo "=== $(date) A/B"; timeout 600 python tools/ab_interleaved.py --libs new=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so,nogen=tools/ab/lib_nogen.so,nogen_nogath=tools/ab/lib_nogen_nogath.so --cases chain3,chain3_video --frames 64 --rounds 5 --json $O/ab_ablate_general.json 2>&1 | grep "^\[ab\]" | cut -c1-1200
  echo "=== $(date) done"
} > $O/check.log 2>&1
cat $O/check.log | cut -c1-1300
