#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
exec < /dev/null
P=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
{
  echo "=== $(date) A/B round-4 march, product (nt loads + stores in the steady rows), general rows nt as well"
  timeout 300 python tools/ab_interleaved.py --libs r05=$P,r04m=tools/ab/lib_r04m.so,gennt=tools/ab/lib_gennt.so --cases chain3,chain3_video,grain_sharpen --frames 64 --rounds 9 --json $O/ab_r04_r05_march.json 2>&1 | grep "^\[ab\]" | cut -c1-1000
  echo "=== $(date) done"
} > $O/run.log 2>&1
cat $O/run.log
