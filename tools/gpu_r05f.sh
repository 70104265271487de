#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
exec < /dev/null
P=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
timeout 300 python tools/ab_interleaved.py --libs base=$P,apnt2=tools/ab/lib_apnt2.so --cases chain4 --frames 64 --rounds 9 --json $O/ab_apply_nt_loads_stores.json 2>&1 | grep "^\[ab\]" | cut -c1-700
