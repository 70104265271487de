#!/bin/bash
# round 6: what do the Ziv fallback branches (the ocml transcription, taken by a whole wave when one lane's rounding test fails) cost the headline's passes?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06m; mkdir -p $O
exec < /dev/null
L=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
timeout 900 python tools/ab_interleaved.py --libs base=$L,nofb=tools/ab/lib_r6_nofb.so --cases chain4,chain4_video --frames 64 --rounds 5 --json $O/ab_ziv_no_fallback.json 2>&1 | grep "^\[ab\]" > $O/ab.log
tail -40 $O/ab.log
