#!/bin/bash
# round 6: Ziv route without its domain test where the caller guarantees the domain (pass 1), exp core with integer scaling: parity, then A/B against tools/ab/lib_r6_head.so
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06p; mkdir -p $O
exec < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q -k "pow or ziv or sigma_division or colour or color or cm_ or lab_ or chain or fused or selfcheck or toolchain" > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-300
L=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
timeout 900 python tools/ab_interleaved.py --libs head=tools/ab/lib_r6_head.so,new=$L --cases chain4,chain4_video,colormatch --frames 64 --rounds 5 --json $O/ab.json 2>&1 | grep "^\[ab\]" > $O/ab.log
cut -c1-330 $O/ab.log
