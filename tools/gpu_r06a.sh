#!/bin/bash
# round 6, batch A: occupancy curve of the shipped march (1 / 2 / 3 workgroups per CU), 1-wave workgroups, no inter-sibling fences, persistent waves
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
L=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
LIBS="base=$L"
for n in same wg1 occ2 occ1 nofence persist3 wg1nf; do LIBS="$LIBS,$n=tools/ab/lib_r6_$n.so"; done
timeout 1200 python tools/ab_interleaved.py --libs $LIBS --cases chain3,chain3_video,grain_sharpen --frames 64 --rounds 5 --json gpurun_out/r06_ab_a.json > gpurun_out/r06_ab_a.log 2>&1
grep -v amdgpu.ids gpurun_out/r06_ab_a.log | tail -30
