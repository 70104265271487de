#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
exec < /dev/null
timeout 400 python tools/bench_flat_march.py --rounds 7 --cubes 33,25 --sizes 16x1080x1920,32x1080x1920,64x1080x1920,8x2160x3840,16x2160x3840,32x2160x3840,24x720x1280,96x720x1280 --json $O/bench_flat_march_sizes.json 2>&1 | grep "^\[flat\]" | cut -c1-330
