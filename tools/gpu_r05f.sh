#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
exec < /dev/null
for S in "2160x3840 8" "2158x3838 8" "768x1366 64" "480x854 128" "1080x1920 32" "1080x1918 32"; do set -- $S
  timeout 120 python tools/bench_u8_enhancer.py --size $1 --frames $2 --rounds 5 --json $O/u8_$1.json 2>&1 | grep "^\[u8\]" | grep -v "host-fed" | cut -c1-260
done
