#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06n; mkdir -p $O
exec < /dev/null
for d in uniform video; do timeout 600 python tools/probe_ziv_fallback_on_frames.py --dist $d --frames 4 --out $O/ziv_fallback_$d.json 2>&1 | grep "^\[ziv\]"; done
