#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
exec < /dev/null
timeout 1500 python tools/diff_libraries.py --base tools/ab/lib_r6_head.so --out gpurun_out/r06x/diff_head_vs_final.json 2>&1 | grep -E "^\[diff\]|Error|error|Traceback" | cut -c1-300
