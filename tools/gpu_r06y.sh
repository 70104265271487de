#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06y; mkdir -p $O
exec < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q -k "pow or ziv or colour or color or cm_ or lab_ or chain or fused or apply or march or graph or nodes or special or nan or selfcheck or toolchain" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 600 python tools/diff_libraries.py --base tools/ab/lib_r6_head.so --out $O/diff.json 2>&1 | grep -E "^\[diff\] total|DIFFERENT|Error|Traceback" | cut -c1-200
L=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
timeout 1200 python tools/ab_interleaved.py --libs head=tools/ab/lib_r6_head.so,prev=tools/ab/lib_r6_prev.so,new=$L --cases chain4,chain4_video,colormatch --frames 64 --rounds 9 --json $O/ab.json 2>&1 | grep "^\[ab\]" > $O/ab.log
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","r06y","ab.json")))
for k,v in d["metrics"].items(): print(k,{n:(r["median_ms"],r["spread_pct"],r.get("vs_head_pct")) for n,r in v.items()})
print(d["bit_identical"])
PY
