#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab; mkdir -p $O
exec < /dev/null
L=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
timeout 1500 python tools/ab_interleaved.py --libs head=tools/ab/lib_r6_head.so,prev=tools/ab/lib_r6_prev.so,new=$L --cases chain4,chain4_video --frames 64 --rounds 13 --json $O/ab.json 2>&1 | grep "^\[ab\]" > $O/ab.log
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","r06ab","ab.json")))
for k,v in d["metrics"].items(): print(k,{n:(r["median_ms"],r["min_ms"],r["spread_pct"],r.get("vs_head_pct")) for n,r in v.items()})
PY
