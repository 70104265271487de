#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06r; mkdir -p $O
exec < /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q  > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
L=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
timeout 1200 python tools/ab_interleaved.py --libs head=tools/ab/lib_r6_head.so,c2=tools/ab/lib_r6_c2.so,new=$L --cases chain4,chain4_video,colormatch --frames 64 --rounds 7 --json $O/ab.json 2>&1 | grep "^\[ab\]" > $O/ab.log
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","r06r","ab.json")))
for k,v in d["metrics"].items(): print(k,{n:(r["median_ms"],r["spread_pct"],r.get("vs_head_pct")) for n,r in v.items()})
print(d["bit_identical"])
PY
