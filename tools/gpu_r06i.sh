#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06i; mkdir -p $O
exec < /dev/null
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "deferred or lazy or adjacent or inference or stream or node" 2>&1 | grep -v amdgpu.ids | tail -6
timeout 300 python tools/host_fed.py --frames 16 --out $O/host_fed_nodes.json 2>&1 | grep "^\[host\]" | cut -c1-200 | tail -4
L=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
timeout 600 python tools/ab_interleaved.py --libs base=tools/ab/lib_r6_base.so,keysp=tools/ab/lib_r6_keysp.so,keysm=tools/ab/lib_r6_keysm.so --cases chain4,chain3,chain3_video --frames 64 --rounds 5 --json $O/ab_keys.json 2>&1 | grep "^\[ab\]" > $O/ab_keys.log
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","r06i","ab_keys.json")))
for k,v in d["metrics"].items(): print(k,{n:(r["median_ms"],r.get("verdict")) for n,r in v.items()})
print(d["bit_identical"])
PY
