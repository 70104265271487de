#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06l; mkdir -p $O
exec < /dev/null
L=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
LIBS="base=$L"
for n in occ2 occ2r r2 s3 s3r; do LIBS="$LIBS,$n=tools/ab/lib_r6_$n.so"; done
timeout 900 python tools/ab_interleaved.py --libs $LIBS --cases chain3,chain3_video,grain_lut,grain_sharpen --frames 64 --rounds 5 --json $O/ab_slots_rotate.json 2>&1 | grep "^\[ab\]" > $O/ab.log
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","r06l","ab_slots_rotate.json")))
for k,v in d["metrics"].items(): print(k,{n:(r["median_ms"],r.get("gpix_s")) for n,r in v.items()})
print(d["bit_identical"])
PY
