#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06f; mkdir -p $O
exec < /dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "march or fused_chain or chain3 or headline or randomized_chain or nan" 2>&1 | grep -v amdgpu.ids | tail -30 > $O/pytest_march.log
tail -8 $O/pytest_march.log
L=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so
timeout 600 python tools/ab_interleaved.py --libs base=tools/ab/lib_r6_prefast.so,new=$L --cases grain_sharpen,chain3,chain3_video,grain_lut --frames 64 --rounds 7 --json $O/ab_no_nan_rows.json 2>&1 | grep "^\[ab\]" > $O/ab_no_nan_rows.log
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","r06f","ab_no_nan_rows.json")))
for k,v in d["metrics"].items(): print(k,{n:(r["median_ms"],r.get("gpix_s"),r.get("verdict")) for n,r in v.items()})
print(d["bit_identical"])
PY
timeout 400 python tools/probe_valu_classes.py --waves 2 --ms 6 --json $O/valu_classes_all.json 2>&1 | grep -v amdgpu.ids > $O/valu_classes_all.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clkp -o p -- python $GRAFT_REPO_ROOT/tools/probe_valu_classes.py --ms 4 --waves 2 > $O/prof_classes.log 2>&1 )
python - <<'PY' > $O/clock_probe_classes.txt 2>&1
import csv, glob, os, re
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06f", "clkp")
dur = {}
for f in glob.glob(os.path.join(O, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur and "valu_rate" in r["Kernel_Name"]:
            n, d = dur[r["Dispatch_Id"]]
            if d > 2e6:
                m = re.search(r"valu_rate<(\d+)>", n)
                print(m.group(1) if m else n[:40], "dur_us", round(d / 1e3, 1), "clock_GHz", round(float(r["Counter_Value"]) / 8 / d, 3))
PY
rm -rf $O/clkp
cat $O/valu_classes_all.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['mode'], r['instr'], r['tera_lane_instr_s'], [v for k,v in r.items() if k.startswith('simd_cycles')][0])
"
