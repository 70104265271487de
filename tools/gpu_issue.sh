#!/bin/bash
# VALU instruction counts per kernel (one PMC pass, kernel-trace only) + the issue-rate probe as calibration.
# bash tools/gpu_issue.sh <tag>
TAG=${1:-issue}
OUT=$GRAFT_REPO_ROOT/gpurun_out/issue_${TAG}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/p -o p -- python $GRAFT_REPO_ROOT/tools/prof_driver.py issue > $OUT/run.log 2>&1
cd $OUT; python - <<'PY'
import csv, glob, collections, json
rows = collections.OrderedDict()
for f in glob.glob('p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '')
        if 'vrg' not in name: continue
        key = (r.get('Dispatch_Id'), name[:90])
        rows.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
PX = 16 * 2160 * 3840
out = []
for (disp, name), d in rows.items():
    v = d.get('SQ_INSTS_VALU', 0)
    rec = {"dispatch": int(disp), "kernel": name, **d}
    if 'k_dbg_valu_rate' in name:
        rec["expected_wave_instr"] = 2048 * 4 * 512 * 64
        rec["counter_over_expected"] = v / rec["expected_wave_instr"]
    else:
        rec["valu_lane_instr_per_px"] = v * 64 / PX
    out.append(rec)
json.dump(out, open('summary.json', 'w'), indent=1)
for r in out:
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})
PY
