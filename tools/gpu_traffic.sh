#!/bin/bash
# HBM traffic of the headline kernels from PMC counters (separate passes; kernel-trace only).  bash tools/gpu_traffic.sh <tag>
TAG=${1:-t}
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_${TAG}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o p -- python $GRAFT_REPO_ROOT/tools/prof_driver.py traffic > $OUT/$C.log 2>&1
done
cd $OUT; python - <<'PY'
import csv, glob, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "vrg" not in n: continue
            res.setdefault(n[:90], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
px = 16 * 2160 * 3840
out = {}
for k, d in res.items():
    out[k] = {c: {"per_launch": sum(v) / len(v), "launches": len(v), "per_pixel": sum(v) / len(v) / px} for c, v in d.items()}
json.dump({"pixels_per_launch": px, "kernels": out}, open("traffic.json", "w"), indent=1)
for k, d in out.items():
    print(k[:80]); [print("    ", c, v) for c, v in d.items()]
PY
