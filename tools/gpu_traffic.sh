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
# bench.py summary: HBM bytes per pixel of the two passes of the headline chain.  FETCH_SIZE/WRITE_SIZE are in KB;
# WRITE_SIZE calibrates to 12.00 B/px on every 12-B/px writer; FETCH_SIZE reads half of the known 12 B/px of k_lut3d
# (the gfx950 under-count of MI355X_MICROARCH.md), hence the factor 2.
def bpp(match):
    f = w = 0.0
    for k, d in out.items():
        if match(k):
            f += d.get("FETCH_SIZE", {}).get("per_pixel", 0.0) * 1024 * 2.0
            w += d.get("WRITE_SIZE", {}).get("per_pixel", 0.0) * 1024
    return {"read": round(f, 2), "written": round(w, 2), "total": round(f + w, 2)}
cal = out.get(next((k for k in out if "k_lut3d" in k), ""), {})
summary = {"fetch_correction": 2.0, "calibration_k_lut3d_read_bytes_per_px_raw": cal.get("FETCH_SIZE", {}).get("per_pixel", 0) * 1024,
           "stats": bpp(lambda k: "k_produce_lab<3" in k or "k_lab_partials<3" in k), "apply": bpp(lambda k: "k_apply_march<20" in k or "k_chain_tile<20" in k),
           "tstats": bpp(lambda k: "k_tstats_frame" in k or "k_tstats_rows<" in k),
           "chain3_apply": bpp(lambda k: "k_chain_march<3" in k or "k_chain_tile<3," in k or "k_chain_tile<3>" in k)}
json.dump({"pixels_per_launch": px, "kernels": out, "summary": summary}, open("traffic.json", "w"), indent=1)
print(summary)
for k, d in out.items():
    print(k[:80]); [print("    ", c, v) for c, v in d.items()]
PY
