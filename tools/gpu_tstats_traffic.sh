#!/bin/bash
# HBM read traffic of the device-statistics kernel in its one-workgroup-per-frame form (128 x 4K frames), and kernel time under the trace.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/ts_drv.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops
dev = torch.device("cuda", 0)
lab = torch.empty((128, 2160, 3840, 3), device=dev)
for i in range(0, 128, 16):
    lab[i:i + 16] = torch.rand((16, 2160, 3840, 3), device=dev) * 100 - 30
for _ in range(3):
    ops.lab_stats_device(lab, 1)
torch.cuda.synchronize()
PY
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ts_traffic/$C -o p -- python /tmp/ts_drv.py > $GRAFT_REPO_ROOT/gpurun_out/ts_traffic_$C.log 2>&1)
done
python - <<'PY' | tee gpurun_out/ts_traffic_summary.txt
import csv, glob
px = 128 * 2160 * 3840
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/ts_traffic/{c}/**/*counter_collection.csv", recursive=True):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_tstats_frame" in r["Kernel_Name"] and r["Counter_Name"] == c]
        if vals:
            kb = sum(vals) / len(vals)
            print(f"k_tstats_frame<false,1>, 128 x 4K frames, {c}: {kb:.0f} KB per launch = {kb * 1024 / px:.2f} B/px raw"
                  + (f" -> x2 (gfx950 FETCH_SIZE calibration, DESIGN.md section 5) = {kb * 2048 / px:.2f} B/px" if c == "FETCH_SIZE" else ""))
    for f in glob.glob(f"gpurun_out/ts_traffic/{c}/**/*kernel_trace.csv", recursive=True):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "k_tstats_frame" in r["Kernel_Name"]]
        if d and c == "FETCH_SIZE":
            print(f"kernel time under the trace: {min(d):.3f} .. {max(d):.3f} ms -> {px * 12 / min(d) / 1e9:.0f} GB/s algorithmic at best")
PY
