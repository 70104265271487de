#!/bin/bash
# Tuning sweep of the device-statistics frame kernel: split threshold x loads in flight, at several batch sizes.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for CFG in "0 1 1" "0 2 2" "1000 1 1" "1000 2 2" "1000 4 4"; do
  set -- $CFG
  export VRG_TS_SPLIT_MAX=$1 VRG_TS_DEPTH_SPLIT=$2 VRG_TS_DEPTH_BATCH=$3
  for F in 1 4 16 64 128 256; do
    timeout 300 python - $F <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops
F = int(sys.argv[1]); dev = torch.device("cuda", 0)
lab = torch.rand((F, 2160, 3840, 3), device=dev) * 100 - 30
lab2 = torch.rand((4 * F, 1080, 1920, 3), device=dev) * 100 - 30
res = {}
for name, t in (("4k", lab), ("1080p_x4", lab2)):
    ts = []
    for it in range(6):
        a, b = ops.HipEvent(), ops.HipEvent()
        a.record(); ops.lab_stats_device(t, 1); b.record(); torch.cuda.synchronize()
        if it >= 2: ts.append(a.elapsed_ms(b))
    res[name] = round(statistics.median(ts), 3)
print("split_max", os.environ["VRG_TS_SPLIT_MAX"], "depth", os.environ["VRG_TS_DEPTH_SPLIT"], "frames", F, res)
PY
  done
done
} > gpurun_out/tstats_ab.log 2>&1
cat gpurun_out/tstats_ab.log
