"""The PMC passes bench.py runs live (bench.live_traffic: three `rocprofv3 --kernel-trace --pmc` runs of tools/prof_driver.py traffic), as
a committed summary: python tools/collect_bench_pmc.py [gpurun_out/pmc_bench_kernels.json] -> copy to profiles/rNN_pmc_bench_kernels.json.
bench.py falls back to that file for its legs' HBM-traffic / VALU-busy figures when the live passes are switched off or fail."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_bench_kernels.json")
res = bench.live_traffic(timeout_s=300)
import torch
doc = {"device": torch.cuda.get_device_name(0) if torch.cuda.is_available() else None,
       "command": "rocprofv3 --kernel-trace --pmc <set> -- python tools/prof_driver.py traffic   (sets: FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 "
                  "SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY; 16 x 4K frames, uniform pixels)",
       "units": "per pixel: bytes (FETCH_SIZE x2, gfx950 calibration), VALU lane-instructions, SIMD-cycles with a VALU instruction executing (SQ_ACTIVE_INST_VALU x4)",
       "passes": {("|".join(k) if isinstance(k, tuple) else k): v for k, v in res.items()}}
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
with open(out, "w") as fh:
    json.dump(doc, fh, indent=1)
for k, v in doc["passes"].items():
    print("[pmc]", k, json.dumps(v), flush=True)
