"""Exhaustive sweep of the statistics kernels' division by the running count (reciprocal + two FMAs) against the IEEE quotient: every
count 1 .. 2^20 x every fp32 significand (8.8e12 divisions, a few seconds of GPU).   python tools/welford_division_sweep.py [--out ...]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip
ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/welford_division_sweep.json")
ap.add_argument("--max-count", type=int, default=1 << 20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
mis = torch.zeros(1, dtype=torch.int64, device=dev)
t0 = time.perf_counter()
step = 1 << 16
for n0 in range(1, a.max_count + 1, step):
    cnt = min(step, a.max_count + 1 - n0)
    _hip.check(_hip.lib().vrg_selftest_welford_division(_hip.ptr(mis), n0, cnt, _hip.current_stream()), "vrg_selftest_welford_division")
torch.cuda.synchronize()
res = {"counts": [1, a.max_count], "significands_per_count": 1 << 23, "divisions": a.max_count * (1 << 23), "mismatches": int(mis.item()),
       "seconds": round(time.perf_counter() - t0, 2), "device": torch.cuda.get_device_name(0)}
print(res)
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
json.dump(res, open(a.out, "w"), indent=1)
