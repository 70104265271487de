"""Exhaustive sweep of div_uniform_ieee (the colour-match division by the per-frame sigma: reciprocal + two corrections) against the IEEE
quotient: every fp32 significand of sigma x every fp32 significand of the numerator = 7.04e13 divisions (about half a minute of GPU).
    python tools/div_sigma_sweep.py [--out ...] [--sigmas N]      (--sigmas: only the first N significands + N spread over the rest)"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip
ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/div_sigma_sweep.json")
ap.add_argument("--sigmas", type=int, default=1 << 23)
a = ap.parse_args()
dev = torch.device("cuda", 0)
mis = torch.zeros(1, dtype=torch.int64, device=dev)
t0 = time.perf_counter()
step = 1 << 16
n = 0
for s0 in range(0, a.sigmas, step):
    cnt = min(step, a.sigmas - s0)
    _hip.check(_hip.lib().vrg_selftest_div_sigma(_hip.ptr(mis), s0, cnt, _hip.current_stream()), "vrg_selftest_div_sigma")
    n += cnt
    if (s0 // step) % 16 == 15:
        torch.cuda.synchronize()
        print(f"[div] {n} sigmas, {int(mis.item())} mismatches, {time.perf_counter() - t0:.1f} s", flush=True)
torch.cuda.synchronize()
res = {"sigma_significands": n, "numerator_significands": 1 << 23, "divisions": n * (1 << 23), "mismatches": int(mis.item()),
       "seconds": round(time.perf_counter() - t0, 2), "device": torch.cuda.get_device_name(0)}
print(res)
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
json.dump(res, open(a.out, "w"), indent=1)
