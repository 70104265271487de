"""Torch-order statistics of a stored Lab image (ops.lab_stats_device, batch_size 1) for small batches, library builds interleaved:
median and spread of ROUNDS rounds.   python tools/bench_stats.py --libs a=...,b=... [--frames 1,2,4,8,16,32] [--rounds 7] [--json out]"""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, _hip
ap = argparse.ArgumentParser()
ap.add_argument("--libs", required=True)
ap.add_argument("--frames", default="1,2,4,8,16,32")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--json", default="")
ap.add_argument("--forms", default="throughput,latency", help="vrg_lab_stats_torch_ws_f32 / vrg_lab_stats_torch_lat_f32")
a = ap.parse_args()
libs = [(i.split("=", 1)[0] + ":" + f, _hip.load_library(os.path.abspath(i.split("=", 1)[1])), f == "latency") for i in a.libs.split(",") for f in a.forms.split(",")]
dev = torch.device("cuda", 0)
H, W = 2160, 3840
g = torch.Generator(device=dev).manual_seed(7)
Fmax = max(int(f) for f in a.frames.split(","))
lab = torch.rand((Fmax, H, W, 3), generator=g, device=dev) * 100.0 - 30.0
rows = []
for F in [int(f) for f in a.frames.split(",")]:
    x = lab[:F]
    outs, ts = {}, {n: [] for n, _, _ in libs}
    for rnd in range(a.rounds + 1):
        for n, lib, lat in libs:
            _hip._lib = lib
            e0, e1 = ops.HipEvent(), ops.HipEvent()
            e0.record(); o = ops.lab_stats_device(x, 1, latency_form=lat); e1.record()
            t = e0.elapsed_ms(e1)
            if rnd == 0:
                outs[n] = o.clone()
            else:
                ts[n].append(t)
    want = torch.stack([x.permute(0, 3, 1, 2).contiguous().mean(dim=[2, 3]), x.permute(0, 3, 1, 2).contiguous().std(dim=[2, 3]) + 1e-5], dim=-1)
    for n, _, _ in libs:
        med = statistics.median(ts[n])
        rows.append({"frames": F, "lib": n, "ms_median": round(med, 4), "ms_min": round(min(ts[n]), 4), "ms_max": round(max(ts[n]), 4),
                     "spread_pct": round(100 * (max(ts[n]) - min(ts[n])) / med, 1), "equals_torch": bool(torch.equal(outs[n], want))})
        print("[stats]", rows[-1], flush=True)
if a.json:
    with open(a.json, "w") as fh:
        json.dump(rows, fh, indent=1)
