"""Host-fed four-node graph (deferred, fused) against the size of a pipelined piece (_devices.PIPE_BYTES): pieces are whole grain chunks, so the
sweep only moves batches whose chunk is smaller than the piece.   python tools/sweep_piece_size.py [--json gpurun_out/piece_size.json]"""
import argparse, json, os, statistics, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import nodes, _devices, VRGDG_IV_Adjustments as iv
ap = argparse.ArgumentParser()
ap.add_argument("--json", default="")
a = ap.parse_args()
rows = []
for H, W, F, bs in ((1080, 1920, 32, 4), (1080, 1920, 16, 4), (720, 1280, 48, 4), (2160, 3840, 16, 4), (2160, 3840, 16, 1), (1080, 1920, 32, 1)):
    x = torch.rand((F, H, W, 3), generator=torch.Generator().manual_seed(3))
    ref = x[:1].clone()

    def graph():
        t = nodes.FastFilmGrain().apply_grain(x, 0.04, 0.5, bs)[0]
        t = iv.VRGDG_LUTS().apply_lut(t, "AMD_TealOrange_33.cube", "auto", 10.0)[0]
        t = nodes.ColorMatchToReference().match_color(t, ref, 1.0, 1)[0]
        t = nodes.FastUnsharpSharpen().apply_unsharp(t, 0.5, False)[0]
        _devices.materialise(t)
        torch.cuda.synchronize()
    for mb in (32, 64, 128, 256, 512):
        _devices.PIPE_BYTES = mb << 20
        for _ in range(3):
            graph()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); graph(); ts.append(time.perf_counter() - t0)
        ms = statistics.median(ts) * 1e3
        rows.append({"H": H, "W": W, "frames": F, "grain_batch_size": bs, "piece_MB": mb, "ms": round(ms, 2), "Mpix_s": round(F * H * W / ms / 1e3, 1)})
        print(json.dumps(rows[-1]), flush=True)
if a.json:
    os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
    json.dump({"rows": rows}, open(a.json, "w"), indent=1)
