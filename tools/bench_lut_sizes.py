"""Chain 3 (grain -> LUT -> unsharp) and the LUT node alone against the cube size: the reference ships 25^3 / 32^3 / 33^3 cubes; this pack's
own are 17^3 (LDS-resident path), 25^3, 33^3; 32^3 and 65^3 are synthesised here.  64 x 4K frames, uniform and video-like pixels, median of ROUNDS.
    python tools/bench_lut_sizes.py [--frames 64] [--rounds 5] [--json out.json]"""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, VRGDG_IV_Adjustments as iv
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--json", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
H, W = 2160, 3840
px = a.frames * H * W


def synth(n):
    g = torch.linspace(0, 1, n)
    b, gg, r = torch.meshgrid(g, g, g, indexing="ij")
    t = torch.stack([r ** 0.9, gg * 0.95 + 0.02, b ** 1.1], dim=-1).contiguous()
    return {"lut": t, "size": n, "domain_min": torch.zeros(3), "domain_max": torch.ones(3)}


luts = {17: cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_Identity_17.cube")), 25: cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_WarmFilm_25.cube")),
        32: synth(32), 33: cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), 65: synth(65)}
rows = []
for dist in ("uniform", "video"):
    x = bench.make_frames(a.frames, H, W, dev, 1234, dist)
    for n, data in luts.items():
        try:
            lut = ops.upload_lut(data, dev)
        except Exception as exc:                       # a synthesised dict the uploader does not take
            print("[lut]", n, "skipped:", exc, flush=True)
            continue
        cases = {"LUT alone": lambda: ops.lut3d(x, lut, 10.0),
                 "grain -> LUT -> unsharp": lambda: ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False)))}
        for name, fn in cases.items():
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(a.rounds):
                e0, e1 = ops.HipEvent(), ops.HipEvent()
                e0.record(); o = fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_ms(e1)); del o
            med = statistics.median(ts)
            rows.append({"pixels": dist, "cube": n, "case": name, "ms_median": round(med, 3), "Gpix_s": round(px / med / 1e6, 1), "spread_pct": round(100 * (max(ts) - min(ts)) / med, 1)})
            print("[lut]", rows[-1], flush=True)
    del x
if a.json:
    json.dump(rows, open(a.json, "w"), indent=1)
