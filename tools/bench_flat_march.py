"""grain -> LUT (no stencil) and grain alone: the march kernel (variant 2) against the point-wise kernels (variant 1), and the automatic choice.
    python tools/bench_flat_march.py [--json out.json]"""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, VRGDG_IV_Adjustments as iv
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--json", default="")
ap.add_argument("--sizes", default="128x1080x1920,64x2160x3840,8x1080x1920", help="FxHxW,...")
ap.add_argument("--cubes", default="33,25,17")
a = ap.parse_args()
dev = torch.device("cuda", 0)
luts = {n: ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, f)), dev) for n, f in ((33, "AMD_TealOrange_33.cube"), (25, "AMD_WarmFilm_25.cube"), (17, "AMD_Identity_17.cube"))}
rows = []
for (F, H, W) in [tuple(int(v) for v in z.split("x")) for z in a.sizes.split(",")]:
    for dist in ("uniform", "video"):
        x = bench.make_frames(F, H, W, dev, 1234, dist)
        px = F * H * W
        for n, lut in luts.items():
            if str(n) not in a.cubes.split(","):
                continue
            for bs in (4,):
                outs, ts = {}, {}
                for rnd in range(a.rounds + 1):
                    for variant in (1, 2, 0):
                        torch.manual_seed(5)
                        e0, e1 = ops.HipEvent(), ops.HipEvent()
                        e0.record(); o = ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, bs), lut=(lut, 10.0), variant=variant)); e1.record(); torch.cuda.synchronize()
                        if rnd == 0:
                            outs[variant] = o
                        else:
                            ts.setdefault(variant, []).append(e0.elapsed_ms(e1)); del o
                same = torch.equal(outs[1].view(torch.int32), outs[2].view(torch.int32)) and torch.equal(outs[1].view(torch.int32), outs[0].view(torch.int32))
                row = {"frames": F, "H": H, "pixels": dist, "cube": n, "bit_equal": bool(same)}
                for v, name in ((1, "pointwise"), (2, "march"), (0, "auto")):
                    med = statistics.median(ts[v])
                    row[name + "_ms"] = round(med, 3); row[name + "_Gpix_s"] = round(px / med / 1e6, 1)
                rows.append(row)
                print("[flat]", row, flush=True)
                del outs
        del x
if a.json:
    json.dump(rows, open(a.json, "w"), indent=1)
