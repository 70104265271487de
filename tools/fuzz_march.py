"""Random mid-size geometries through the march kernel (variant 2: steady rows, quad-cooperative gathers, edge rows with shared calls)
against the LDS-tile / point-wise kernels (variant 1: an independent implementation of the same chain): output bits.
    python tools/fuzz_march.py [--cases 150] [--seed 1] [--json out.json]"""
import argparse, json, os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, VRGDG_IV_Adjustments as iv
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=150)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--json", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
rnd = random.Random(a.seed)
luts = {n: ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, n)), dev) for n in ("AMD_TealOrange_33.cube", "AMD_WarmFilm_25.cube", "AMD_Identity_17.cube")}
bad, rows = [], []
for i in range(a.cases):
    W = rnd.choice([rnd.randint(64, 400), rnd.randint(400, 2100), 1920, 3840, 1280, 61 * rnd.randint(2, 30), 61 * rnd.randint(2, 30) + 1])
    H = rnd.choice([rnd.randint(3, 60), rnd.randint(60, 700), 1080, 720])
    bs = rnd.choice([1, 1, 2, 3, 4])
    F = bs * rnd.randint(1, 3)
    while F * H * W * 3 > 160_000_000:
        H = max(3, H // 2)
    grain = (round(rnd.uniform(0.005, 0.3), 3), round(rnd.uniform(0, 1), 2), bs) if rnd.random() < 0.85 else None
    name = rnd.choice(list(luts))
    lut = (luts[name], rnd.choice([10.0, 10.0, round(rnd.uniform(0.5, 9.5), 1)])) if rnd.random() < 0.8 else None
    sharpen = (rnd.choice(["unsharp", "unsharp", "laplacian", "sobel"]), round(rnd.uniform(0.05, 2.0), 2), rnd.random() < 0.3) if rnd.random() < 0.9 else None
    if grain is None and lut is None and sharpen is None:
        sharpen = ("unsharp", 0.5, False)
    g = torch.Generator(device=dev).manual_seed(1000 + i)
    kind = rnd.choice(["uniform", "wide", "smooth"])
    x = torch.rand((F, H, W, 3), generator=g, device=dev)
    if kind == "wide":
        x = x * 1.2 - 0.1
    elif kind == "smooth":
        x = (x * 0.05 + torch.linspace(0, 1, W, device=dev)[None, None, :, None] * 0.9).clamp_(0, 1).contiguous()
    outs = []
    for variant in (1, 2):
        torch.manual_seed(77 + i)
        outs.append(ops.fused_chain(x, ops.ChainSpec(grain=grain, lut=lut, sharpen=sharpen, variant=variant)))
    same = torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    row = {"case": i, "F": F, "H": H, "W": W, "bs": bs, "grain": grain, "lut": (name, lut[1]) if lut else None, "sharpen": sharpen, "data": kind, "bit_equal": bool(same)}
    rows.append(row)
    if not same:
        d = (outs[0] != outs[1]) & ~(torch.isnan(outs[0]) & torch.isnan(outs[1]))
        row["differing"] = int(d.sum())
        bad.append(row)
        print("[fuzz] DIFF", row, flush=True)
print("[fuzz]", a.cases, "cases,", len(bad), "differ", flush=True)
if a.json:
    json.dump({"cases": a.cases, "seed": a.seed, "differ": bad, "rows": rows}, open(a.json, "w"), indent=0)
sys.exit(1 if bad else 0)
