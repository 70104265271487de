"""Differential run of two builds of libvrgdg_hip.so over adversarial frames: every chain with a colour transfer in it (the kernels the arithmetic
series of round 6 touched) on uniform / video-like / out-of-range / NaN- and Inf-sprinkled / constant / grey / denormal frames, both policies, three
stencils, both borders, partial match strength, partial LUT strength, a non-unit LUT domain -- output bits compared (NaN positions and payload-blind).
    python tools/diff_libraries.py --base tools/ab/lib_r6_head.so [--new comfyui-vrgamedevgirl_amd/libvrgdg_hip.so] [--out gpurun_out/diff.json]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import bench
from comfyui_vrgamedevgirl_amd import ops, cube, _hip, VRGDG_IV_Adjustments as iv

ap = argparse.ArgumentParser()
ap.add_argument("--base", required=True)
ap.add_argument("--new", default=os.path.join(ROOT, "comfyui-vrgamedevgirl_amd", "libvrgdg_hip.so"))
ap.add_argument("--out", default="")
a = ap.parse_args()
libs = {"base": _hip.load_library(os.path.abspath(a.base)), "new": _hip.load_library(os.path.abspath(a.new))}
dev = torch.device("cuda", 0)
F, H, W = 8, 540, 960
g = torch.Generator(device=dev).manual_seed(11)


def frames(kind):
    x = torch.rand((F, H, W, 3), generator=g, device=dev)
    if kind == "video":
        return bench.make_frames(F, H, W, dev, 77, "video")
    if kind == "wide":                       # [-2, 3]: below zero and above one
        return x * 5.0 - 2.0
    if kind == "huge":                       # up to 1e30, with negatives
        return (x - 0.3) * torch.exp(torch.rand((F, H, W, 3), generator=g, device=dev) * 70.0)
    if kind == "specials":
        m = torch.rand((F, H, W, 3), generator=g, device=dev)
        x = torch.where(m < 0.01, torch.full_like(x, float("nan")), x)
        x = torch.where((m >= 0.01) & (m < 0.02), torch.full_like(x, float("inf")), x)
        x = torch.where((m >= 0.02) & (m < 0.03), torch.full_like(x, -float("inf")), x)
        x = torch.where((m >= 0.03) & (m < 0.05), torch.zeros_like(x), x)
        x = torch.where((m >= 0.05) & (m < 0.07), torch.ones_like(x), x)
        x = torch.where((m >= 0.07) & (m < 0.08), -torch.zeros_like(x), x)
        return x
    if kind == "constant":
        return torch.full((F, H, W, 3), 0.37, device=dev) + torch.zeros((F, 1, 1, 3), device=dev)
    if kind == "grey":
        return x[..., :1].expand(F, H, W, 3).contiguous()
    if kind == "denormal":
        return x * 1e-39
    if kind == "black_and_one":
        return (x > 0.5).float()
    return x


lut33 = cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube"))
lut_dom = dict(lut33)
lut_dom["domain_min"] = torch.tensor([-0.25, 0.0, 0.1])
lut_dom["domain_max"] = torch.tensor([1.5, 1.0, 0.9])
gen = torch.Generator(device=dev)
report, bad = [], 0
for kind in ("uniform", "video", "wide", "huge", "specials", "constant", "grey", "denormal", "black_and_one"):
    x = frames(kind)
    ref = frames("uniform")[:1] if kind != "specials" else x[:1].clone()
    cases = []
    for math in ("device", "fast"):
        for sharpen in (None, ("unsharp", 0.5, False), ("unsharp", 1.5, True), ("laplacian", 0.7, False), ("sobel", 0.4, True)):
            for k in (1.0, 0.6):
                cases.append(("cm", math, sharpen, k, None, None))
        cases.append(("chain4", math, ("unsharp", 0.5, False), 1.0, (lut33, 10.0), (0.04, 0.5, 4)))
        cases.append(("chain4 partial LUT strength, laplacian", math, ("laplacian", 0.3, False), 0.8, (lut33, 4.0), (0.04, 0.5, 2)))
        cases.append(("chain4 non-unit LUT domain, no sharpen", math, None, 1.0, (lut_dom, 10.0), (0.1, 0.2, 1)))
        cases.append(("grain -> cm", math, ("sobel", 0.2, False), 1.0, None, (0.04, 0.5, 4)))
    for name, math, sharpen, k, lut, grain in cases:
        outs = {}
        for ln, lib in libs.items():
            _hip._lib = lib
            ops._TOOLCHAIN.clear() if hasattr(ops, "_TOOLCHAIN") and isinstance(ops._TOOLCHAIN, dict) else None
            ref_ms = ops.reference_stats(ref, cm_math=math)
            dl = ops.upload_lut(lut[0], dev) if lut else None
            gen.manual_seed(5)
            spec = ops.ChainSpec(grain=grain, lut=(dl, lut[1]) if lut else None, colormatch=(ref_ms, k), sharpen=sharpen, cm_math=math, cm_chunk=1)
            outs[ln] = ops.fused_chain(x, spec, generator=gen).clone()
        b, n = outs["base"], outs["new"]
        same = bool(((b.view(torch.int32) == n.view(torch.int32)) | (torch.isnan(b) & torch.isnan(n))).all())
        if not same:
            bad += 1
            d = (b.view(torch.int32) != n.view(torch.int32)) & ~(torch.isnan(b) & torch.isnan(n))
            print("[diff] DIFFERENT", kind, name, math, sharpen, k, int(d.sum()), "elements", flush=True)
        report.append({"frames": kind, "case": name, "cm_math": math, "sharpen": sharpen, "k": k, "identical": same, "nan_share": float(torch.isnan(n).float().mean())})
    print("[diff]", kind, "done:", sum(1 for r in report if r["frames"] == kind and r["identical"]), "of", sum(1 for r in report if r["frames"] == kind), "identical", flush=True)
print("[diff] total", len(report), "cases,", bad, "different")
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"base": a.base, "new": a.new, "frames": [F, H, W], "cases": report, "different": bad}, open(a.out, "w"), indent=1)
sys.exit(1 if bad else 0)
