"""A/B of two library builds: per-pass HIP-event times of the 4-stage chain (or chain 3), one build per process.
    VRGDG_HIP_LIB=tools/ab/lib_a.so python tools/ab_pass_times.py [chain4|chain3] [frames] [iters]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, VRGDG_IV_Adjustments as iv

which = sys.argv[1] if len(sys.argv) > 1 else "chain4"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
x = torch.rand((F, 2160, 3840, 3), generator=g, device=dev)
out = torch.empty_like(x); ws = torch.empty_like(x)
lut = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), dev)
ref_ms = ops.finalize_stats(ops.lab_stats(x[:1]))
gen = torch.Generator(device=dev).manual_seed(5)
spec = ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), colormatch=(ref_ms, 1.0) if which.startswith("chain4") else None,
                     sharpen=("unsharp", 0.5, False), cm_math="fast" if which == "chain4fast" else None, variant=int(os.environ.get("VRGDG_VARIANT", "0")))
if which == "kernels":
    # stand-alone kernels: median HIP-event ms each
    from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
    lut25 = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_WarmFilm_25.cube")), dev)
    t_cl = ops.adjust_terms(LVT._normalize_adjust_settings({"clarity": 40, "contrast": 12}))
    cases = {"grain": lambda: ops.film_grain(x, 0.04, 0.5, chunk_frames=4, generator=gen),
             "lut33": lambda: ops.lut3d(x, lut, 10.0), "lut25": lambda: ops.lut3d(x, lut25, 10.0),
             "unsharp": lambda: ops.stencil3x3(x, "unsharp", 0.5, False), "sobel": lambda: ops.stencil3x3(x, "sobel", 0.5, False),
             "clarity": lambda: ops.adjust(x, t_cl, out=out),
             "seeded grain": lambda: ops.film_grain_seeded_frames(x, 0.04, 0.5, 42, 0),
             "unsharp>seeded grain (two kernels)": lambda: ops.film_grain_seeded_frames(ops.stencil3x3(x, "unsharp", 0.5, False), 0.04, 0.5, 42, 0),
             "unsharp>seeded grain (fused)": lambda: ops.sharpen_then_seeded_grain(x, 0.5, False, 0.04, 0.5, 42, 0),
             "unsharp>seeded grain (fused, zero border)": lambda: ops.sharpen_then_seeded_grain(x, 0.5, True, 0.04, 0.5, 42, 0),
             "grain+lut": lambda: ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0)), generator=gen, out=out),
             "grain+sharpen": lambda: ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), sharpen=("unsharp", 0.5, False)), generator=gen, out=out),
             "chain3_25": lambda: ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut25, 10.0), sharpen=("unsharp", 0.5, False)), generator=gen, out=out)}
    res = {}
    for name, fn in cases.items():
        ts = []
        for it in range(iters + 2):
            a, b = ops.HipEvent(), ops.HipEvent()
            a.record(); fn(); b.record()
            if it >= 2:
                ts.append(a.elapsed_ms(b))
        res[name] = round(statistics.median(ts), 3)
    print(os.environ.get("VRGDG_HIP_LIB", "default"), which, res)
    sys.exit(0)
acc = {}
for it in range(iters + 2):
    ev = []
    w0, w1 = ops.HipEvent(), ops.HipEvent()
    w0.record()
    ops.fused_chain(x, spec, generator=gen, out=out, lab_workspace=ws, kernel_events=ev)
    w1.record()
    torch.cuda.synchronize()
    if it >= 2:
        tot = {}
        for name, a, b, nf in ev:
            tot[name] = tot.get(name, 0.0) + a.elapsed_ms(b)          # pieces of a pass added up (they may overlap other streams' work)
        tot["wall"] = w0.elapsed_ms(w1)
        for k, v in tot.items():
            acc.setdefault(k, []).append(v)
print(os.environ.get("VRGDG_HIP_LIB", "default"), which, "variant=" + os.environ.get("VRGDG_VARIANT", "0"), {k: (round(statistics.median(v), 3), round(min(v), 3)) for k, v in acc.items()})
