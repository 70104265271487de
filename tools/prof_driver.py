"""Tiny driver for rocprofv3 runs: executes selected kernels a few times on 16 4K frames.

    python tools/prof_driver.py lut|chain3|chain3_v1|chain4|grain|cm|sharpen [--smooth]
"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube
from comfyui_vrgamedevgirl_amd import VRGDG_IV_Adjustments as iv

which = sys.argv[1] if len(sys.argv) > 1 else "lut"
dev = torch.device("cuda", 0)
F, H, W = 16, 2160, 3840
g = torch.Generator(device=dev).manual_seed(3)
x = torch.rand((F, H, W, 3), generator=g, device=dev)
if "--smooth" in sys.argv:
    x = (x * 0.05 + 0.5 * (torch.linspace(0, 1, W, device=dev).view(1, 1, W, 1) + torch.linspace(0, 1, H, device=dev).view(1, H, 1, 1)) * 0.9).contiguous()
out = torch.empty_like(x)
lut = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), dev)
ref_ms = ops.finalize_stats(ops.lab_stats(x[:1]))
gen = torch.Generator(device=dev).manual_seed(5)
specs = {
    "chain3": ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False)),
    "chain3_v1": ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False), variant=1),
    "chain4": ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False)),
    "chain4fast": ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False), cm_math="fast"),
    "lutsharp": ops.ChainSpec(lut=(lut, 10.0), sharpen=("unsharp", 0.5, False)),
    "grainsharp": ops.ChainSpec(grain=(0.04, 0.5, 4), sharpen=("unsharp", 0.5, False)),
    "sharpen": ops.ChainSpec(sharpen=("unsharp", 0.5, False)),
}
if which == "traffic":
    # known-byte kernels first (calibration), then the headline chain
    ops.lab_stats(x)                         # k_lab_partials<0>: reads 12 B/px, writes ~nothing
    ops.lut3d(x, lut, 10.0)                  # k_lut3d: reads 12 B/px (+LUT, cache resident), writes 12 B/px
    ops.stencil3x3(x, "unsharp", 0.5, False) # k_chain_tile<0>: reads 12 B/px (+9.6% halo), writes 12 B/px
    lab_ws = torch.empty_like(x)
    ops.fused_chain(x, specs["chain4"], generator=gen, out=out, lab_workspace=lab_ws)
    ops.fused_chain(x, specs["chain3"], generator=gen, out=out)
    ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0)), generator=gen, out=out)      # bench leg grain_lut_1080p's kernel
    ops.fused_chain(x, ops.ChainSpec(colormatch=(ref_ms, 1.0)), out=out, lab_workspace=lab_ws)             # bench leg colormatch_4k's three passes
    ops.sharpen_then_seeded_grain(x, 0.5, False, 0.04, 0.5, 42, 0)      # k_sharpen_grain: 12 + 12 B/px if the row re-reads stay in L2
    torch.cuda.synchronize()
    print("done traffic")
    sys.exit(0)
if which == "l2":
    # L2 request counts of the march vs the tile form, with and without the LUT / grain stages
    import dataclasses
    for name in ("sharpen", "lutsharp", "grainsharp", "chain3"):
        for v in (1, 2):
            ops.fused_chain(x, dataclasses.replace(specs[name], variant=v), generator=gen, out=out)
    ops.lut3d(x, lut, 10.0)
    torch.cuda.synchronize()
    print("done l2")
    sys.exit(0)
if which == "issue":
    # calibration (known instruction count) + the kernels whose VALU instruction counts DESIGN.md quotes
    from comfyui_vrgamedevgirl_amd import _hip, VRGDG_LUTVideoTools as LVT
    probe = torch.empty(2048 * 256, dtype=torch.float32, device=dev)
    _hip.check(_hip.lib().vrg_debug_valu_rate(_hip.ptr(probe), 2048, 512, 0, _hip.current_stream()), "valu")   # 2048*4 waves * 512*64 v_fma
    lab_ws = torch.empty_like(x)
    ops.fused_chain(x, specs["chain4"], generator=gen, out=out, lab_workspace=lab_ws)
    ops.fused_chain(x, specs["chain4fast"], generator=gen, out=out, lab_workspace=lab_ws)
    ops.fused_chain(x, specs["chain3"], generator=gen, out=out)
    ops.fused_chain(x, specs["grainsharp"], generator=gen, out=out)
    ops.film_grain(x, 0.04, 0.5, chunk_frames=4, generator=gen)
    ops.lut3d(x, lut, 10.0)
    ops.stencil3x3(x, "unsharp", 0.5, False)
    ops.sharpen_then_seeded_grain(x, 0.5, False, 0.04, 0.5, 42, 0)
    ops.adjust(x, ops.adjust_terms(LVT._normalize_adjust_settings({"clarity": 40, "contrast": 12})), out=out)
    ops.adjust(x, ops.adjust_terms(LVT._normalize_adjust_settings({"sharpen": 40, "contrast": 12})), out=out)
    ops.adjust(x, ops.adjust_terms(LVT._normalize_adjust_settings({"temperature": 20, "exposure": 10, "contrast": 12, "saturation": 8,
                                                                   "highlights": -20, "shadows": 15, "fade": 10, "vignette": 30})), out=out)
    torch.cuda.synchronize()
    print("done issue")
    sys.exit(0)
for _ in range(3):
    if which == "lut":
        ops.lut3d(x, lut, 10.0)
    elif which == "grain":
        ops.film_grain(x, 0.04, 0.5, chunk_frames=4, generator=gen)
    elif which == "cm":
        ops.color_match(x, None, 1.0, ref_ms=ref_ms)
    else:
        ops.fused_chain(x, specs[which], generator=gen, out=out)
torch.cuda.synchronize()
print("done", which)
