"""Driver for `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE`: the kernels of the bench legs and the v_fma probe, a few launches each on 64 4K frames.
The shader clock a kernel ran at = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / its duration (tools/summarize_pmc.py prints counter / duration)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip, ops, cube, VRGDG_IV_Adjustments as iv
import bench
dev = torch.device("cuda", 0)
F, H, W = int(os.environ.get("PROF_FRAMES", "64")), 2160, 3840
x = torch.rand((F, H, W, 3), generator=torch.Generator(device=dev).manual_seed(3), device=dev)
xv = bench.make_frames(F, H, W, dev, 1234, "video")
out = torch.empty_like(x); ws = torch.empty_like(x)
lut = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), dev)
gen = torch.Generator(device=dev)
ref_ms = ops.reference_stats(x[:1])
probe = torch.empty(768 * 256, dtype=torch.float32, device=dev)
def chain(src, **kw):
    gen.manual_seed(5)
    ops.fused_chain(src, ops.ChainSpec(**kw), generator=gen, out=out, lab_workspace=ws)
for rep in range(3):
    for mode in (0, 4, 2, 1):
        _hip.check(_hip.lib().vrg_debug_valu_rate(_hip.ptr(probe), 512, 20000, mode, _hip.current_stream()), "valu")
    _hip.check(_hip.lib().vrg_debug_copy_f32(_hip.ptr(x), _hip.ptr(out), x.numel(), 1, _hip.current_stream()), "copy")
    chain(x, grain=(0.04, 0.5, 4), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False))
    chain(xv, grain=(0.04, 0.5, 4), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False))
    chain(x, grain=(0.04, 0.5, 4), sharpen=("unsharp", 0.5, False))
    ops.film_grain(x, 0.04, 0.5, chunk_frames=4, generator=gen)
    ops.stencil3x3(x, "unsharp", 0.5, False)
    ops.lut3d(x, lut, 10.0)
    chain(x, grain=(0.04, 0.5, 4), lut=(lut, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False), cm_chunk=1)
    chain(x, colormatch=(ref_ms, 1.0), cm_chunk=1)
    torch.cuda.synchronize()
print("done clock")
