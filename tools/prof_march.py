"""One launch of chain 3 (grain -> LUT 33^3 -> unsharp) on uniform and on video-like 4K frames, for rocprofv3 --pmc passes.
    python tools/prof_march.py [frames]        (VRGDG_HIP_LIB selects the build)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, VRGDG_IV_Adjustments as iv
import bench
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
H, W = 2160, 3840
lut = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), dev)
gen = torch.Generator(device=dev)
for dist in ("uniform", "video"):
    x = bench.make_frames(F, H, W, dev, 1234, dist)
    out = torch.empty_like(x)
    for rep in range(2):
        gen.manual_seed(5)
        ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False)), generator=gen, out=out)
    torch.cuda.synchronize()
    del x, out
