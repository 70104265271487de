"""One launch each of the two gather-free forms of the march on 16 x 4K uniform frames, for rocprofv3 --pmc passes:
grain -> unsharp (k_chain_march<1, true, 4>) and grain -> LUT 17^3 -> unsharp with the node table in LDS (k_chain_march<3, true, 12>)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, VRGDG_IV_Adjustments as iv
import bench
dev = torch.device("cuda", 0)
x = bench.make_frames(16, 2160, 3840, dev, 1234, "uniform")
out = torch.empty_like(x)
lut17 = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_Identity_17.cube")), dev)
gen = torch.Generator(device=dev)
for rep in range(2):
    gen.manual_seed(5)
    ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), sharpen=("unsharp", 0.5, False)), generator=gen, out=out)
    gen.manual_seed(5)
    ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut17, 10.0), sharpen=("unsharp", 0.5, False)), generator=gen, out=out)
torch.cuda.synchronize()
