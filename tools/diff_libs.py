"""Where do two library builds differ?  chain 3 (grain -> LUT -> unsharp) on F x 4K frames with both builds; prints the rows / columns /
frames of differing pixels.   python tools/diff_libs.py a.so b.so [frames]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, _hip, VRGDG_IV_Adjustments as iv
la, lb = _hip.load_library(os.path.abspath(sys.argv[1])), _hip.load_library(os.path.abspath(sys.argv[2]))
F = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
H, W = 2160, 3840
g = torch.Generator(device=dev).manual_seed(99)
x = torch.rand((F, H, W, 3), generator=g, device=dev)
lut = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), dev)
gen = torch.Generator(device=dev)
outs = []
for lib in (la, lb):
    _hip._lib = lib
    gen.manual_seed(5)
    o = torch.empty_like(x)
    ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False)), generator=gen, out=o)
    torch.cuda.synchronize()
    outs.append(o)
d = (outs[0] != outs[1]).any(dim=-1)
idx = torch.nonzero(d).cpu()
print("differing pixels:", idx.shape[0])
rows = collections.Counter((int(f), int(y)) for f, y, xx in idx.tolist())
print("by (frame,row):", sorted(rows.items())[:40])
cols = collections.Counter(int(xx) % 61 for f, y, xx in idx.tolist())
print("by column mod 61:", sorted(cols.items())[:70])
G = 524288
for f, y, xx in idx[:12].tolist():
    e = ((f % 4) * H * W + y * W + xx) * 3
    print((f, y, xx), "elem", e, "q", e // G, "k", e // (4 * G), "idx", e % G, "a", outs[0][f, y, xx].tolist(), "b", outs[1][f, y, xx].tolist())
