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
spec = ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), colormatch=(ref_ms, 1.0) if which == "chain4" else None,
                     sharpen=("unsharp", 0.5, False))
acc = {}
for it in range(iters + 2):
    ev = []
    ops.fused_chain(x, spec, generator=gen, out=out, lab_workspace=ws, kernel_events=ev)
    torch.cuda.synchronize()
    if it >= 2:
        for name, a, b, nf in ev:
            acc.setdefault(name, []).append(a.elapsed_ms(b))
print(os.environ.get("VRGDG_HIP_LIB", "default"), which, {k: (round(statistics.median(v), 3), round(min(v), 3)) for k, v in acc.items()})
