"""One-off probe: which op of the vignette differs on the device?  x = 1 makes the output the mask itself."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, VRGDG_LUTVideoTools as LVT
f = np.float32
for (H, W) in ((20, 27), (270, 480)):
    x = torch.ones(1, H, W, 3)
    v = 0.65
    got = ops.adjust(x.cuda(), ops.adjust_terms(LVT._normalize_adjust_settings({"vignette": v * 100}))).cpu().numpy()[0, :, :, 0]
    yy = torch.linspace(-1, 1, H).numpy().reshape(H, 1); xx = torch.linspace(-1, 1, W).numpy().reshape(1, W)
    a = (xx * xx).astype(f); b = (yy * yy).astype(f)
    s = (a + b).astype(f)
    s_fma = (xx.astype(np.float64) ** 2 + b.astype(np.float64)).astype(f)
    s_fma2 = (yy.astype(np.float64) ** 2 + a.astype(np.float64)).astype(f)
    def finish(d, recip=False):
        e = (d - f(0.35)).astype(f)
        q = (e * f(1.0 / 1.05)).astype(f) if recip else (e / f(1.05)).astype(f)
        q = np.clip(q, 0, 1)
        m = (f(1.0) - ((q * f(v)).astype(f) * f(0.75)).astype(f)).astype(f)
        return np.clip(m, 0, 1)
    alts = {"ieee": finish(np.sqrt(s)), "recip": finish(np.sqrt(s), True), "fma_x": finish(np.sqrt(s_fma)), "fma_y": finish(np.sqrt(s_fma2)),
            "torch_cpu_sqrt": finish(torch.sqrt(torch.from_numpy(s)).numpy()),
            "torch_gpu_sqrt": finish(torch.sqrt(torch.from_numpy(s).cuda()).cpu().numpy()),
            "sqrt64": finish(np.sqrt(s.astype(np.float64)).astype(f))}
    for k, m in alts.items():
        print(H, W, k, "mismatches", int(np.sum(m != got)), "of", got.size)
    yg = torch.linspace(-1, 1, H, device="cuda").cpu().numpy().reshape(H, 1)
    print("torch gpu linspace == cpu:", np.array_equal(yg, yy))
    tg = ((torch.ones(1, device="cuda") * torch.from_numpy(s).cuda() - 0.35) / 1.05).cpu().numpy()
    print("torch gpu (d-0.35)/1.05 == ieee:", np.array_equal(tg, ((s - f(0.35)).astype(f) / f(1.05)).astype(f)))
