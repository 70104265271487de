"""dev_pow_ziv against torch.pow on the device: every fp32 base of each call site's fast-path domain, a stride-7 sample of
everything above and below it, specials; and the fraction of lanes the Ziv test sends to the transcription.
    python tools/ziv_check.py"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip
dev = torch.device("cuda", 0)
def dbg(x, op, y):
    out = torch.empty_like(x)
    _hip.check(_hip.lib().vrg_debug_cm_math(_hip.ptr(x), _hip.ptr(out), x.numel(), op, float(np.float32(y)), _hip.current_stream()), "dbg")
    return out
def floats(lo, hi, step=1):
    a, b = int(np.float32(lo).view(np.int32)), int(np.float32(hi).view(np.int32))
    return torch.arange(a, b + 1, step, dtype=torch.int64, device=dev).to(torch.int32).view(torch.float32)
res = {}
for op, y, lo, hi in ((12, 2.4, 0.0625, 2.0), (13, 1 / 2.4, 0.0031308, 4.0), (14, 1 / 3.0, 0.008856, 4.0)):
    x = floats(lo, hi)
    want = torch.pow(x, float(np.float32(y)))
    got = dbg(x, op, y)
    bad = int((got != want).sum())
    wide = torch.cat([floats(2.0 ** -20, lo, 7), floats(hi, 3.0e38, 7), torch.tensor([float("inf"), float("nan"), 1.0, lo, hi], device=dev)])
    gw, ww = dbg(wide, op, y), torch.pow(wide, float(np.float32(y)))
    badw = int(((gw != ww) & ~(torch.isnan(gw) & torch.isnan(ww))).sum())
    slow = float(dbg(x, 15, y).mean()) if True else None
    res[str(op)] = {"y": y, "domain": [lo, hi], "inputs": x.numel(), "mismatches": bad, "outside_sampled": wide.numel(), "outside_mismatches": badw,
                    "ziv_fallback_fraction": slow}
    print(op, res[str(op)], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/ziv_check.json", "w"), indent=1)
