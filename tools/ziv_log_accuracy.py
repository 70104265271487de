"""The two double-word logarithms of the device colour-match policy, over EVERY fp32 of a domain (default [0.0031308, 4], the
union of the fast-path domains of the Lab transforms' powers):
  * each against a float64 log: ocml's epln as transcribed (vrg_debug_cm_math ops 16 / 17) and dev_pow_ziv's table log (18 / 19);
  * the DISTANCE between the two, relative to max(|e ln2|, |ln x|) and absolute -- the two maxima that ziv_delta()
    (csrc/vrg_pixel_math.hpp) is built from.
    python tools/ziv_log_accuracy.py [lo hi]      -> gpurun_out/ziv_log_accuracy.json"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_vrgamedevgirl_amd import _hip  # noqa: E402


def measure(lo=0.0031308, hi=4.0, dev=None):
    dev = dev or torch.device("cuda", 0)

    def dbg(x, op):
        out = torch.empty_like(x)
        _hip.check(_hip.lib().vrg_debug_cm_math(_hip.ptr(x), _hip.ptr(out), x.numel(), op, 1.0, _hip.current_stream()), "vrg_debug_cm_math")
        return out

    a, b = int(np.float32(lo).view(np.int32)), int(np.float32(hi).view(np.int32))
    w = {"ocml_rel": 0.0, "ocml_abs": 0.0, "table_rel": 0.0, "table_abs": 0.0, "distance_rel_to_max_eln2_lnx": 0.0, "distance_abs": 0.0}
    n = 0
    for s in range(a, b + 1, 1 << 25):
        x = torch.arange(s, min(b + 1, s + (1 << 25)), dtype=torch.int64, device=dev).to(torch.int32).view(torch.float32)
        n += x.numel()
        t = torch.log(x.double())
        e = ((x.view(torch.int32) - 0x3F2AAAAB) >> 23).double()
        scale = torch.maximum((e * float(np.log(2.0))).abs(), t.abs()).clamp_min(1e-300)
        lo_ = dbg(x, 16).double() + dbg(x, 17).double()
        lz = dbg(x, 18).double() + dbg(x, 19).double()
        for name, L in (("ocml", lo_), ("table", lz)):
            err = (L - t).abs()
            w[name + "_rel"] = max(w[name + "_rel"], float(torch.where(t != 0, err / t.abs().clamp_min(1e-300), torch.zeros_like(err)).max()))
            w[name + "_abs"] = max(w[name + "_abs"], float(err.max()))
        d = (lo_ - lz).abs()
        w["distance_rel_to_max_eln2_lnx"] = max(w["distance_rel_to_max_eln2_lnx"], float((d / scale).max()))
        w["distance_abs"] = max(w["distance_abs"], float(d.max()))
    out = {"domain": [lo, hi], "inputs": n}
    out.update({k + "_log2": float(np.log2(v)) for k, v in w.items()})
    out.update({k: v for k, v in w.items()})
    return out


if __name__ == "__main__":
    lo, hi = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (0.0031308, 4.0)
    res = measure(lo, hi)
    print({k: (round(v, 3) if isinstance(v, float) and k.endswith("_log2") else v) for k, v in res.items() if k.endswith("_log2") or k in ("domain", "inputs")})
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/ziv_log_accuracy.json", "w") as fh:
        json.dump(res, fh, indent=1)
