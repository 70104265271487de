"""dev_pow_ziv's rounding test: (a) the distance between ocml's double-word logarithm and the table logarithm per table index j (and
exponent e), over EVERY fp32 of [0.0031308, 4] -- is a per-index bound tighter than the global one ziv_delta() uses? -- and (b) how
often the test fails per lane and per wave for the three call sites on uniform arguments.
    python tools/ziv_per_index.py -> gpurun_out/ziv_per_index.json"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip
dev = torch.device("cuda", 0)


def dbg(x, op, y=1.0):
    out = torch.empty_like(x)
    _hip.check(_hip.lib().vrg_debug_cm_math(_hip.ptr(x), _hip.ptr(out), x.numel(), op, y, _hip.current_stream()), "vrg_debug_cm_math")
    return out


def measure(dev=dev, lo=0.0031308, hi=4.0):
    """Every fp32 of [lo, hi]: per table index j the min / max of the signed difference ocml - table and the largest |difference|, per
    (exponent, index) the largest |difference|, and the largest difference relative to max(|e ln2|, |ln x|)."""
    a, b = int(np.float32(lo).view(np.int32)), int(np.float32(hi).view(np.int32))
    E0 = -10
    smin = torch.full((128,), 1.0, dtype=torch.float64, device=dev)
    smax = torch.full((128,), -1.0, dtype=torch.float64, device=dev)
    dmax = torch.zeros((16, 128), dtype=torch.float64, device=dev)          # [e - E0][j]: max |ocml - table|
    rmax = torch.zeros((16, 128), dtype=torch.float64, device=dev)          # the same relative to max(|e ln2|, |ln x|)
    for s in range(a, b + 1, 1 << 25):
        x = torch.arange(s, min(b + 1, s + (1 << 25)), dtype=torch.int64, device=dev).to(torch.int32).view(torch.float32)
        d32 = x.view(torch.int32) - 0x3F2AAAAB
        e = d32 >> 23
        j = (d32 & 0x007FFFFF) >> 16
        t = torch.log(x.double())
        scale = torch.maximum((e.double() * float(np.log(2.0))).abs(), t.abs()).clamp_min(1e-300)
        sd = (dbg(x, 16).double() + dbg(x, 17).double()) - (dbg(x, 18).double() + dbg(x, 19).double())
        smin.scatter_reduce_(0, j.long(), sd, reduce="amin")
        smax.scatter_reduce_(0, j.long(), sd, reduce="amax")
        d = sd.abs()
        key = ((e - E0) * 128 + j).long()
        dmax.view(-1).scatter_reduce_(0, key, d, reduce="amax")
        rmax.view(-1).scatter_reduce_(0, key, d / scale, reduce="amax")
    dm = dmax.cpu().numpy(); rm = rmax.cpu().numpy()
    per_j = dm.max(axis=0)
    out = {"global_abs_log2": float(np.log2(dm.max())), "global_rel_log2": float(np.log2(rm.max())), "global_rel": float(rm.max()),
           "per_j_abs_max": [float(v) for v in per_j],
           "per_j_signed_min": [float(v) for v in smin.cpu().numpy()], "per_j_signed_max": [float(v) for v in smax.cpu().numpy()],
           "per_j_abs_log2_quantiles": {q: round(float(np.log2(np.quantile(per_j[per_j > 0], q))), 2) for q in (0.1, 0.5, 0.9, 1.0)},
           "per_e_abs_log2": {int(E0 + i): round(float(np.log2(dm[i].max())), 2) for i in range(16) if dm[i].max() > 0},
           "per_e_rel_log2": {int(E0 + i): round(float(np.log2(rm[i].max())), 2) for i in range(16) if rm[i].max() > 0}}
    hw = (smax - smin).cpu().numpy() / 2
    out["per_j_halfwidth_log2_quantiles"] = {q: round(float(np.log2(np.quantile(hw, q))), 2) for q in (0.1, 0.5, 0.9, 1.0)}
    return out


if __name__ == "__main__":
    out = measure()
    # (b) failure rates: op 15 = 1.0 where the test fails.  Arguments as the Lab transforms see them for uniform pixels.
    g = torch.Generator(device=dev).manual_seed(5)
    u = torch.rand((1 << 24,), generator=g, device=dev)
    sites = {"srgb_to_linear (q^2.4, q = (v + 0.055) / 1.055)": ((u + 0.055) / 1.055, 2.4),
             "lab cube root (t^(1/3), t in [0.008856, 1.1])": (u * 1.09 + 0.008856, 1.0 / 3.0),
             "linear_to_srgb (c^(1/2.4))": (u * 0.9968692 + 0.0031308, 1.0 / 2.4)}
    rates = {}
    for name, (x, y) in sites.items():
        f = dbg(x.contiguous(), 15, y)
        lane = float(f.mean())
        wave = float(f.view(-1, 64).amax(dim=1).mean())
        rates[name] = {"lane_fail": lane, "wave_fail": wave}
    out["failure_rates"] = rates
    print(json.dumps({k: out[k] for k in ("global_abs_log2", "global_rel_log2", "per_j_abs_log2_quantiles", "per_j_halfwidth_log2_quantiles", "failure_rates")}, indent=0))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open(os.environ.get("ZIV_OUT", "gpurun_out/ziv_per_index.json"), "w"), indent=1)
