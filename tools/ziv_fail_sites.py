"""How often dev_pow_ziv's rounding test fails (lane / wave) on the arguments the headline chain's pass 1 really feeds it: uniform frames ->
grain -> LUT 33 (product kernels), then the reference's sRGB -> linear -> XYZ -> f(t) arithmetic in torch for the arguments.
    python tools/ziv_fail_sites.py"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, _hip, VRGDG_IV_Adjustments as iv
import bench
dev = torch.device("cuda", 0)


def fails(x, y):
    x = x.contiguous().view(-1)
    out = torch.empty_like(x)
    _hip.check(_hip.lib().vrg_debug_cm_math(_hip.ptr(x), _hip.ptr(out), x.numel(), 15, y, _hip.current_stream()), "dbg")
    n = x.numel() // 64 * 64
    return {"lane_fail": float(out.mean()), "wave_fail": float(out[:n].view(-1, 64).amax(dim=1).mean())}, out


x = bench.make_frames(2, 2160, 3840, dev, 1234, "uniform")
lut = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), dev)
torch.manual_seed(1)
rgb = ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0)))
v = rgb.reshape(-1, 3)
q = torch.clamp_min((v + 0.055) / 1.055, 0.0625)
res = {}
res["srgb_to_linear"], f1 = fails(q, 2.4)
lin = torch.where(v > 0.04045, torch.pow(q, 2.4), v / 12.92)
M = torch.tensor([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], device=dev)
xyz = lin @ M.t()
t = xyz / torch.tensor([0.95047, 1.0, 1.08883], device=dev)
tc = torch.clamp_min(t, 0.008856)
res["lab cube root"], f2 = fails(tc, 1.0 / 3.0)
# which arguments fail: histogram over log2 distance to 1 and the most frequent failing values
for name, arg, f in (("srgb_to_linear", q.reshape(-1), f1), ("lab cube root", tc.reshape(-1), f2)):
    bad = arg[f > 0]
    vals, counts = torch.unique(bad, return_counts=True)
    top = torch.argsort(counts, descending=True)[:5]
    res[name]["failing_lanes"] = int(bad.numel())
    res[name]["most_frequent_failing_values"] = [(float(vals[i]), int(counts[i])) for i in top]
    res[name]["share_of_failures_within_1pct_of_1"] = float(((bad - 1).abs() < 0.01).float().mean()) if bad.numel() else 0.0
    res[name]["share_of_args_equal_to_clamp"] = float((arg == arg.min()).float().mean())
print(json.dumps(res, indent=0))
