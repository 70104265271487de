"""Which pixels send dev_pow_ziv to the transcription on the benchmark's frames?  The exhaustive sweeps measure the fallback rate over every
fp32 of the domain (0.02-0.03 % of the lanes); a frame is not that distribution -- clamped pixels (exact 0 and 1 after grain's and the cube's
clamp) are a few per cent of it and each is ONE argument whose rounding test passes or fails for all of them.
    python tools/probe_ziv_fallback_on_frames.py [--dist uniform|video] [--frames 4]
Prints per call site (sRGB -> linear, y = 2.4; the Lab cube root, y = 1/3) the share of lanes and of 64-lane groups with a failing test, and the
most frequent failing arguments."""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import bench
from comfyui_vrgamedevgirl_amd import ops, cube, _hip, VRGDG_IV_Adjustments as iv

ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="uniform")
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
x = bench.make_frames(a.frames, 2160, 3840, dev, 1234, a.dist)
lut = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), dev)
gen = torch.Generator(device=dev); gen.manual_seed(5)
v = ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0)), generator=gen)


def flags(arg, y):
    out = torch.empty_like(arg)
    _hip.check(_hip.lib().vrg_debug_cm_math(_hip.ptr(arg), _hip.ptr(out), arg.numel(), 15, float(np.float32(y)), _hip.current_stream()), "dbg")
    return out


def report(name, arg, y):
    arg = arg.contiguous()
    f = flags(arg.view(-1), y).view(arg.shape)
    lane = float(f.mean())
    per_px = f.amax(dim=-1).view(-1)                       # a pixel with any failing channel
    n64 = per_px.numel() // 64 * 64
    grp = float(per_px[:n64].view(-1, 64).amax(dim=1).mean())
    bad = arg[f > 0]
    vals, counts = torch.unique(bad, return_counts=True)
    top = torch.argsort(counts, descending=True)[:8]
    row = {"site": name, "lanes_failing": lane, "pixels_with_a_failing_channel": float(per_px.mean()), "groups_of_64_pixels_with_one": grp,
           "distinct_failing_arguments": int(vals.numel()),
           "top": [(float(vals[i]), hex(int(vals[i].view(torch.int32)) & 0xffffffff), int(counts[i]), round(int(counts[i]) / max(1, bad.numel()), 4)) for i in top]}
    print("[ziv]", json.dumps(row), flush=True)
    return row


rows = []
q = torch.clamp_min((v + 0.055) / 1.055, 0.0625)
rows.append(report("srgb_to_linear y=2.4", q, 2.4))
lin = torch.where(v > 0.04045, torch.pow(q, 2.4), v / 12.92)
M = torch.tensor([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], device=dev)
r, g, b = lin[..., 0], lin[..., 1], lin[..., 2]
xyz = torch.stack([M[i, 0] * r + M[i, 1] * g + M[i, 2] * b for i in range(3)], dim=-1)
t = xyz / torch.tensor([0.95047, 1.0, 1.08883], device=dev)
rows.append(report("lab cube root y=1/3", torch.clamp_min(t, 0.008856), 1 / 3.0))
share = {"exact_zero": float((v == 0).float().mean()), "exact_one": float((v == 1).float().mean())}
print("[ziv] clamped values of the pre-stage output:", share)
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"dist": a.dist, "frames": a.frames, "rows": rows, "clamped": share}, open(a.out, "w"), indent=1)
