"""The enhancer's loop body on decoded frames: ONE uint8 -> uint8 kernel (vrg_sharpen_grain_u8) against the converter -> fused fp32
kernel -> converter route, device-resident (HIP events) and host-fed (numpy frames in, numpy frames out), interleaved rounds, median
and spread.    python tools/bench_u8_enhancer.py [--frames 8] [--rounds 7] [--json out.json]     (PROBE_PMC=1: one launch of each)"""
import argparse, json, os, statistics, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, VRGDG_LUTVideoTools as LVT, VRGDG_StandaloneVideoEnhancerNodes as enh

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--json", default="")
ap.add_argument("--size", default="2160x3840", help="HxW (round 5: any -- 768x1366, 480x854, 1080x1918 ... run k_sharpen_grain_u8_any)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
F = a.frames
H, W = (int(v) for v in a.size.split("x"))
px = F * H * W
g = torch.Generator().manual_seed(11)
host = torch.randint(0, 256, (F, H, W, 3), generator=g, dtype=torch.uint8)
x = host.to(dev)
st = {"sharpen_enabled": True, "sharpen_strength": 0.5, "grain_enabled": True, "grain_intensity": 0.04, "saturation_mix": 0.5, "seed": 42, "use_gpu": True}


def route():
    return ops.f32_to_frames_u8(ops.sharpen_then_seeded_grain(ops.frames_u8_to_f32(x), 0.5, True, 0.04, 0.5, 42, 0))


def fused():
    return ops.sharpen_then_seeded_grain(x, 0.5, True, 0.04, 0.5, 42, 0)


if os.environ.get("PROBE_PMC"):
    route(); fused(); torch.cuda.synchronize(); sys.exit(0)
assert torch.equal(route(), fused())
frames = [host[i].numpy() for i in range(F)]


def loop_old():      # round 3's loop: converter kernel, fused fp32 kernel, converter kernel
    t = LVT._frames_to_tensor(frames)
    y, _ = enh._process_with_retry(t, st, 0)
    return LVT._tensor_to_frames(y)


def loop_new():
    t = enh._frames_to_tensor(frames)
    y, _ = enh._process_with_retry(t, st, 0)
    return enh._tensor_to_frames(y)


def ev(fn):
    e0, e1 = ops.HipEvent(), ops.HipEvent()
    e0.record(); r = fn(); e1.record()
    dt = e0.elapsed_ms(e1); del r
    return dt


def wall(fn):
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3; del r
    return dt


cases = [("device: converter -> fp32 fused -> converter (three kernels)", route, ev), ("device: vrg_sharpen_grain_u8 (one kernel)", fused, ev),
         ("host-fed loop body, three kernels", loop_old, wall), ("host-fed loop body, one kernel", loop_new, wall)]
for _, fn, t in cases:
    t(fn)
ts = [[] for _ in cases]
for _ in range(a.rounds):
    for i, (_, fn, t) in enumerate(cases):
        ts[i].append(t(fn))
a0 = np.stack(loop_old()); a1 = np.stack(loop_new())
rows = []
for (name, _, _), v in zip(cases, ts):
    med = statistics.median(v)
    rows.append({"case": name, "frames": F, "size": a.size, "ms_median": round(med, 4), "ms_min": round(min(v), 4), "ms_max": round(max(v), 4),
                 "spread_pct": round(100 * (max(v) - min(v)) / med, 2), "Mpix_s": round(px / med / 1e3, 1)})
    print("[u8]", rows[-1], flush=True)
res = {"rows": rows, "speedup_device": round(rows[0]["ms_median"] / rows[1]["ms_median"], 3), "speedup_host_fed": round(rows[2]["ms_median"] / rows[3]["ms_median"], 3),
       "bytes_equal": bool(np.array_equal(a0, a1))}
print("[u8]", res["speedup_device"], res["speedup_host_fed"], res["bytes_equal"])
if a.json:
    with open(a.json, "w") as fh:
        json.dump(res, fh, indent=1)
