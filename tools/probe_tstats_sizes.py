"""k_tstats_frame (torch's mean / std replayed over a stored Lab image, batch_size 1) on buffers of 64 ... 512 4K frames: is its time per frame a
function of the buffer size (address-translation reach), of what else is resident, or of what ran before it?  VERDICT round 5, weak 6.

    python tools/probe_tstats_sizes.py [--json gpurun_out/tstats_sizes.json] [--only 256,512]      (--only: for a rocprofv3 --pmc pass)
"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip, ops
ap = argparse.ArgumentParser()
ap.add_argument("--json", default="")
ap.add_argument("--only", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
H, W = 2160, 3840
sizes = [int(v) for v in a.only.split(",")] if a.only else [64, 128, 256, 384, 512]
big = torch.empty((max(sizes), H, W, 3), dtype=torch.float32, device=dev)
for i in range(0, max(sizes), 16):
    big[i:i + 16].copy_(torch.rand((min(16, max(sizes) - i), H, W, 3), device=dev) * 100.0 - 20.0)
rows = []


def timed(lab, reps=5):
    ms_out = torch.empty((lab.shape[0], 3, 2), dtype=torch.float32, device=dev)
    ops.lab_stats_device(lab, 1, out=ms_out)
    ts = []
    for _ in range(reps):
        e0, e1 = ops.HipEvent(), ops.HipEvent()
        e0.record(); ops.lab_stats_device(lab, 1, out=ms_out); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_ms(e1))
    ts.sort()
    return ts[len(ts) // 2]


for n in sizes:
    ms = timed(big[:n])
    rows.append({"case": "first n frames of one 512-frame buffer", "frames": n, "ms": round(ms, 3), "us_per_frame": round(1e3 * ms / n, 2),
                 "TB_s": round(n * H * W * 12 / ms / 1e9, 3)})
    print(json.dumps(rows[-1]), flush=True)
if not a.only:
    # the same 256 frames at the far end of the buffer (other addresses, same size)
    ms = timed(big[256:512])
    rows.append({"case": "frames 256..511 of the buffer", "frames": 256, "ms": round(ms, 3), "TB_s": round(256 * H * W * 12 / ms / 1e9, 3)})
    print(json.dumps(rows[-1]), flush=True)
    # 512 frames as two launches of 256 (what a caller could do if the per-launch size matters)
    e0, e1 = ops.HipEvent(), ops.HipEvent()
    o = torch.empty((512, 3, 2), dtype=torch.float32, device=dev)
    ops.lab_stats_device(big[:256], 1, out=o[:256]); ops.lab_stats_device(big[256:], 1, out=o[256:])
    e0.record(); ops.lab_stats_device(big[:256], 1, out=o[:256]); ops.lab_stats_device(big[256:], 1, out=o[256:]); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_ms(e1)
    rows.append({"case": "512 frames as two launches of 256", "frames": 512, "ms": round(ms, 3), "TB_s": round(512 * H * W * 12 / ms / 1e9, 3)})
    print(json.dumps(rows[-1]), flush=True)
    # with 102 GB more resident (the colour-match leg holds input + output + Lab image)
    other = torch.empty((2, 512, H, W, 3), dtype=torch.float32, device=dev)
    other[0, :8].fill_(1.0)
    for n in (256, 512):
        ms = timed(big[:n])
        rows.append({"case": "with 102 GB more allocated", "frames": n, "ms": round(ms, 3), "TB_s": round(n * H * W * 12 / ms / 1e9, 3)})
        print(json.dumps(rows[-1]), flush=True)
    del other
    # right behind a pass that WROTE the buffer (the chain's situation: pass 1 stores the Lab image, then the reductions read it)
    for n in (256, 512):
        ts = []
        o = torch.empty((n, 3, 2), dtype=torch.float32, device=dev)
        for _ in range(3):
            big[:n].mul_(1.0)
            e0, e1 = ops.HipEvent(), ops.HipEvent()
            e0.record(); ops.lab_stats_device(big[:n], 1, out=o); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_ms(e1))
        ts.sort()
        rows.append({"case": "right behind a kernel that rewrote the buffer", "frames": n, "ms": round(ts[1], 3), "TB_s": round(n * H * W * 12 / ts[1] / 1e9, 3)})
        print(json.dumps(rows[-1]), flush=True)
if a.json:
    os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
    json.dump({"device": torch.cuda.get_device_name(0), "rows": rows}, open(a.json, "w"), indent=1)
