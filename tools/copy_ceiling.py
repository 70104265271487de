"""The streaming ceiling of the box: vrg_debug_copy_f32 (float4 copy, plain / non-temporal / 4 per thread, read only, write only) and
torch's own copy on 16 and 256 x 4K fp32 frames, HIP-event timed; every streaming kernel of the library next to it.
    python tools/copy_ceiling.py [--out gpurun_out/copy_ceiling.json]
(under rocprofv3 --kernel-trace --stats the same launches appear as k_dbg_copy<MODE>)"""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip, ops, cube, VRGDG_IV_Adjustments as iv

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/copy_ceiling.json")
ap.add_argument("--iters", type=int, default=7)
ap.add_argument("--frames", default="16,256")
args = ap.parse_args()
dev = torch.device("cuda", 0)
H, W = 2160, 3840


def timed(fn, iters=args.iters):
    ts = []
    for it in range(iters + 2):
        a, b = ops.HipEvent(), ops.HipEvent()
        a.record(); fn(); b.record()
        t = a.elapsed_ms(b)
        if it >= 2:
            ts.append(t)
    return statistics.median(ts), min(ts)


res = {"device": torch.cuda.get_device_name(0), "rows": []}
for F in [int(v) for v in args.frames.split(",")]:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.empty((F, H, W, 3), device=dev)
    for i in range(0, F, 16):
        x[i:i + 16] = torch.rand((min(16, F - i), H, W, 3), generator=g, device=dev)
    y = torch.empty_like(x)
    nbytes = x.numel() * 4
    mpx = F * H * W / 1e6

    def row(name, fn, bytes_moved):
        med, best = timed(fn)
        r = {"frames": F, "kernel": name, "ms": round(med, 4), "ms_best": round(best, 4), "GB_s": round(bytes_moved / med / 1e6, 1),
             "frac_of_8TBs": round(bytes_moved / med / 1e6 / 8000.0, 4), "Mpix_s": round(mpx / med * 1e3, 0)}
        res["rows"].append(r)
        print("[copy]", r, flush=True)

    lib = _hip.lib()
    for mode, name, moved in ((0, "copy float4 plain", 2), (1, "copy float4 nt", 2), (2, "copy 4 x float4 nt", 2), (3, "read only", 1), (4, "write only", 1)):
        row(f"vrg_debug_copy_f32 mode {mode}: {name}",
            lambda m=mode: _hip.check(lib.vrg_debug_copy_f32(_hip.ptr(x), _hip.ptr(y), x.numel(), m, _hip.current_stream()), "copy"), moved * nbytes)
    row("torch copy_", lambda: y.copy_(x), 2 * nbytes)
    gen = torch.Generator(device=dev).manual_seed(5)
    lut17 = None
    row("k_grain (chunk 4)", lambda: ops.film_grain(x, 0.04, 0.5, chunk_frames=4, generator=gen), 2 * nbytes)
    for op in ("unsharp", "laplacian", "sobel"):
        row(f"stencil {op} replicate", lambda op=op: ops.stencil3x3(x, op, 0.5, False), 2 * nbytes)
    row("stencil unsharp zero border", lambda: ops.stencil3x3(x, "unsharp", 0.5, True), 2 * nbytes)
    lab = x * 100.0 - 30.0
    row("k_tstats_frame (torch-order statistics of a Lab image)", lambda: ops.lab_stats_device(lab, 1), nbytes)
    del lab
    del x, y
    torch.cuda.empty_cache()
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
with open(args.out, "w") as fh:
    json.dump(res, fh, indent=1)
