"""Per-piece timeline of a host-fed node call (pageable input): when the host thread entered / left each upload call, and when the upload,
the kernels and the download of each piece finished on the GPU (ms from the first upload call).
    python tools/host_fed_timeline.py [--frames 16] [--nodes grain,colormatch]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import nodes, _devices
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--nodes", default="grain,colormatch")
ap.add_argument("--json", default="")
a = ap.parse_args()
x = torch.rand((a.frames, 2160, 3840, 3), generator=torch.Generator().manual_seed(3))
ref = x[:1].clone()
fns = {"grain": lambda: nodes.FastFilmGrain().apply_grain(x, 0.04, 0.5, 4)[0],
       "colormatch": lambda: nodes.ColorMatchToReference().match_color(x, ref, 1.0, 1)[0]}
out = {}
for name in a.nodes.split(","):
    fns[name](); fns[name](); torch.cuda.synchronize()
    _devices._TRACE = []
    start = torch.cuda.Event(enable_timing=True)
    t_call = time.perf_counter(); start.record(); r = fns[name](); torch.cuda.synchronize(); t_end = time.perf_counter()
    tr, _devices._TRACE = _devices._TRACE, None
    rows = []
    for i, t0, t1, up, ran, done in tr:
        rows.append({"piece": i, "host_enter_upload_ms": round((t0 - t_call) * 1e3, 2), "host_leave_upload_ms": round((t1 - t_call) * 1e3, 2),
                     "upload_done_ms": round(start.elapsed_time(up), 2), "kernels_done_ms": round(start.elapsed_time(ran), 2),
                     "download_done_ms": round(start.elapsed_time(done), 2)})
        print("[tl]", name, rows[-1], flush=True)
    print("[tl]", name, "call", round((t_end - t_call) * 1e3, 2), "ms", flush=True)
    out[name] = {"call_ms": round((t_end - t_call) * 1e3, 2), "pieces": rows}
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
