"""Per-call timing of the four-node graph with the lazy download: where does an occasional slow repetition spend its time?
    python tools/diag_lazy_graph.py [--frames 16] [--reps 14]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import nodes, _devices, VRGDG_IV_Adjustments as iv
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--reps", type=int, default=14)
ap.add_argument("--inside", action="store_true", help="also time the host-side steps inside a node call (result buffer, poison, content stamp) and Python's collector")
a = ap.parse_args()
spent = {}
if a.inside:
    import gc

    def timed(name, fn):
        def wrapped(*args, **kw):
            s = time.perf_counter()
            try:
                return fn(*args, **kw)
            finally:
                spent[name] = spent.get(name, 0.0) + time.perf_counter() - s
        return wrapped
    for name in ("_poison", "_result_buffer", "_content_stamp"):
        setattr(_devices, name, timed(name, getattr(_devices, name)))
    gc_t = {}

    def on_gc(phase, info):
        if phase == "start":
            gc_t["s"] = time.perf_counter()
        else:
            spent[f"gc gen{info['generation']}"] = spent.get(f"gc gen{info['generation']}", 0.0) + time.perf_counter() - gc_t["s"]
    gc.callbacks.append(on_gc)
g = torch.Generator().manual_seed(3)
x = torch.rand((a.frames, 2160, 3840, 3), generator=g)
ref = x[:1].clone()
calls = [("grain", lambda t: nodes.FastFilmGrain().apply_grain(t, 0.04, 0.5, 4)[0]),
         ("lut", lambda t: iv.VRGDG_LUTS().apply_lut(t, "AMD_TealOrange_33.cube", "auto", 10.0)[0]),
         ("match", lambda t: nodes.ColorMatchToReference().match_color(t, ref, 1.0, 1)[0]),
         ("unsharp", lambda t: nodes.FastUnsharpSharpen().apply_unsharp(t, 0.5, False)[0])]
for rep in range(a.reps):
    t = x
    marks = []
    t0 = time.perf_counter()
    for name, fn in calls:
        s = time.perf_counter()
        t = fn(t)
        marks.append((name, time.perf_counter() - s))
    s = time.perf_counter()
    _devices.materialise(t)
    marks.append(("download", time.perf_counter() - s))
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    del t
    print(f"[diag] rep {rep}: total {total * 1e3:7.1f} ms  " + "  ".join(f"{n} {d * 1e3:6.1f}" for n, d in marks) +
          ("   inside: " + "  ".join(f"{k} {v * 1e3:.1f}" for k, v in sorted(spent.items())) if spent else ""), flush=True)
    spent.clear()
