"""Where the reference frame's statistics spend their time: Lab conversion / torch-order statistics / both, per library build, interleaved.
    python tools/bench_refstats.py --libs a=..,b=.. [--rounds 9]"""
import argparse, json, os, statistics, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, _hip
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--libs", required=True)
ap.add_argument("--rounds", type=int, default=9)
a = ap.parse_args()
libs = [(i.split("=", 1)[0], _hip.load_library(os.path.abspath(i.split("=", 1)[1]))) for i in a.libs.split(",")]
dev = torch.device("cuda", 0)
ref = bench.make_frames(1, 2160, 3840, dev, 4321, "uniform")
_hip._lib = libs[0][1]
lab = torch.empty_like(ref)
d = ops._chain_desc(ops.ChainSpec(), None, [], ref)
def conv():
    _hip.check(_hip.lib().vrg_chain_stats_lab_f32(_hip.ptr(ref), _hip.ptr(lab), 1, 2160, 3840, C.byref(d), None, None, _hip.current_stream()), "lab")
conv(); torch.cuda.synchronize()
synth = torch.rand_like(ref) * 100 - 30
cases = {"lab conversion": conv, "stats of the converted frame": lambda: ops.lab_stats_device(lab, 1), "stats of a synthetic frame": lambda: ops.lab_stats_device(synth, 1),
         "stats of the converted frame, latency form": lambda: ops.lab_stats_device(lab, 1, latency_form=True),
         "reference_stats (both)": lambda: ops.reference_stats(ref), "reference_stats (both), small step": lambda: ops.reference_stats(ref, step_frames=4)}
ts = {}
for rnd in range(a.rounds + 1):
    for name, fn in cases.items():
        for n, lib in libs:
            _hip._lib = lib
            e0, e1 = ops.HipEvent(), ops.HipEvent()
            e0.record(); fn(); e1.record()
            t = e0.elapsed_ms(e1)
            torch.cuda.synchronize()
            if rnd:
                ts.setdefault((name, n), []).append(t)
for (name, n), v in ts.items():
    print("[ref]", name, n, round(statistics.median(v), 4), "min", round(min(v), 4), "max", round(max(v), 4), flush=True)
