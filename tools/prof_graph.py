"""The four-node graph (grain -> LUT -> colour match -> unsharp through NODE_CLASS_MAPPINGS, CPU tensors in and out) ONCE, for a
`rocprofv3 --kernel-trace --stats` record of what it launches: with the deferred graph fusion one k_produce_lab + the statistics kernels +
one k_apply_march per piece; with VRGDG_DEFER_GRAPH=0 the four nodes' own kernels.   python tools/prof_graph.py [--frames 4]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import comfyui_vrgamedevgirl_amd as pack
from comfyui_vrgamedevgirl_amd import _devices
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4)
a = ap.parse_args()
N = pack.NODE_CLASS_MAPPINGS
x = torch.rand((a.frames, 2160, 3840, 3), generator=torch.Generator().manual_seed(3))
ref = x[:1].clone()
t = getattr(N["FastFilmGrain"](), N["FastFilmGrain"].FUNCTION)(x, 0.04, 0.5, 4)[0]
t = getattr(N["VRGDG_LUTS"](), N["VRGDG_LUTS"].FUNCTION)(t, "AMD_TealOrange_33.cube", "auto", 10.0)[0]
t = getattr(N["ColorMatchToReference"](), N["ColorMatchToReference"].FUNCTION)(t, ref, 1.0, 1)[0]
t = getattr(N["FastUnsharpSharpen"](), N["FastUnsharpSharpen"].FUNCTION)(t, 0.5, False)[0]
print("pending before the first host use:", _devices.pending_of(t) is not None, "fused nodes:", _devices._LAZY.fused)
print("checksum", float(t.double().sum()))
torch.cuda.synchronize()
