"""vrg_host_copy pageable -> page-locked on this host, by thread count (no GPU work besides the page-locking).   python tools/host_copy_rate.py"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import _hip
lib = _hip.load_library()
n = 199065600            # two 4K fp32 frames
src = torch.rand(n // 4)
dst = torch.empty(n // 4, pin_memory=True)
print("[copy] cores", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
for th in (1, 2, 4, 6, 8, 12, 16, 24, 32):
    ts = []
    for _ in range(6):
        t = time.perf_counter(); lib.vrg_host_copy(dst.data_ptr(), src.data_ptr(), n, th); ts.append(time.perf_counter() - t)
    print("[copy] threads", th, "GB/s median", round(n / statistics.median(ts[1:]) / 1e9, 1), "min", round(n / max(ts[1:]) / 1e9, 1), "max", round(n / min(ts[1:]) / 1e9, 1), flush=True)
ts = []
for _ in range(6):
    t = time.perf_counter(); dst.copy_(src); ts.append(time.perf_counter() - t)
print("[copy] torch copy_ GB/s median", round(n / statistics.median(ts[1:]) / 1e9, 1), flush=True)
