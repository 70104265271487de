"""Round 5, VERDICT item 3: the one overlap of the headline step that is physically complementary -- the statistics reductions
(k_tstats_frame: HBM-bound, VALU idle) of frame range i under pass 1 (k_produce_lab: VALU-bound, HBM at 10 %) of range i + 1 -- taken once
with the round-4 kernels.  Schedules of one step over F 4K frames (grain -> LUT 33^3 -> colour match -> unsharp, device statistics):
  seq      pass 1 (all) -> reductions (all) -> pass 2 (all): what ops.fused_chain runs
  ovl R    the frames cut into R chunk-aligned ranges; pass 1 of range i on the main stream, its reductions on a second stream behind an
           event, pass 1 of range i + 1 meanwhile; pass 2 (all) behind the last reduction
  ovl R hi the same with the reductions' stream at high priority
Every schedule leaves the same output bits (checked).  Wall time per step = HIP events on the main stream around the whole step (the main stream
waits for the side stream before pass 2), median of `rounds` interleaved rounds.
    python tools/exp_tstats_overlap.py [--frames 256] [--rounds 7] [--json out.json] [--once SCHEDULE]     (--once: one step, for rocprofv3 --kernel-trace)
"""
import argparse, ctypes as C, json, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, _hip, VRGDG_IV_Adjustments as iv

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--json", default="")
ap.add_argument("--once", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
F, H, W, chunk = a.frames, 2160, 3840, 4
fe = H * W * 3
g = torch.Generator(device=dev).manual_seed(3)
x = torch.empty((F, H, W, 3), device=dev)
for i in range(0, F, 16):
    x[i:i + 16].copy_(torch.rand((min(16, F - i), H, W, 3), generator=g, device=dev))
out = torch.empty_like(x)
lab = torch.empty_like(x)
ms = torch.empty((F, 3, 2), dtype=torch.float32, device=dev)
lut = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), dev)
ref_ms = ops.reference_stats(x[:1])
gen = torch.Generator(device=dev).manual_seed(42)
stream = ops.rng.reserve(chunk * fe, F // chunk, dev, gen)
spec = ops.ChainSpec(grain=(0.04, 0.5, chunk), lut=(lut, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False), cm_chunk=1)
lib = _hip.lib()
side = {False: torch.cuda.Stream(device=dev), True: torch.cuda.Stream(device=dev, priority=-1)}


def pass1(f0, nf):
    keep = []
    d = ops._chain_desc(spec, ops.NoisePlan(chunk, stream, chunk0=f0 // chunk), keep, x)
    _hip.check(lib.vrg_chain_stats_lab_f32(C.c_void_p(x.data_ptr() + f0 * fe * 4), C.c_void_p(lab.data_ptr() + f0 * fe * 4), nf, H, W, C.byref(d), None, None,
                                          _hip.current_stream()), "pass 1")


def pass2():
    keep = []
    d = ops._chain_desc(spec, ops.NoisePlan(chunk, stream, chunk0=0), keep, x)
    d.stages = (d.stages & _hip.STAGE_SHARPEN) | _hip.STAGE_COLORMATCH | _hip.STAGE_FROM_LAB
    d.img_ms = ms.data_ptr()
    _hip.check(lib.vrg_fused_chain_f32(C.c_void_p(lab.data_ptr()), C.c_void_p(out.data_ptr()), F, H, W, C.byref(d), _hip.current_stream()), "pass 2")


def step(schedule):
    main = torch.cuda.current_stream()
    if schedule == "seq":
        pass1(0, F)
        ops.lab_stats_device(lab, 1, out=ms)
    else:
        R, hi = int(schedule.split()[1]), schedule.endswith("hi")
        per = max(chunk, (F // R) // chunk * chunk)
        s = side[hi]
        s.wait_stream(main)
        for f0 in range(0, F, per):
            nf = min(per, F - f0)
            pass1(f0, nf)
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(s):
                s.wait_event(ev)
                ops.lab_stats_device(lab[f0:f0 + nf], 1, out=ms[f0:f0 + nf])
        main.wait_stream(s)
    pass2()


schedules = ["seq", "ovl 2", "ovl 4", "ovl 8", "ovl 16", "ovl 4 hi", "ovl 8 hi"]
if a.once:
    ops.toolchain_selfcheck(dev)
    step(a.once); step(a.once)
    torch.cuda.synchronize()
    print("done", a.once)
    sys.exit(0)
digest = {}
for sc in schedules:        # warm-up + the bit comparison
    step(sc)
    torch.cuda.synchronize()
    d = out.view(torch.int32)
    digest[sc] = (int(d.sum(dtype=torch.int64)), int((d[:8].to(torch.int64) * 31 % 1000003).sum()))
times = {sc: [] for sc in schedules}
for rnd in range(a.rounds):
    order = schedules[rnd % len(schedules):] + schedules[:rnd % len(schedules)]
    step(order[-1]); torch.cuda.synchronize()
    for sc in order:
        e0, e1 = ops.HipEvent(), ops.HipEvent()
        e0.record(); step(sc); e1.record()
        torch.cuda.synchronize()
        times[sc].append(e0.elapsed_ms(e1))
rows = []
base = statistics.median(times["seq"])
for sc in schedules:
    med = statistics.median(times[sc])
    rows.append({"schedule": sc, "frames": F, "ms_median": round(med, 3), "ms_min": round(min(times[sc]), 3), "ms_max": round(max(times[sc]), 3),
                 "spread_pct": round(100 * (max(times[sc]) - min(times[sc])) / med, 2), "vs_seq_pct": round(100 * (med - base) / base, 2),
                 "same_bits_as_seq": digest[sc] == digest["seq"], "mpix_s": round(F * H * W / med / 1e3, 1)})
    print("[ovl]", rows[-1], flush=True)
if a.json:
    os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
    with open(a.json, "w") as fh:
        json.dump({"device": torch.cuda.get_device_name(0), "rounds": a.rounds, "rows": rows}, fh, indent=1)
