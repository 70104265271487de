"""GPU-box diagnostics + per-kernel micro-benchmarks in one go (writes gpurun_out/diag.json and prints a table).

    python tools/gpu_diag.py [--frames 32] [--quick]

Each section is independent (a failure is recorded, not fatal) so that one gpurun call yields as much
information as possible: device properties, noise-stream agreement with torch.randn (with mismatch anatomy),
and HIP-event timings of every stand-alone kernel and fused chain at 1080p / 4K with achieved algorithmic GB/s.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

OUT_DIR = os.path.join(ROOT, "gpurun_out")
RESULT = {"sections": {}}


def section(name):
    def deco(fn):
        def run(*a, **k):
            t0 = time.time()
            try:
                RESULT["sections"][name] = {"ok": True, "data": fn(*a, **k)}
            except Exception as exc:
                RESULT["sections"][name] = {"ok": False, "error": f"{type(exc).__name__}: {exc}", "trace": traceback.format_exc()[-2000:]}
                print(f"[diag] section {name} FAILED: {exc}", flush=True)
            RESULT["sections"][name]["seconds"] = round(time.time() - t0, 2)
        return run
    return deco


@section("device")
def device_info():
    p = torch.cuda.get_device_properties(0)
    d = {"name": p.name, "cus": p.multi_processor_count, "max_threads_per_cu": p.max_threads_per_multi_processor,
         "total_mem_gb": round(p.total_memory / 2 ** 30, 1), "torch": torch.__version__, "hip": torch.version.hip,
         "gcn_arch": getattr(p, "gcnArchName", "?"), "host_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads()}
    print("[diag] device:", d, flush=True)
    return d


@section("noise")
def noise_check(ops):
    dev = torch.device("cuda", 0)
    res = {}
    for frames, fe, chunk in ((1, 105, 1), (2, 3 * 64 * 64, 1), (8, 3 * 512 * 512, 4)):
        torch.manual_seed(1)
        want = torch.cat([torch.randn(chunk * fe, device=dev) for _ in range(frames // chunk)])
        torch.manual_seed(1)
        main, tail, n_full = ops.plan_noise(frames, fe, chunk, dev)
        got = ops.torch_stream_noise(frames, fe, main, dev).flatten()
        neq = got != want
        info = {"mismatch": int(neq.sum()), "numel": got.numel(), "max_abs": float((got - want).abs().max())}
        if info["mismatch"]:
            idx = torch.nonzero(neq).flatten()[:8].tolist()
            info["first"] = [(i, float(got[i]), float(want[i])) for i in idx]
            # anatomy: are the values close (Box-Muller rounding) or unrelated (mapping bug)?
            info["close_frac"] = float(((got - want).abs() < 1e-4).float().mean())
        res[f"{frames}x{fe}/{chunk}"] = info
        print("[diag] noise", frames, fe, chunk, info, flush=True)
    return res


def time_it(fn, iters, ops):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = ops.HipEvent(), ops.HipEvent()
        a.record()
        fn()
        b.record()
        ts.append(a.elapsed_ms(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


@section("valu_rate")
def valu_rate(ops):
    """Issue-rate probe: lane-instructions per second for a few instruction kinds (the VALU roofline)."""
    from comfyui_vrgamedevgirl_amd import _hip
    dev = torch.device("cuda", 0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    names = ["v_fma_f32", "v_mad_u64_u32", "v_log_f32", "v_pk_fma_f32", "v_xor_b32", "sqrt/sin/cos/rcp", "v_cmp+v_cndmask", "v_mul/fma_f64"]
    rows = []
    for waves_per_simd in (8, 2):
        blocks = cus * waves_per_simd           # 4 waves per block -> waves_per_simd waves on each of the 4 SIMDs of every CU
        out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
        iters = 4096 // waves_per_simd
        for mode, name in enumerate(names):
            fn = lambda m=mode: _hip.check(_hip.lib().vrg_debug_valu_rate(_hip.ptr(out), blocks, iters, m, _hip.current_stream()), "valu")
            med, best = time_it(fn, 3, ops)
            lane_instr = blocks * 256 * iters * 64
            row = {"instr": name, "waves_per_simd": waves_per_simd, "ms": round(best, 4), "tera_lane_instr_s": round(lane_instr / best / 1e9, 2),
                   "cycles_per_wave_instr_at_2p4GHz": round(best * 1e-3 * 2.4e9 / (iters * 64 * waves_per_simd), 2)}
            rows.append(row)
            print("[diag]", row, flush=True)
    return rows


@section("host_fed")
def host_fed(ops):
    """ComfyUI-realistic rates: CPU tensors in, CPU tensors out (PCIe inclusive), sequential vs pipelined staging."""
    import time as _t
    import numpy as np
    from comfyui_vrgamedevgirl_amd import nodes, _devices, VRGDG_LUTVideoTools as LVT, VRGDG_IV_Adjustments as iv
    dev = torch.device("cuda", 0)
    rows = []
    F, H, W = 16, 2160, 3840
    x = torch.rand(F, H, W, 3)
    px = F * H * W
    nbytes = x.numel() * 4

    def wall(fn, reps=3):
        r = fn(); torch.cuda.synchronize(); del r
        best = 1e9
        for _ in range(reps):
            t0 = _t.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, _t.perf_counter() - t0); del r
        return best

    pinned = torch.empty_like(x, pin_memory=True)
    g = torch.empty_like(x, device=dev)
    t0 = _t.perf_counter(); big = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True); t_pin = _t.perf_counter() - t0
    del big
    raw = {
        "pageable H2D GB/s": nbytes / wall(lambda: g.copy_(x)) / 1e9,
        "pinned H2D GB/s": nbytes / wall(lambda: g.copy_(pinned, non_blocking=True)) / 1e9,
        "pageable D2H GB/s": nbytes / wall(lambda: x.copy_(g)) / 1e9,
        "pinned D2H GB/s": nbytes / wall(lambda: pinned.copy_(g, non_blocking=True)) / 1e9,
        "host copy_ pageable->pinned GB/s": nbytes / wall(lambda: pinned.copy_(x)) / 1e9,
        "pin 1 GiB seconds": t_pin, "torch threads": torch.get_num_threads(),
    }
    print("[diag] host raw", {k: round(v, 2) for k, v in raw.items()}, flush=True)
    rows.append({"raw": {k: round(v, 3) for k, v in raw.items()}})
    del pinned, g
    lut_name = "AMD_TealOrange_33.cube"
    cases = [
        ("FastUnsharpSharpen", lambda: nodes.FastUnsharpSharpen().apply_unsharp(x, 0.5, False)),
        ("FastFilmGrain bs=4", lambda: nodes.FastFilmGrain().apply_grain(x, 0.04, 0.5, 4)),
        ("ColorMatchToReference", lambda: nodes.ColorMatchToReference().match_color(x, x[:1], 1.0, 1)),
    ]
    torch.cuda.synchronize()
    nodes.PIPELINED = True
    t0 = _t.perf_counter(); r = cases[0][1](); torch.cuda.synchronize(); t_first = _t.perf_counter() - t0; del r
    row = {"node": "FastUnsharpSharpen FIRST call (page-locks the result)", "seconds": round(t_first, 4), "mpix_s": round(px / t_first / 1e6, 1)}
    rows.append(row)
    print("[diag]", row, flush=True)
    for name, fn in cases:
        for mode in (False, True):
            nodes.PIPELINED = mode
            t = wall(fn, reps=2)
            row = {"node": name, "pipelined": mode, "frames": F, "seconds": round(t, 4), "mpix_s": round(px / t / 1e6, 1),
                   "pcie_gbs_each_way": round(nbytes / t / 1e9, 2)}
            rows.append(row)
            print("[diag]", row, flush=True)
    nodes.PIPELINED = True
    xp = x.pin_memory()
    t = wall(lambda: nodes.FastUnsharpSharpen().apply_unsharp(xp, 0.5, False), reps=3)
    row = {"node": "FastUnsharpSharpen, page-locked input (e.g. result of a previous node)", "seconds": round(t, 4), "mpix_s": round(px / t / 1e6, 1),
           "pcie_gbs_each_way": round(nbytes / t / 1e9, 2)}
    rows.append(row)
    print("[diag]", row, flush=True)
    del xp
    # uint8 route batches: 8 decoded 4K frames per call, as the routes do

    class Sink:
        def write(self, frame):
            pass

    batch = [np.random.randint(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(8)]
    t = wall(lambda: LVT._process_video_batch(batch, Sink(), lut_name, 10.0, "cuda"), reps=3)
    row = {"route": "_process_video_batch (uint8 in/out, 8 x 4K)", "seconds": round(t, 4), "mpix_s": round(8 * H * W / t / 1e6, 1)}
    rows.append(row)
    print("[diag]", row, flush=True)
    t = wall(lambda: LVT._tensor_to_frames(LVT._apply_lut_tensor(LVT._frames_to_tensor(batch), lut_name, 10.0, "cuda")), reps=3)
    row = {"route": "convert -> _apply_lut_tensor -> convert (8 x 4K)", "seconds": round(t, 4), "mpix_s": round(8 * H * W / t / 1e6, 1)}
    rows.append(row)
    print("[diag]", row, flush=True)
    return rows


@section("kernels")
def kernel_bench(ops, frames_4k, iters, match=""):
    from comfyui_vrgamedevgirl_amd import VRGDG_IV_Adjustments as iv
    from comfyui_vrgamedevgirl_amd import cube
    dev = torch.device("cuda", 0)
    rows = []
    lut33 = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube")), dev)
    lut25 = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_WarmFilm_25.cube")), dev)
    lut17 = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_Identity_17.cube")), dev)
    lut21 = ops.upload_lut({"lut": torch.rand(21, 21, 21, 3), "domain_min": torch.zeros(3), "domain_max": torch.ones(3)}, dev)
    for label, H, W, F in (("4K", 2160, 3840, frames_4k), ("1080p", 1080, 1920, frames_4k * 4)):
        g = torch.Generator(device=dev).manual_seed(3)
        x = torch.rand((F, H, W, 3), generator=g, device=dev)
        smooth = (x * 0.05 + 0.5 * (torch.linspace(0, 1, W, device=dev).view(1, 1, W, 1) + torch.linspace(0, 1, H, device=dev).view(1, H, 1, 1)) * 0.9).contiguous()
        out = torch.empty_like(x)
        px = F * H * W
        ref_ms = ops.finalize_stats(ops.lab_stats(x[:1]))
        gen = torch.Generator(device=dev).manual_seed(5)
        img_ms = ops.finalize_stats(ops.lab_stats(x))

        def chain(stages_spec, src=x):
            return lambda: ops.fused_chain(src, stages_spec, generator=gen, out=out)

        cases = [
            ("copy (torch)", 24, lambda: out.copy_(x)),
            ("grain bs=4", 24, lambda: ops.film_grain(x, 0.04, 0.5, chunk_frames=4, generator=gen)),
            ("grain injected", 36, lambda: ops.film_grain_injected(x, x, 0.04, 0.5)),
            ("lut33 uniform", 24, lambda: ops.lut3d(x, lut33, 10.0)),
            ("lut33 smooth", 24, lambda: ops.lut3d(smooth, lut33, 10.0)),
            ("lut25 uniform", 24, lambda: ops.lut3d(x, lut25, 10.0)),
            ("lut17 uniform (LDS-resident table)", 24, lambda: ops.lut3d(x, lut17, 10.0)),
            ("lut21 uniform (LDS-resident table)", 24, lambda: ops.lut3d(x, lut21, 10.0)),
            ("lut17 smooth (LDS-resident table)", 24, lambda: ops.lut3d(smooth, lut17, 10.0)),
            ("lut33 blend 0.5", 24, lambda: ops.lut3d(x, lut33, 5.0)),
            ("unsharp replicate", 24, lambda: ops.stencil3x3(x, "unsharp", 0.5, False)),
            ("unsharp zero", 24, lambda: ops.stencil3x3(x, "unsharp", 0.5, True)),
            ("laplacian", 24, lambda: ops.stencil3x3(x, "laplacian", 0.5, False)),
            ("sobel", 24, lambda: ops.stencil3x3(x, "sobel", 0.5, False)),
            ("lab stats", 12, lambda: ops.lab_stats(x)),
            ("colormatch apply", 24, lambda: ops.colormatch_apply(x, img_ms, ref_ms, 1.0)),
            ("colormatch 2-pass", 36, lambda: ops.color_match(x, None, 1.0, ref_ms=ref_ms)),
            ("fused tile sharpen only", 24, chain(ops.ChainSpec(sharpen=("unsharp", 0.5, False)))),
            ("fused lut+sharpen", 24, chain(ops.ChainSpec(lut=(lut33, 10.0), sharpen=("unsharp", 0.5, False)))),
            ("fused lut+sharpen smooth", 24, chain(ops.ChainSpec(lut=(lut33, 10.0), sharpen=("unsharp", 0.5, False)), smooth)),
            ("fused grain+lut", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0)))),
            ("fused grain+lut+sharpen", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), sharpen=("unsharp", 0.5, False)))),
            ("fused grain+lut17+sharpen (LUT in LDS)", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut17, 10.0), sharpen=("unsharp", 0.5, False)))),
            ("fused lut17+sharpen march (LUT in LDS)", 24, chain(ops.ChainSpec(lut=(lut17, 10.0), sharpen=("unsharp", 0.5, False), variant=2))),
            ("fused lut17+sharpen tile (global LUT)", 24, chain(ops.ChainSpec(lut=(lut17, 10.0), sharpen=("unsharp", 0.5, False), variant=1))),
            ("fused grain+lut17 auto", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut17, 10.0)))),
            ("fused grain+lut17 pointwise (global LUT)", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut17, 10.0), variant=1))),
            ("fused lut17+sharpen auto", 24, chain(ops.ChainSpec(lut=(lut17, 10.0), sharpen=("unsharp", 0.5, False)))),
            ("fused grain+sharpen", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), sharpen=("unsharp", 0.5, False)))),
            ("fused 4-stage", 36, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False)))),
            ("v1 tile sharpen only", 24, chain(ops.ChainSpec(sharpen=("unsharp", 0.5, False), variant=1))),
            ("v1 tile lut+sharpen", 24, chain(ops.ChainSpec(lut=(lut33, 10.0), sharpen=("unsharp", 0.5, False), variant=1))),
            ("v1 pointwise grain+lut", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), variant=1))),
            ("v1 tile grain+lut+sharpen", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), sharpen=("unsharp", 0.5, False), variant=1))),
            ("v1 tile 4-stage", 36, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False), variant=1))),
            ("fused grain+lut+sharpen smooth", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), sharpen=("unsharp", 0.5, False)), smooth)),
            ("v2 march sharpen only", 24, chain(ops.ChainSpec(sharpen=("unsharp", 0.5, False), variant=2))),
            ("v2 march lut+sharpen", 24, chain(ops.ChainSpec(lut=(lut33, 10.0), sharpen=("unsharp", 0.5, False), variant=2))),
            ("v2 march grain+lut", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), variant=2))),
            ("v2 march grain+lut+sharpen", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), sharpen=("unsharp", 0.5, False), variant=2))),
            ("v2 march grain+sharpen", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), sharpen=("unsharp", 0.5, False), variant=2))),
            ("v1 tile grain+sharpen", 24, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), sharpen=("unsharp", 0.5, False), variant=1))),
            ("v2 march 4-stage", 36, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False), variant=2))),
            ("4-stage general-stats-kernel", 36, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False), variant=0x200))),
            ("4-stage stats-unroll2", 36, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False), variant=0x300))),
            ("4-stage smooth", 36, chain(ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False)), smooth)),
        ]
        from comfyui_vrgamedevgirl_amd import VRGDG_LUTVideoTools as LVT
        adj_ws = torch.empty_like(x)

        def adj(settings):
            terms = ops.adjust_terms(LVT._normalize_adjust_settings(settings))
            return lambda: ops.adjust(x, terms, out=out, workspace=adj_ws)

        cases += [
            ("adjust point+tail", 24, adj({"temperature": 20, "exposure": 10, "contrast": 12, "saturation": 8, "highlights": -20,
                                            "shadows": 15, "fade": 10, "vignette": 30})),
            ("adjust clarity", 24, adj({"clarity": 40, "contrast": 12})),
            ("adjust sharpen", 24, adj({"sharpen": 40, "contrast": 12})),
            ("adjust clarity+sharpen", 48, adj({"clarity": 40, "sharpen": 30, "contrast": 12, "vignette": 30})),
        ]
        xu8 = torch.randint(0, 256, x.shape, dtype=torch.uint8, device=dev)
        outu8 = torch.empty_like(xu8)
        adj_pt = ops.adjust_terms(LVT._normalize_adjust_settings({"temperature": 20, "exposure": 10, "contrast": 12, "saturation": 8,
                                                                   "highlights": -20, "shadows": 15, "fade": 10, "vignette": 30}))
        cases += [
            ("u8 -> f32 convert", 15, lambda: ops.frames_u8_to_f32(xu8)),
            ("f32 -> u8 convert", 15, lambda: ops.f32_to_frames_u8(x)),
            ("u8 lut33", 6, lambda: ops.fused_chain(xu8, ops.ChainSpec(lut=(lut33, 10.0)), out=outu8)),
            ("u8 grain", 6, lambda: ops.fused_chain(xu8, ops.ChainSpec(grain=(0.04, 0.5, 4)), generator=gen, out=outu8)),
            ("u8 grain+lut+sharpen", 6, lambda: ops.fused_chain(xu8, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut33, 10.0),
                                                                 sharpen=("unsharp", 0.5, False)), generator=gen, out=outu8)),
            ("u8 unsharp", 6, lambda: ops.fused_chain(xu8, ops.ChainSpec(sharpen=("unsharp", 0.5, False)), out=outu8)),
            ("u8 adjust point+tail", 6, lambda: ops.adjust(xu8, adj_pt, out=outu8)),
        ]
        if label == "4K":
            import ctypes as C
            from comfyui_vrgamedevgirl_amd import _hip
            probe = torch.empty((px,), dtype=torch.float32, device=dev)
            wide = torch.zeros(((lut33.size - 1) ** 2 * lut33.size * 16 + 16,), dtype=torch.float32, device=dev)   # 64-B records for mode 3
            nc = lut33.size - 1
            raw33 = lut33.nodes
            recs = torch.stack([raw33[db:db + nc, dg:dg + nc, :, ch] for ch in range(3) for dg in (0, 1) for db in (0, 1)], dim=-1).reshape(nc * nc, lut33.size, 12).contiguous()
            rec_table = recs.reshape(-1).contiguous()
            wide[:-16].view(-1, 16)[:, :12] = recs.view(-1, 12)
            nc = lut33.size - 1
            cellmajor = torch.zeros((nc * nc, nc, 32), dtype=torch.float32, device=dev)     # 128-B aligned record per cell (mode 4)
            cellmajor[:, :, :12] = recs[:, :-1]
            cellmajor[:, :, 12:24] = recs[:, 1:]
            cellmajor = cellmajor.reshape(-1).contiguous()
            for src_name, src in (("uniform", x), ("smooth", smooth)):
                for mode, tab in ((0, rec_table), (1, rec_table), (2, rec_table), (3, wide), (4, cellmajor)):
                    cases.append((f"probe lut fetch mode {mode} {src_name}", 16, (lambda m=mode, t=tab, s_=src: _hip.check(_hip.lib().vrg_debug_lut_fetch(
                        _hip.ptr(s_), _hip.ptr(probe), px, _hip.ptr(t), lut33.size, m, _hip.current_stream()), "probe"))))
                # channel split: one (33^3, 32^3) / two (25^3) channels of the node table in LDS, the rest gathered
                for nsz in (25, 28, 30, 32):      # cell-major 128-byte records (mode 4) against the 48-byte record runs (mode 0), by cube size
                    for mode, tab in ((0, rec_table), (4, cellmajor)):
                        cases.append((f"probe lut cellmajor-vs-records n={nsz} mode {mode} {src_name}", 16, (lambda m=mode, n_=nsz, t=tab, s_=src: _hip.check(
                            _hip.lib().vrg_debug_lut_fetch(_hip.ptr(s_), _hip.ptr(probe), px, _hip.ptr(t), n_, m, _hip.current_stream()), "probe"))))
                for nsz, mode in ((33, 5), (32, 5), (25, 6), (25, 5), (25, 0), (17, 6), (33, 7), (33, 8), (25, 7), (25, 8), (17, 7)):
                    cases.append((f"probe lut split n={nsz} mode {mode} {src_name}", 16, (lambda m=mode, n_=nsz, s_=src: _hip.check(_hip.lib().vrg_debug_lut_fetch(
                        _hip.ptr(s_), _hip.ptr(probe), px, _hip.ptr(rec_table), n_, m, _hip.current_stream()), "probe"))))
        for name, bpp, fn in cases:
            if match and match not in name:
                continue
            try:
                med, best = time_it(fn, iters, ops)
                row = {"size": label, "frames": F, "kernel": name, "ms": round(med, 4), "best_ms": round(best, 4),
                       "mpix_s": round(px / med / 1e3, 1), "algo_gbs": round(px * bpp / med / 1e6, 1),
                       "frac_8tbs": round(px * bpp / med / 1e6 / 8000.0, 4)}
            except Exception as exc:
                row = {"size": label, "kernel": name, "error": f"{type(exc).__name__}: {exc}"}
            rows.append(row)
            print("[diag]", row, flush=True)
        del x, smooth, out
        torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16, help="4K frames per timing batch (1080p uses 4x)")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--host", action="store_true", help="also measure host-fed (PCIe inclusive) node calls")
    ap.add_argument("--valu", action="store_true", help="also run the VALU issue-rate probe")
    ap.add_argument("--match", default="", help="only time kernels whose label contains this")
    ap.add_argument("--out", default=os.path.join(OUT_DIR, "diag.json"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    load_package()
    from comfyui_vrgamedevgirl_amd import ops
    device_info()
    noise_check(ops)
    if args.valu:
        valu_rate(ops)
    if args.host:
        host_fed(ops)
    kernel_bench(ops, args.frames, args.iters, args.match)
    with open(args.out, "w") as fh:
        json.dump(RESULT, fh, indent=1)
    print("[diag] written", args.out)


if __name__ == "__main__":
    main()
