#!/usr/bin/env python
"""Headline benchmark: Mpixels/s of the fused grain -> 33^3 LUT -> colour match -> unsharp chain at 4K.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N=1: plain python; N>1: launched by torchrun

Workload (BASELINE.json metric / configs[4] per-GPU shard): every rank holds `--frames` (default 256) synthetic
4K fp32 RGB frames resident in HBM (generated on device, never touched by the host), chain = Fast Film Grain
(I=0.04, s=0.5, one torch.randn draw per 4 frames) -> 3D LUT 33^3 (strength 10) -> Color Match to a 4K
reference frame (k=1) -> Fast Unsharp (0.5, edge-replicate).  A step = one pass of that chain over the rank's
batch: reference-frame statistics (rows split across ranks + all-reduce when N>1), the statistics pass over the
batch, and the fused apply pass.  Weak scaling: per-GPU work is fixed, value = all ranks' pixels / max time.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel of the step, timed with HIP events on the stream
it is launched on; `cpu_baseline` times the reference's own node classes on the host cores where the reference checkout
exists (the build container: kind "reference"), otherwise the oracle port of them (kind "port"), on a bounded sample.

`--gpus N` with N > 1 and no torchrun environment re-executes itself under `python -m torch.distributed.run` (one rank
per GPU, RCCL); it fails loudly when fewer than N GPUs are visible.  Colour-match arithmetic: the default "device" policy
(bit-equal to torch-ROCm's element-wise ops); the "fast" policy is timed next to it and reported
under `fast_variant`, never as `value`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CM_BATCH = 1                    # ColorMatchToReference's batch_size widget default: frames per statistics call of the reference
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable by a copy kernel
WORKLOADS = {
    # name: (H, W, stages)
    "chain4_4k": (2160, 3840, ("grain", "lut", "colormatch", "sharpen")),      # headline (configs[4] per-GPU shard)
    "chain3_4k": (2160, 3840, ("grain", "lut", "sharpen")),                    # configs[2]
    "grain_lut_1080p": (1080, 1920, ("grain", "lut")),                         # configs[1]
    "colormatch_4k": (2160, 3840, ("colormatch",)),                            # configs[3]
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU (default: BASELINE's 256; 128 for the 1080p workload, 512 for colour match alone)")
    ap.add_argument("--workload", default="chain4_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--dist", default="uniform", choices=["uniform", "video"], help="synthetic pixel distribution")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2, help="4K frames of the bounded CPU-baseline sample (x4 at 1080p)")
    ap.add_argument("--no-fast-variant", action="store_true", help="skip the extra timing of the fast colour-match policy")
    ap.add_argument("--sync-ref", action="store_true", help="reference-frame statistics on the main stream in front of pass 1 (A/B of the side-stream form)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes for roofline.traffic (use profiles/)")
    ap.add_argument("--no-verify", action="store_true", help="skip the output check after the timed region (first and last RNG chunk of rank 0 against "
                                                             "the stand-alone operators and the device oracle)")
    ap.add_argument("--digest", action="store_true", help="add the SHA-256 of every rank's output (per RNG chunk) to the line: equal frame ranges of "
                                                          "runs with different GPU counts must give equal digests")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed (CPU tensors in and out, PCIe inclusive) node rates of the line's `host_fed` key")
    ap.add_argument("--same-data", action="store_true", help="frames are a function of their ABSOLUTE index in the job (rank r holds frames "
                                                             "[r*frames, (r+1)*frames) of one job-wide batch), so that runs with different GPU counts process "
                                                             "the same data; default: an independent batch per rank")
    ap.add_argument("--cm-stats", default=None, choices=["device", "fp64"], help="colour statistics of the headline leg (default: device)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: become `torch.distributed.run --nproc-per-node N bench.py ...`."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    n_vis = torch.cuda.device_count()
    if n_vis < args.gpus and os.environ.get("VRGDG_DIST_BACKEND", "nccl") == "nccl":
        sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but only {n_vis} GPU(s) are visible to PyTorch-ROCm; "
                         "refusing to print a line for fewer GPUs than asked for\n")
        raise SystemExit(2)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def make_frames(n, H, W, dev, seed, dist, first_frame=None):
    """Synthetic frames generated on the device.  uniform: iid U[0,1) (worst case for LUT gathers);
    video: smooth low-frequency field + N(0, 0.02) texture, clamped (LUT-coherent like real footage).
    `first_frame` (--same-data): frame i is drawn from a generator seeded by its absolute index first_frame + i."""
    g = torch.Generator(device=dev).manual_seed(seed)
    if first_frame is not None:
        x = torch.empty((n, H, W, 3), dtype=torch.float32, device=dev)
        for i in range(n):
            x[i] = make_frames(1, H, W, dev, seed * 1000003 + first_frame + i, dist)[0]
        return x
    if dist == "uniform":
        x = torch.empty((n, H, W, 3), dtype=torch.float32, device=dev)
        for i in range(0, n, 16):
            x[i:i + 16].copy_(torch.rand((min(16, n - i), H, W, 3), generator=g, device=dev))
        return x
    yy = torch.linspace(0, 1, H, device=dev).view(1, H, 1, 1)
    xx = torch.linspace(0, 1, W, device=dev).view(1, 1, W, 1)
    x = torch.empty((n, H, W, 3), dtype=torch.float32, device=dev)
    for i in range(n):
        ph = torch.rand((1, 1, 1, 3), generator=g, device=dev) * 6.28
        fr = 0.5 + 0.25 * torch.sin(6.0 * xx + ph) * torch.cos(4.0 * yy + 0.5 * ph) + 0.15 * torch.sin(9.0 * yy - ph)
        fr = fr + 0.02 * torch.randn((1, H, W, 3), generator=g, device=dev)
        x[i] = fr.clamp_(0, 1)[0]
    return x


def _profile_json(*names):
    """first of the committed PMC summaries that exists (newest round first)"""
    for n in names:
        path = os.path.join(ROOT, "profiles", n)
        if os.path.exists(path):
            with open(path) as fh:
                return json.load(fh), n
    raise FileNotFoundError(names[0])


def live_traffic(timeout_s=150):
    """HBM bytes and VALU lane-instructions per pixel of the headline kernels from the PMC counters ON THIS BOX: three extra
    `rocprofv3 --kernel-trace --pmc` runs (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU: separate passes, as MI355X_MICROARCH.md
    prescribes) of tools/prof_driver.py, which executes
    the same kernels on 16 x 4K frames in its own process (rocprofv3 wraps a process, so it cannot observe this one).
    FETCH_SIZE is doubled (the gfx950 under-count for wide coalesced reads; calibrated on k_lut3d's known 12 B/px in the same
    run), both counters are in KB.  Returns {"stats": {...}, "apply": {...}, "chain3_apply": {...}} or raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        raise RuntimeError("rocprofv3 not on PATH")
    tmp = tempfile.mkdtemp(prefix="vrg_traffic_")
    per_px = {}
    px = 16 * 2160 * 3840
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT64")):
            out = os.path.join(tmp, counters[0])
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            subprocess.run([exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out, "-o", "p", "--",
                            sys.executable, os.path.join(ROOT, "tools", "prof_driver.py"), "traffic"],
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if "vrg" in r["Kernel_Name"] and r["Counter_Name"] in counters:
                            # FETCH_SIZE / WRITE_SIZE count KB; the SQ_INSTS_* count wave64 instructions (x64 lanes)
                            scale = 64.0 if r["Counter_Name"].startswith("SQ_") else 1024.0
                            per_px.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]) * scale / px)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def bpp(match):
        rd = sum(sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 2.0 for k, v in per_px.items() if match(k) and "FETCH_SIZE" in v)
        wr = sum(sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) for k, v in per_px.items() if match(k) and "WRITE_SIZE" in v)
        def avg(c):
            return sum(sum(v[c]) / len(v[c]) for k, v in per_px.items() if match(k) and c in v)
        vi, tr, i64 = avg("SQ_INSTS_VALU"), avg("SQ_INSTS_VALU_TRANS_F32"), avg("SQ_INSTS_VALU_INT64")
        return {"read": round(rd, 2), "written": round(wr, 2), "total": round(rd + wr, 2), "valu_lane_instr": round(vi, 1),
                "valu_trans": round(tr, 1), "valu_int64": round(i64, 1)}
    res = {"stats": bpp(lambda k: "k_produce_lab<3" in k), "apply": bpp(lambda k: "k_apply_march<20" in k or "k_chain_tile<20" in k),
           "tstats": bpp(lambda k: "k_tstats_frame" in k or "k_tstats_rows<" in k),
           "chain3_apply": bpp(lambda k: "k_chain_march<3" in k), "calibration_k_lut3d": bpp(lambda k: "k_lut3d" in k)}
    if not res["calibration_k_lut3d"]["total"]:
        raise RuntimeError("no counters collected")
    return res


def _median_time(fn, warmup=1, reps=3):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def cpu_baseline(stages, cpu_frames, H, W, lut_cpu, per_node=False):
    """The chain on the host cores over a bounded sample: warm-up 1, median of 3.  kind "reference": the reference's own
    node classes (FastFilmGrain / VRGDG_LUTS / ColorMatchToReference / FastUnsharpSharpen, stub-loaded by
    oracle/reference_loader.py; kornia's Lab transforms are the restated ones) -- only where the reference checkout exists,
    i.e. not on the GPU box; kind "port": oracle/restated.py, the op-for-op port of them."""
    from oracle import reference_loader as RL
    g = torch.Generator().manual_seed(7)
    x = torch.rand((cpu_frames, H, W, 3), generator=g)
    ref = torch.rand((1, H, W, 3), generator=g)
    mpix = cpu_frames * H * W / 1e6
    if RL.reference_available():
        nodes, iv = RL.load_nodes(), RL.load_iv_adjustments()
        lut_name = "Vintage Color.cube"                      # the reference's own 33^3 asset
        fns = {"grain": lambda y: nodes.FastFilmGrain().apply_grain(y, 0.04, 0.5, 4)[0],
               "lut": lambda y: iv.VRGDG_LUTS().apply_lut(y, lut_name, "cpu", 10.0)[0],
               "colormatch": lambda y: nodes.ColorMatchToReference().match_color(y, ref, 1.0, 1)[0],
               "sharpen": lambda y: nodes.FastUnsharpSharpen().apply_unsharp(y, 0.5, False)[0]}
        kind, what = "reference", "the reference's node classes (oracle/reference_loader.py; kornia Lab transforms restated)"
    else:
        from oracle import restated as R
        fns = {"grain": lambda y: R.fast_film_grain(y, 0.04, 0.5, 4),
               "lut": lambda y: R.apply_lut_with_strength(y, lut_cpu, 10.0),
               "colormatch": lambda y: R.color_match(y, ref, 1.0, 1),
               "sharpen": lambda y: R.unsharp(y, 0.5, False)}
        kind, what = "port", "oracle/restated.py (op-for-op port of the reference's eager torch / numpy ops)"

    def chain():
        y = x
        for st in ("grain", "lut", "colormatch", "sharpen"):
            if st in stages:
                y = fns[st](y)
        return y

    # torch's default on a 256-thread host is 128 intra-op threads: on 2 frames that mostly measures its threading overhead, so a
    # one-run probe picks the better of the default and 32 threads before the timed runs (`cores` = the count actually used)
    default_threads = torch.get_num_threads()
    probe = {}
    for nt in sorted({default_threads, min(32, default_threads)}):
        torch.set_num_threads(nt)
        probe[nt] = _median_time(chain, warmup=1 if not probe else 0, reps=1)
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    dt = _median_time(chain, warmup=0)
    out = {"value": round(mpix / dt, 2), "unit": "Mpixels/s", "cores": torch.get_num_threads(), "kind": kind,
           "thread_probe_s": {str(k): round(v, 2) for k, v in probe.items()},
           "sample": f"{cpu_frames} frames {W}x{H}, chain {'+'.join(stages)} via {what}; warm-up 1, median of 3 = {dt:.2f} s; "
                     f"os.cpu_count()={os.cpu_count()}, torch.get_num_threads()={torch.get_num_threads()} (numpy unsharp is single-threaded)"}
    if per_node:
        out["per_node_mpix_s"] = {st: round(mpix / _median_time(lambda st=st: fns[st](x)), 2) for st in stages}
    return out


def verify_output(ops, x, out, ref, lut, lut_cpu, stages, geom_stream, rank, frames, chunk, dev, cm_stats):
    """Outside the timed region, on rank 0's own output: the first and the last RNG chunk of the rank's batch (at 4K with chunk 4 the
    Philox quarters of a chunk span 11.4 rows and their sibling runs cross frame boundaries inside every chunk) are re-derived
      (a) with the stand-alone operators of this library, one after the other (film_grain -> lut3d -> color_match -> stencil3x3), and
      (b) with the oracle composition (oracle/chain_oracle.py: torch.randn on the device, the CPU restatements pinned to the reference's
          fixtures for grain / LUT / unsharp, the restated colour match evaluated by torch on the device) -- the checker, never timed,
    and compared bit for bit with what the timed steps left in `out`.  (b) holds with the device statistics; with fp64 statistics the
    oracle differs by the statistics band and only (a) is asserted."""
    from oracle import chain_oracle as CO          # checker only (like the cpu_baseline leg)
    res = {"frames_checked": [], "vs_standalone_operators": True, "vs_device_oracle": None, "max_abs_vs_oracle": 0.0}
    chunks = sorted({0, frames // chunk - 1}) if "grain" in stages else [0, max(frames // chunk - 1, 0)]
    ref_ms = ops.reference_stats(ref, cm_stats=cm_stats) if "colormatch" in stages else None
    for c in sorted(set(chunks)):
        f0, f1 = c * chunk, min(frames, (c + 1) * chunk)
        xs = x[f0:f1]
        y = xs
        if "grain" in stages:
            plan = ops.NoisePlan(chunk, geom_stream, chunk0=rank * (frames // chunk) + c)
            y = ops.film_grain(y, 0.04, 0.5, chunk_frames=chunk, plans=(plan, None, 1))
        if "lut" in stages:
            y = ops.lut3d(y, lut, 10.0)
        if "colormatch" in stages:
            y = ops.color_match(y, None, 1.0, ref_ms=ref_ms, cm_chunk=CM_BATCH, cm_stats=cm_stats)
        if "sharpen" in stages:
            y = ops.stencil3x3(y, "unsharp", 0.5, False)
        got = out[f0:f1]
        res["vs_standalone_operators"] = bool(res["vs_standalone_operators"] and torch.equal(got, y))
        want = CO.headline_chain(xs.cpu(), dev, stages=stages, stream=geom_stream, chunk0=rank * (frames // chunk) + c, chunk_frames=chunk,
                                 lut_cpu=lut_cpu, reference_dev=ref, cm_batch=CM_BATCH)
        same = torch.equal(got.cpu(), want)
        res["max_abs_vs_oracle"] = max(res["max_abs_vs_oracle"], float((got.cpu() - want).abs().max()))
        res["vs_device_oracle"] = same if res["vs_device_oracle"] is None else (res["vs_device_oracle"] and same)
        res["frames_checked"].append([f0, f1])
        del y, want
    oracle_must_hold = "colormatch" not in stages or (cm_stats or "device") == "device"
    res["verified"] = bool(res["vs_standalone_operators"] and (res["vs_device_oracle"] or not oracle_must_hold))
    return res


def output_digests(out, chunk):
    """SHA-256 per RNG chunk of a rank's output (bytes of the fp32 frames): independent of how many ranks share the job."""
    import hashlib
    return [hashlib.sha256(out[f:f + chunk].cpu().numpy().tobytes()).hexdigest() for f in range(0, out.shape[0], chunk)]


def main():
    args = parse_args()
    self_launch(args)
    # stdout carries exactly one line, the JSON: everything else that writes to fd 1 while we run (RCCL prints a
    # version banner to stdout when a communicator is created) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from __graft_entry__ import load_package
    load_package()
    from comfyui_vrgamedevgirl_amd import ops, sharding
    from comfyui_vrgamedevgirl_amd import VRGDG_IV_Adjustments as iv
    from comfyui_vrgamedevgirl_amd import cube
    import torch.distributed as dist

    rank, local, world = sharding.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with torchrun --nproc-per-node {args.gpus}, or let "
                         "bench.py launch itself)")
    if world > 1:
        want_backend = os.environ.get("VRGDG_DIST_BACKEND", "nccl")      # gloo only for functional tests on a 1-GPU box
        if not dist.is_initialized() or dist.get_world_size() != world or dist.get_backend() != want_backend:
            raise SystemExit("bench.py: RCCL process group did not come up with the requested world size")
    dev = torch.device("cuda", torch.cuda.current_device())
    H, W, stages = WORKLOADS[args.workload]
    frames = args.frames or {"grain_lut_1080p": 128, "colormatch_4k": 512}.get(args.workload, 256)      # BASELINE.json configs[1..4]
    chunk = 4

    lut_cpu = cube.parse_cube_file(os.path.join(iv.LUTS_DIR, "AMD_TealOrange_33.cube"))
    lut = ops.upload_lut(lut_cpu, dev)
    x = make_frames(frames, H, W, dev, 1234 + (0 if args.same_data else rank), args.dist, first_frame=rank * frames if args.same_data else None)
    out = torch.empty_like(x)
    lab_ws = torch.empty_like(x) if "colormatch" in stages else None      # Lab image between the two colour-match passes
    ref = make_frames(1, H, W, dev, 4321, args.dist)          # same reference frame on every rank
    fe = H * W * 3

    # one job-wide generator state: rank r takes chunks [r*frames/chunk, ...) of the same stream, so the result
    # does not depend on the number of GPUs
    geom_stream = None
    if "grain" in stages:
        gen = torch.Generator(device=dev).manual_seed(42)
        geom_stream = ops.rng.reserve(chunk * fe, world * frames // chunk, dev, gen)

    ref_events = []

    def step(kernel_events=None, cm_math=None, cm_stats=args.cm_stats):
        ref_ms = ref_ev = None
        if "colormatch" in stages:
            if ops._cm_stats(cm_stats, cm_math, dev) == "device":
                # device statistics = torch's own reduction over the WHOLE reference frame: every rank evaluates it (no exchange), on
                # the side stream -- only the apply pass needs it, pass 1 of the batch runs meanwhile
                if args.sync_ref:
                    ref_ms = ops.reference_stats(ref, cm_math, step_frames=frames)
                else:
                    ref_ms, ref_ev = ops.reference_stats_async(ref, cm_math, step_frames=frames)
            else:
                # fp64 (n, mean, M2): rows split across the ranks + all-reduce over RCCL (BASELINE configs[4])
                if kernel_events is not None:
                    r0, r1 = ops.HipEvent(), ops.HipEvent()
                    r0.record()
                ref_ms = sharding.reference_stats_sharded(ref, rank, world, cm_math=cm_math)      # fp64 triples, RCCL all-reduce
                if kernel_events is not None:
                    r1.record()
                    ref_events.append((r0, r1))
        plans = None
        if geom_stream is not None:
            plans = (ops.NoisePlan(chunk, geom_stream, chunk0=rank * (frames // chunk)), None, frames // chunk)
        spec = ops.ChainSpec(grain=(0.04, 0.5, chunk) if "grain" in stages else None,
                             lut=(lut, 10.0) if "lut" in stages else None,
                             colormatch=(ref_ms, 1.0) if "colormatch" in stages else None,
                             sharpen=("unsharp", 0.5, False) if "sharpen" in stages else None, cm_math=cm_math, cm_chunk=CM_BATCH,
                             cm_ref_event=ref_ev, cm_stats=(cm_stats if cm_math is None else None))
        ops.fused_chain(x, spec, plans=plans, out=out, kernel_events=kernel_events, lab_workspace=lab_ws)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [round(float(v.item()) / args.steps * 1e3, 3) for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the timed steps' own output, checked and fingerprinted BEFORE anything else overwrites it (never inside the timed region)
    verify = digests = None
    if rank == 0 and not args.no_verify:
        try:
            verify = verify_output(ops, x, out, ref, lut, lut_cpu, stages, geom_stream, rank, frames, chunk, dev, args.cm_stats)
        except Exception as exc:              # a checker problem must not lose the measurement; it is reported as unverified
            verify = {"verified": False, "error": f"{type(exc).__name__}: {exc}"}
    if args.digest:
        mine = output_digests(out, chunk)
        digests = [mine]
        if dist.is_initialized():
            digests = [None] * world
            dist.all_gather_object(digests, mine)
    # the fast colour-match policy, same data, timed the same way (reported beside the headline, never as `value`)
    fast_variant = None
    if "colormatch" in stages and not args.no_fast_variant:
        step(cm_math="fast", cm_stats=None)
        barrier()
        fast_events = []
        f0 = time.perf_counter()
        for _ in range(args.steps):
            step(fast_events, cm_math="fast", cm_stats=None)
        barrier()
        fel = time.perf_counter() - f0
        if dist.is_initialized():
            t = torch.tensor([fel], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fel = float(t.item())
        fast_variant = {"cm_math": "fast", "value": round(world * frames * H * W * args.steps / fel / 1e6, 1), "unit": "Mpixels/s",
                        "ms_per_step": round(fel / args.steps * 1e3, 3),
                        "cm_stats": "fp64 (reference-frame rows split across the ranks, merged by the RCCL all-reduce when N > 1)",
                        "note": "table-driven powers (<= 0.534 ulp) instead of ocml powf and fp64-accumulated statistics instead of torch's "
                                "fp32 reductions: a few ulp from the reference, not bit-equal to it (DESIGN.md section 4)"}

    px_rank = frames * H * W
    value = world * px_rank * args.steps / elapsed / 1e6
    # per-pass device time from HIP events on the launch stream; the dominant pass carries the roofline object
    passes = {}
    for name, a, b, nf in events:
        passes.setdefault(name, []).append(a.elapsed_ms(b))
    pass_ms = {k: sum(v) / args.steps for k, v in passes.items()}          # per step (pieces of a step added up)
    launches_per_step = {k: len(v) // max(args.steps, 1) for k, v in passes.items()}
    algo_bpp = {"stats": 12, "apply": 24, "tstats": 12}         # SURVEY.md section 8d; tstats re-reads the Lab image (12 B/px)
    kern_names = {"stats": ("k_produce_lab, Lab-only form (grain->LUT->Lab pass 1: shared Philox, stores the Lab image; the statistics are reduced from it "
                            "by k_tstats_frame)" if "grain" in stages else
                            "k_lab_partials, Lab-only form (rgb->Lab pass 1, stores the Lab image; the statistics are reduced from it by k_tstats_frame)"),
                  "tstats": "k_tstats_frame (torch's mean / Welford reductions replayed over the stored Lab image)",
                  "apply": "k_apply_march<COLORMATCH|FROM_LAB> (match -> Lab->RGB -> 3x3 sharpen, register-resident wave march)" if "colormatch" in stages
                  else ("k_chain_march (fused grain -> LUT -> sharpen, register-resident wave march)" if "sharpen" in stages and "grain" in stages
                        else "k_chain_tile / k_chain_pointwise (fused apply pass)")}
    dom = max(pass_ms, key=pass_ms.get)
    kern_avg_ms = pass_ms[dom]
    algo_bytes = algo_bpp[dom] * px_rank
    achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
    bytes_per_px_chain = 36 if "colormatch" in stages else 24
    # HBM traffic of the dominant pass from the PMC run committed under profiles/ (rocprofv3 cannot run inside this
    # process): bytes per pixel measured there x the pixels of one launch here
    traffic, traffic_note, live_ipp, live_classes = None, None, None, (None, None)
    key = dom if "colormatch" in stages else "chain3_apply"
    if rank == 0 and world == 1 and not args.no_live_traffic and args.workload in ("chain4_4k", "chain3_4k"):
        try:
            del out, lab_ws                                   # the profiled side process needs ~5 GB of the HBM
            torch.cuda.empty_cache()
            summ = live_traffic()
            live_ipp = summ.get(key, {}).get("valu_lane_instr") or None
            live_classes = (summ.get(key, {}).get("valu_trans"), summ.get(key, {}).get("valu_int64"))
            if summ.get(key, {}).get("total"):
                traffic = round(summ[key]["total"] * px_rank / 1e9, 2)
                traffic_note = (f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE on this box right after the timed region (separate "
                                f"passes, own process, the same kernels on 16x4K frames): {summ[key]['read']} B/px read + "
                                f"{summ[key]['written']} B/px written (FETCH_SIZE x2 per the gfx950 calibration; k_lut3d in the same run: "
                                f"{summ['calibration_k_lut3d']['read']} + {summ['calibration_k_lut3d']['written']} for its known 12 + 12) x this "
                                "launch's pixels" + ("; the written bytes are the Lab image kept for pass 2, not re-reads" if key == "stats" else ""))
        except Exception as exc:
            traffic_note = f"live PMC passes failed ({type(exc).__name__}: {exc}); "
    try:
        if traffic is not None or args.workload not in ("chain4_4k", "chain3_4k"):      # the PMC passes cover these two workloads' kernels
            raise StopIteration
        tj, tname = _profile_json("r04_pmc_traffic_fetch_write.json", "r03_pmc_traffic_fetch_write.json", "r02_pmc_traffic_fetch_write.json")
        summ = tj.get("summary", {})
        if summ.get(key, {}).get("total"):
            traffic = round(summ[key]["total"] * px_rank / 1e9, 2)
            traffic_note = (traffic_note or "") + (f"NOT collected in this process (rocprofv3 wraps a process): profiles/{tname}, the same kernels on 16x4K frames: "
                            f"{summ[key]['read']} B/px read + {summ[key]['written']} B/px written (rocprofv3 --pmc "
                            "FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 per the gfx950 calibration) x this launch's pixels"
                            + ("; the written bytes are the Lab image kept for pass 2 (design choice measured in DESIGN.md section 3: "
                               "40.3 vs 25.4 Gpix/s against the recomputing form), not re-reads" if key == "stats" else ""))
    except Exception:
        pass

    # VALU issue roofline of the same pass (the one that actually binds it): lane-instructions per
    # pixel from the committed PMC pass (SQ_INSTS_VALU, calibrated on the issue-rate probe) x this run's pixel rate,
    # against the probe's measured peak for plain fp32 / integer ops
    issue = None
    try:
        recs, iname = _profile_json("r04_pmc_valu_instr_per_px.json", "r03_pmc_valu_instr_per_px.json", "r02_pmc_valu_instr_per_px.json")
        rates, rname = _profile_json("r02_valu_issue_rate_long.json", "r01_valu_issue_rate.json")
        peak_t = max(r["tera_lane_instr_s"] for r in rates["rows"] if r["instr"] == "v_fma_f32")
        # only the kernels the committed PMC pass covers: the two passes of the headline chain and the chain-3 march
        want = {("chain4_4k", "stats"): "k_produce_lab<3, false>", ("chain4_4k", "apply"): "k_chain_tile<20",
                ("chain3_4k", "apply"): "k_chain_march<3"}[(args.workload, dom)]
        ipp = live_ipp if live_ipp else next(r["valu_lane_instr_per_px"] for r in recs if want in r["kernel"])
        src = ("a rocprofv3 --pmc SQ_INSTS_VALU pass on this box right after the timed region (own process, 16x4K frames)" if live_ipp
               else f"profiles/{iname} (SQ_INSTS_VALU, not collected in this process)")
        rate_t = ipp * px_rank / (kern_avg_ms * 1e-3) / 1e12
        weighted = None
        if live_ipp and live_classes[0] is not None:
            # issue cost of the measured classes relative to a plain fp32 / integer op (profiles/r02_valu_issue_rate_long.json):
            # transcendental 3.45x, 64-bit integer multiply-add 1.8x (compare + select pairs, 1.65x, have no counter: not included)
            w_ipp = ipp + 2.45 * live_classes[0] + 0.8 * live_classes[1]
            w_rate = w_ipp * px_rank / (kern_avg_ms * 1e-3) / 1e12
            weighted = {"lane_instr_per_px": round(w_ipp, 1), "of_which_transcendental": live_classes[0], "of_which_int64": live_classes[1],
                        "achieved": round(w_rate, 2), "frac_of_measured_peak_66p5": round(w_rate / peak_t, 4),
                        "frac_of_guide_peak_78p6": round(w_rate / 78.6, 4), "unweighted_frac_of_guide_peak_78p6": round(rate_t / 78.6, 4)}
        issue = {"bound": "valu-issue", "lane_instr_per_px": round(ipp, 1), "achieved": round(rate_t, 2), "peak": peak_t,
                 "unit": "T lane-instr/s", "frac": round(rate_t / peak_t, 4), "weighted": weighted,
                 "note": f"lane-instructions per pixel from {src} x this run's pixel rate; "
                         f"peak = v_fma_f32 at 8 waves/SIMD over >= 17 ms launches (profiles/{rname}); unweighted: v_pk_* / fp64 / "
                         "v_mad_u64_u32 issue at 1.8x, compare+select pairs 1.65x, transcendentals 3.45x a plain op"}
    except Exception:
        pass

    # reference-frame statistics: the device form is overlapped with pass 1 in the step, so it is timed on its own here (3 runs, median);
    # the fp64 / all-reduce form of the fast variant was bracketed by events inside its steps
    ref_ms_per_step = ref_allreduce_ms = None
    if "colormatch" in stages:
        ts = []
        for _ in range(4):
            a, b = ops.HipEvent(), ops.HipEvent()
            a.record(); ops.reference_stats(ref, step_frames=frames); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_ms(b))
        ref_ms_per_step = round(sorted(ts[1:])[1], 4)
        if ref_events:
            ref_allreduce_ms = round(sum(a.elapsed_ms(b) for a, b in ref_events) / max(args.steps, 1), 4)
    if rank == 0:
        line = {
            "metric": "Mpixels/s (grain+LUT+colormatch+sharpen) at 4K" if args.workload == "chain4_4k" else f"Mpixels/s ({'+'.join(stages)})",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": f"synthetic-{args.dist} (generated on device, resident in HBM)",
            "config": {"workload": f"{W}x{H} x{frames} frames/GPU, {'+'.join(stages)} fused chain, LUT 33^3 (AMD_TealOrange_33.cube, this "
                                   f"pack's own cube: same size as the reference's Vintage Color.cube), grain chunk {chunk} "
                                   f"(BASELINE configs[4] per-GPU shard)" if args.workload == "chain4_4k"
                       else f"{W}x{H} x{frames} frames/GPU, {'+'.join(stages)}",
                       "frames_per_gpu": frames, "height": H, "width": W, "parallelism": f"frames sharded x{world}",
                       "algorithmic_bytes_per_pixel_chain": bytes_per_px_chain,
                       "cm_math": "device" if "colormatch" in stages else None,
                       "cm_stats": ((f"device (torch-ROCm's reductions bit for bit, batch_size {CM_BATCH})" if (args.cm_stats or "device") == "device"
                                     else "fp64 (reference-frame rows split across the ranks, RCCL all-reduce)") if "colormatch" in stages else None),
                       "lut_note": "synthetic-uniform pixels make the LUT gathers content-independent; with --dist video their cost depends on the "
                                   "cube's shape (this pack's AMD_TealOrange_33.cube, not the reference's Vintage Color.cube)"},
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            "dist_backend": (dist.get_backend() if dist.is_initialized() else None),
            "per_rank_ms_per_step": per_rank_ms,
            "reference_stats_ms_per_step": ref_ms_per_step,
            "reference_stats_note": ("device statistics of the whole reference frame on every rank, on a side stream next to pass 1 (timed alone here); "
                                     "the fast variant's fp64 form -- rows split across the ranks + RCCL all-reduce -- took "
                                     f"{ref_allreduce_ms} ms per step") if "colormatch" in stages else None,
            "chain_hbm_frac": round(value / world * bytes_per_px_chain * 1e6 / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": kern_names[dom],
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(kern_avg_ms, 4),
                         "passes_ms": {k: round(v, 4) for k, v in pass_ms.items()}, "launches_per_step": launches_per_step,
                         "issue": issue},
        }
        if world == 1 and not args.no_host_fed and args.workload == "chain4_4k":
            # what a ComfyUI graph sees: CPU tensors in, CPU tensors out, every node call crossing PCIe both ways -- never `value`
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import host_fed
                del x
                torch.cuda.empty_cache()
                line["host_fed"] = host_fed.measure(frames=8, reps=3)
            except Exception as exc:
                line["host_fed"] = {"error": f"{type(exc).__name__}: {exc}"}
        line["verified"] = None if verify is None else bool(verify.get("verified"))
        line["verify"] = verify
        if digests is not None:
            line["output_sha256_per_rank_per_chunk"] = digests
        if fast_variant is not None:
            line["fast_variant"] = fast_variant
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(stages, args.cpu_frames if H > 1080 else 4 * args.cpu_frames, H, W, lut_cpu)
            except Exception as exc:      # never lose the GPU line to a host-side problem
                line["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": torch.get_num_threads(), "kind": "port",
                                        "sample": f"failed: {type(exc).__name__}: {exc}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
