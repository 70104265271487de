"""Host-fed ("ComfyUI-realistic") rates of the four nodes: CPU tensors in, CPU tensors out, PCIe inclusive -- what a graph sees, never
bench.py's `value`.  16 x 4K fp32 frames (1.6 GB each way per node call); pageable input (fresh torch.rand) and page-locked input (the
result of a previous node of this pack); second call onwards (the first call page-locks the result buffer).
    python tools/host_fed.py [--frames 16] [--out gpurun_out/host_fed_nodes.json]
bench.py imports `measure` for its `host_fed` key (8 frames)."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def measure(frames=16, H=2160, W=3840, reps=3, warmup=2):
    """Wall-clock seconds per call: two warm-up calls per case, then `reps` (>= 3) ROUNDS in which every case runs once -- the cases are
    interleaved so that a slow phase of the box hits all of them --, and the MEDIAN over the rounds with the spread (max - min) beside
    it.  (Round 3 reported the minimum of ONE repetition per case: page-locked input once came out slower than pageable.)"""
    import statistics
    from __graft_entry__ import load_package
    load_package()
    from comfyui_vrgamedevgirl_amd import nodes, _devices, VRGDG_IV_Adjustments as iv
    reps = max(3, int(reps))
    g = torch.Generator().manual_seed(3)
    x = torch.rand((frames, H, W, 3), generator=g)
    ref = x[:1].clone()
    px = frames * H * W
    nbytes = x.numel() * 4

    def once(fn):
        # the result is READ on the host inside the timed region (what the node after it would do if it were not one of this pack):
        # a node of this pack hands out its result before the download (_devices.LazyFrames), materialise() is that first host access
        t0 = time.perf_counter(); r = fn(); _devices.materialise(r); torch.cuda.synchronize(); dt = time.perf_counter() - t0; del r
        return dt

    cases = [("FastFilmGrain (bs 4)", lambda t: nodes.FastFilmGrain().apply_grain(t, 0.04, 0.5, 4)[0]),
             ("VRGDG_LUTS (AMD_TealOrange_33)", lambda t: iv.VRGDG_LUTS().apply_lut(t, "AMD_TealOrange_33.cube", "auto", 10.0)[0]),
             ("ColorMatchToReference (bs 1)", lambda t: nodes.ColorMatchToReference().match_color(t, ref, 1.0, 1)[0]),
             ("FastUnsharpSharpen", lambda t: nodes.FastUnsharpSharpen().apply_unsharp(t, 0.5, False)[0])]

    def chain(t):
        for _, fn in cases:
            t = fn(t)
        return t
    xp = x.pin_memory()
    runs = []
    for name, fn in cases:
        for label, src in (("pageable input", x), ("page-locked input", xp)):
            runs.append((name, label, (lambda f=fn, s_=src: f(s_))))
    def chain_with(t, **flags):
        old = {k: getattr(_devices, k) for k in flags}
        for k, v in flags.items():
            setattr(_devices, k, v)
        try:
            r = chain(t)
            _devices.materialise(r)
            return r
        finally:
            for k, v in old.items():
                setattr(_devices, k, v)
    runs.append(("grain -> LUT -> colour match -> unsharp, four node calls in a graph, deferred and fused (round 6: one upload, ONE fused chain per piece, "
                 "one download, all three in duplex)", "pageable input", lambda: chain(x)))
    runs.append(("grain -> LUT -> colour match -> unsharp, four node calls, each runs when called, intermediates stay in HBM (VRGDG_DEFER_GRAPH=0: round 5's "
                 "1 upload + 4 kernels + 1 download in sequence)", "pageable input", lambda: chain_with(x, DEFER_GRAPH=False)))
    runs.append(("grain -> LUT -> colour match -> unsharp, four node calls, every result downloaded at once (VRGDG_LAZY_DOWNLOAD=0: 1 upload + 4 downloads)",
                 "pageable input", lambda: chain_with(x, LAZY_DOWNLOAD=False)))
    for _ in range(max(1, int(warmup))):             # warm-up, at least twice: the first call page-locks the result buffers, the second still grows torch's device pool
        for _, _, fn in runs:      # (tools/diag_lazy_graph.py: calls 1 and 2 of the graph take 880 / 210 ms, every later one 76.5-77.8)
            once(fn)
    times = [[] for _ in runs]
    for _ in range(reps):
        for i, (_, _, fn) in enumerate(runs):
            times[i].append(once(fn))
    rows = []
    for (name, label, _), ts in zip(runs, times):
        t = statistics.median(ts)
        row = {"node": name, "input": label, "frames": frames, "seconds": round(t, 4), "Mpix_s": round(px / t / 1e6, 1),
               "seconds_min_max": [round(min(ts), 4), round(max(ts), 4)], "spread_pct": round(100.0 * (max(ts) - min(ts)) / t, 1), "reps": reps}
        if not name.startswith("grain ->"):
            row["GB_s_each_way"] = round(nbytes / t / 1e9, 2)
        rows.append(row)
    return {"frames": frames, "height": H, "width": W, "devices": os.environ.get("VRGDG_DEVICES", "") or "one", "timing": f"median of {reps} interleaved rounds after {max(1, int(warmup))} warm-up rounds",
            "rows": rows}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--out", default="gpurun_out/host_fed_nodes.json")
    a = ap.parse_args()
    res = measure(a.frames)
    for r in res["rows"]:
        print("[host]", r, flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1)
