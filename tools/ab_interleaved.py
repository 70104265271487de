"""A/B (/C ...) of library builds with the variants INTERLEAVED in one process on one box: every round runs every case once with every
build, >= 5 rounds; per (case, build) the median and the spread (max - min) of the rounds; a winner is only named when the medians
differ by more than the larger of the two spreads.  Round 3 accepted 1-3 % single-run A/Bs of different processes; this tool replaces
that.

    python tools/ab_interleaved.py --libs base=comfyui-vrgamedevgirl_amd/libvrgdg_hip.so,new=tools/ab/lib_new.so \
           --cases chain3,chain3_video,chain4 --frames 64 --rounds 7 [--json gpurun_out/ab.json]

Cases: chain3 / chain3_video (grain -> LUT 33^3 -> unsharp on uniform / video-like 4K frames), grain_lut, chain4 (per-pass times of the
headline chain: pass1 / stats / pass2 / wall), kernels (grain, lut33, unsharp, sharpen>grain fused).
All builds must export the same ABI; outputs of the builds are compared bit for bit on the first round (a variant that changes bits
is reported, not timed as a candidate)."""
import argparse, ctypes as C, json, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_vrgamedevgirl_amd import ops, cube, _hip, VRGDG_IV_Adjustments as iv

ap = argparse.ArgumentParser()
ap.add_argument("--libs", required=True, help="name=path[,name=path...]; the first is the baseline")
ap.add_argument("--cases", default="chain3,chain3_video")
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--json", default="")
ap.add_argument("--lut", default="AMD_TealOrange_33.cube", help="cube under the package's LUTS/ for the chain cases")
a = ap.parse_args()
if a.rounds < 5:
    raise SystemExit("ab_interleaved: fewer than 5 rounds is not an A/B")
libs = []
for item in a.libs.split(","):
    name, path = item.split("=", 1)
    libs.append((name, _hip.load_library(os.path.abspath(path))))
_hip._lib = libs[0][1]
dev = torch.device("cuda", 0)
F, H, W = a.frames, 2160, 3840
px = F * H * W
g = torch.Generator(device=dev).manual_seed(3)
x = torch.rand((F, H, W, 3), generator=g, device=dev)
need_video = any(c.endswith("_video") for c in a.cases.split(","))
if need_video:
    sys.path.insert(0, ROOT)
    import bench
    xv = bench.make_frames(F, H, W, dev, 1234, "video")
out = torch.empty_like(x)
ws = torch.empty_like(x) if "chain4" in a.cases else None
lut = ops.upload_lut(cube.parse_cube_file(os.path.join(iv.LUTS_DIR, a.lut)), dev)
gen = torch.Generator(device=dev)


def timed(fn):
    e0, e1 = ops.HipEvent(), ops.HipEvent()
    e0.record(); fn(); e1.record()
    return e0.elapsed_ms(e1)


def chain3(src):
    gen.manual_seed(5)
    ops.fused_chain(src, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False)), generator=gen, out=out)


def grain_lut(src):
    gen.manual_seed(5)
    ops.fused_chain(src, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0)), generator=gen, out=out)


ref_ms = None


def chain4_passes(src=None):
    global ref_ms
    src = x if src is None else src
    if ref_ms is None:
        ref_ms = ops.reference_stats(x[:1])
    gen.manual_seed(5)
    ev = []
    w0, w1 = ops.HipEvent(), ops.HipEvent()
    w0.record()
    ops.fused_chain(src, ops.ChainSpec(grain=(0.04, 0.5, 4), lut=(lut, 10.0), colormatch=(ref_ms, 1.0), sharpen=("unsharp", 0.5, False), cm_chunk=1),
                    generator=gen, out=out, lab_workspace=ws, kernel_events=ev)
    w1.record()
    torch.cuda.synchronize()
    tot = {}
    for name, e0, e1, nf in ev:
        tot[name] = tot.get(name, 0.0) + e0.elapsed_ms(e1)
    tot["wall"] = w0.elapsed_ms(w1)
    return tot


def run_case(case):
    """-> {metric: ms} for the currently bound library; leaves the case's result in `out`"""
    if case == "chain3":
        return {"chain3": timed(lambda: chain3(x))}
    if case == "chain3_video":
        return {"chain3_video": timed(lambda: chain3(xv))}
    if case == "grain_lut":
        return {"grain_lut": timed(lambda: grain_lut(x))}
    if case == "grain_sharpen":
        def gs():
            gen.manual_seed(5)
            ops.fused_chain(x, ops.ChainSpec(grain=(0.04, 0.5, 4), sharpen=("unsharp", 0.5, False)), generator=gen, out=out)
        return {"grain_sharpen": timed(gs)}
    if case == "chain4":
        return {"chain4." + k: v for k, v in chain4_passes().items()}
    if case == "chain4_video":
        return {"chain4_video." + k: v for k, v in chain4_passes(xv).items()}
    if case == "colormatch":           # configs[3]: the colour transfer alone (statistics + Lab image, torch-order statistics, apply)
        global ref_ms
        if ref_ms is None:
            ref_ms = ops.reference_stats(x[:1])
        return {"colormatch": timed(lambda: ops.color_match(x, x[:1], 1.0, ref_ms=ref_ms, out=out))}
    if case == "kernels":
        r = {}
        gen.manual_seed(5)
        r["grain"] = timed(lambda: ops.film_grain(x, 0.04, 0.5, chunk_frames=4, generator=gen))
        r["lut33"] = timed(lambda: ops.lut3d(x, lut, 10.0))
        r["unsharp"] = timed(lambda: ops.stencil3x3(x, "unsharp", 0.5, False))
        r["sharpen>grain"] = timed(lambda: ops.sharpen_then_seeded_grain(x, 0.5, False, 0.04, 0.5, 42, 0))
        return r
    raise SystemExit(f"unknown case {case}")


cases = a.cases.split(",")
times = {}
bits = {}
for rnd in range(a.rounds + 1):          # round 0 = warm-up + the bit comparison
    for case in cases:
        # the first library of a round follows another case (other kernels, cold L2, clocks): measured 4-5 % slower whatever it is.
        # Rotate the order round by round, and run one untimed pass of the case first.
        order = libs[rnd % len(libs):] + libs[:rnd % len(libs)]
        _hip._lib = order[-1][1]
        run_case(case)
        torch.cuda.synchronize()
        for name, lib in order:
            _hip._lib = lib
            res = run_case(case)
            torch.cuda.synchronize()
            if rnd == 0:
                if case != "kernels":
                    d = out.view(torch.int32)
                    bits.setdefault(case, {})[name] = (int(d.sum(dtype=torch.int64)), int((d.to(torch.int64) * 31 % 1000003).sum()))
                continue
            for k, v in res.items():
                times.setdefault(k, {}).setdefault(name, []).append(v)
report = {"device": torch.cuda.get_device_name(0), "lut": a.lut, "frames": F, "rounds": a.rounds, "libs": [n for n, _ in libs], "metrics": {}, "bit_identical": {}}
for case, d in bits.items():
    base = d[libs[0][0]]
    report["bit_identical"][case] = {n: (v == base) for n, v in d.items()}
base_name = libs[0][0]
for k, per in times.items():
    row = {}
    for n, ts in per.items():
        med = statistics.median(ts)
        row[n] = {"median_ms": round(med, 4), "min_ms": round(min(ts), 4), "max_ms": round(max(ts), 4), "spread_pct": round(100 * (max(ts) - min(ts)) / med, 2)}
    b = row[base_name]
    for n in row:
        if n == base_name:
            continue
        delta = 100.0 * (row[n]["median_ms"] - b["median_ms"]) / b["median_ms"]
        noise = max(row[n]["spread_pct"], b["spread_pct"])
        row[n]["vs_" + base_name + "_pct"] = round(delta, 2)
        row[n]["verdict"] = ("inside the spread: no winner" if abs(delta) <= noise else (f"{n} faster" if delta < 0 else f"{base_name} faster"))
    if k.startswith("chain3") or k in ("grain_lut", "grain_sharpen"):
        for n in row:
            row[n]["gpix_s"] = round(px / row[n]["median_ms"] / 1e6, 2)
    report["metrics"][k] = row
    print("[ab]", k, json.dumps(row), flush=True)
print("[ab] bit_identical", json.dumps(report["bit_identical"]), flush=True)
if a.json:
    os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
    with open(a.json, "w") as fh:
        json.dump(report, fh, indent=1)
