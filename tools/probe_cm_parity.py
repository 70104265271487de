"""Colour-match parity probe (GPU box): what does torch-ROCm compute for each op of the reference's colour match on
this GPU, and how far are the HIP kernels from it -- per stage, in ulps.  Writes gpurun_out/cm_parity_<tag>.json.

    python tools/probe_cm_parity.py [tag]

The "device oracle" is oracle/restated.py (kornia's Lab formulas + nodes.py:91-124) evaluated by torch ON THE GPU: that is
what the reference executes when ComfyUI runs ColorMatchToReference on this box.  oracle/ is used here as the checker only.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_vrgamedevgirl_amd import _hip, ops  # noqa: E402
from oracle import restated as R, truth64  # noqa: E402

dev = torch.device("cuda", 0)
OUT = {}


def bits(t: torch.Tensor) -> torch.Tensor:
    """monotone integer image of fp32 values (ulp distance = difference of images)"""
    i = t.contiguous().view(torch.int32).to(torch.int64)
    return torch.where(i < 0, -(i & 0x7FFFFFFF), i)


def ulp_stats(got: torch.Tensor, want: torch.Tensor, unit_scale=False):
    """max / p99 / mean distance.  unit_scale: in units of ulp(1.0) = 2^-23 of the absolute difference (for [0,1] outputs,
    where the integer distance explodes next to 0); otherwise integer ulp distance."""
    if unit_scale:
        d = (got.double() - want.double()).abs() / 2.0 ** -23
    else:
        d = (bits(got) - bits(want)).abs().double()
    d = d.flatten()
    nan_mismatch = int((torch.isnan(got) != torch.isnan(want)).sum())
    d = d[~torch.isnan(d)]
    k = max(int(d.numel() * 0.99) - 1, 0)
    return {"max": float(d.max()), "p99": float(d.kthvalue(k + 1).values), "mean": float(d.mean()),
            "differ_frac": float((d > 0).double().mean()), "n": int(d.numel()), "nan_mismatch": nan_mismatch}


def dbg(x: torch.Tensor, op: int, y: float = 0.0, triples=False) -> torch.Tensor:
    x = x.contiguous()
    out = torch.empty_like(x)
    n = x.numel() // (3 if triples else 1)
    _hip.check(_hip.lib().vrg_debug_cm_math(_hip.ptr(x), _hip.ptr(out), n, op, float(np.float32(y)), _hip.current_stream()), "dbg")
    return out


def all_floats(lo: float, hi: float) -> torch.Tensor:
    """every fp32 value in [lo, hi] (positive range)"""
    a = int(np.float32(lo).view(np.int32))
    b = int(np.float32(hi).view(np.int32))
    return torch.arange(a, b + 1, dtype=torch.int32, device=dev).view(torch.float32)


# ------------------------------------------------------------------------------------------ 1. element-wise pieces
def pieces():
    res = {}
    # torch.pow(x, y) on the device vs ocml powf compiled by this hipcc (and vs the fast policy's powers)
    for name, y, lo, hi in (("pow(x,2.4)", 2.4, 2.0 ** -12, 2.0), ("pow(x,1/2.4)", 1 / 2.4, 0.0031308, 4.0), ("pow(x,1/3)", 1 / 3.0, 0.008856, 4.0)):
        x = all_floats(lo, hi)
        want = torch.pow(x, y)
        got = dbg(x, 0, y)
        r = {"inputs": int(x.numel()), "ocml_vs_torch": ulp_stats(got, want), "bit_equal": bool(torch.equal(got, want))}
        if name == "pow(x,1/3)":
            r["fast_vs_torch"] = ulp_stats(dbg(x, 4), want)
        else:
            xs = x[x >= 0.0625] if name == "pow(x,2.4)" else x
            r["fast_vs_torch"] = ulp_stats(dbg(xs, 3, y), torch.pow(xs, y))
        # correctly rounded power through fp64 for the error of both against the truth
        t64 = torch.pow(x.double(), float(np.float32(y)))
        for tag, v in (("torch", want), ("ocml", got)):
            e = ((v.double() - t64).abs() / (t64.abs() * 2.0 ** -23)).max()
            r[f"{tag}_max_err_ulp_vs_fp64"] = float(e)
        res[name] = r
        print("[cm]", name, r, flush=True)
    # negative / zero / special bases (the where() of kornia evaluates pow on every element)
    sp = torch.tensor([0.0, -0.0, -0.5, -1.0, 1.0, float("inf"), float("nan"), 1e-38, 1e-45, -1e-30, 3.0e38], device=dev)
    for y in (2.4, 1 / 2.4, 1 / 3.0):
        a, b = dbg(sp, 0, y), torch.pow(sp, y)
        res[f"special_{y:.4f}"] = bool(torch.equal(a.view(torch.int32), b.view(torch.int32)) or
                                       torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)))
    # x / python scalar on the device = x * fl(1/c) ?   x / tensor = IEEE ?
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.cat([torch.rand(1 << 24, generator=g, device=dev) * 2 - 0.5, all_floats(2.0 ** -10, 2.0 ** -9), torch.randn(1 << 22, generator=g, device=dev) * 100])
    xn = x.cpu().numpy()
    for c in (1.055, 12.92, 116.0, 500.0, 200.0, 7.787):
        want = x / c
        c32 = np.float32(c)
        recip = torch.from_numpy(xn * (np.float32(1.0) / c32))
        recip_d = torch.from_numpy(xn * np.float32(1.0 / c))                    # reciprocal of the Python double, rounded once
        ieee = torch.from_numpy(xn / c32)
        res[f"div_scalar_{c}"] = {"torch_eq_x_times_fl32(1/fl32(c))": bool(torch.equal(want.cpu(), recip)),
                                   "torch_eq_x_times_fl32(1/c_double)": bool(torch.equal(want.cpu(), recip_d)),
                                   "torch_eq_ieee": bool(torch.equal(want.cpu(), ieee))}
    for c in (0.95047, 1.08883, 3.7, 1e-5 + 0.4):
        ct = torch.full((1,), c, dtype=torch.float32, device=dev)
        want = x / ct
        ieee = torch.from_numpy(xn / np.float32(c))
        res[f"div_tensor_{c}"] = {"torch_eq_ieee": bool(torch.equal(want.cpu(), ieee)), "kernel_op2_eq_torch": bool(torch.equal(dbg(x, 2, c), want))}
    # mul/add of python scalars: plain fp32 ops (sanity)
    res["mul_scalar_plain"] = bool(torch.equal((x * 7.787).cpu(), torch.from_numpy(xn * np.float32(7.787))))
    res["cube_is_xx_x"] = bool(torch.equal(torch.pow(x, 3.0).cpu(), torch.from_numpy((xn * xn) * xn)))
    print("[cm] div/mul", {k: v for k, v in res.items() if k.startswith(("div", "mul", "cube", "special"))}, flush=True)
    return res


# ------------------------------------------------------------------------------------------ 2. Lab transforms
def test_image(F, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((F, H, W, 3), generator=g)
    x[0, 0, :8, :] = torch.tensor([0.0, 1.0, 0.04045, 0.040450003, 0.5, 1e-6, 0.0031308, 0.9999999]).view(8, 1)
    return x


def lab_transforms():
    res = {}
    x = test_image(4, 540, 960, 11).to(dev)
    nchw = x.permute(0, 3, 1, 2)
    want_lab = R.kornia_rgb_to_lab(nchw).permute(0, 2, 3, 1).contiguous()
    got_lab = dbg(x, 5, triples=True)
    got_lab_fast = dbg(x, 7, triples=True)
    res["rgb_to_lab.device_vs_torch_device"] = {"bit_equal": bool(torch.equal(got_lab, want_lab)), **ulp_stats(got_lab, want_lab)}
    res["rgb_to_lab.fast_vs_torch_device"] = ulp_stats(got_lab_fast, want_lab)
    lab64 = torch.from_numpy(truth64.rgb_to_lab64(x.cpu().numpy())).to(dev)
    for tag, v in (("torch_device", want_lab), ("hip_device", got_lab), ("hip_fast", got_lab_fast)):
        res[f"rgb_to_lab.{tag}_abs_err_vs_fp64"] = float((v.double() - lab64).abs().max())
    cpu_lab = R.kornia_rgb_to_lab(nchw.cpu()).permute(0, 2, 3, 1).contiguous().to(dev)
    res["rgb_to_lab.torch_cpu_vs_torch_device"] = ulp_stats(cpu_lab, want_lab)
    res["rgb_to_lab.fast_vs_torch_cpu"] = ulp_stats(got_lab_fast, cpu_lab)
    # inverse on in-gamut and out-of-gamut Lab values
    g = torch.Generator(device=dev).manual_seed(5)
    lab = want_lab.clone()
    lab[2:] = lab[2:] * (1.0 + 0.3 * torch.randn(lab[2:].shape, generator=g, device=dev))      # perturbed: clamps / out of gamut
    want_rgb = R.kornia_lab_to_rgb(lab.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    got_rgb = dbg(lab, 6, triples=True)
    res["lab_to_rgb.device_vs_torch_device"] = {"bit_equal": bool(torch.equal(got_rgb, want_rgb)), **ulp_stats(got_rgb, want_rgb, unit_scale=True)}
    res["lab_to_rgb.fast_vs_torch_device"] = ulp_stats(dbg(lab, 8, triples=True), want_rgb, unit_scale=True)
    cpu_rgb = R.kornia_lab_to_rgb(lab.permute(0, 3, 1, 2).cpu()).permute(0, 2, 3, 1).contiguous().to(dev)
    res["lab_to_rgb.torch_cpu_vs_torch_device"] = ulp_stats(cpu_rgb, want_rgb, unit_scale=True)
    for k, v in res.items():
        print("[cm]", k, v, flush=True)
    return res


# ------------------------------------------------------------------------------------------ 3. statistics
def statistics():
    res = {}
    for name, x in (("4x540x960", test_image(4, 540, 960, 21)), ("2x2160x3840", test_image(2, 2160, 3840, 22)), ("3x37x53", test_image(3, 37, 53, 23))):
        xd = x.to(dev)
        lab = R.kornia_rgb_to_lab(xd.permute(0, 3, 1, 2))
        t_mean, t_std = R.lab_stats(lab)                                           # torch reductions on the device (fp32)
        t_mean, t_std = t_mean.flatten(1), t_std.flatten(1)
        ms = ops.finalize_stats(ops.lab_stats(xd))                                 # ours: fp64 sums -> fp32
        lab64 = lab.double()
        mu64 = lab64.mean(dim=[2, 3])
        sd64 = lab64.std(dim=[2, 3]) + float(np.float32(1e-5))
        c_mean, c_std = R.lab_stats(R.kornia_rgb_to_lab(x.permute(0, 3, 1, 2)))     # the reference on the CPU
        c_mean, c_std = c_mean.flatten(1).to(dev), c_std.flatten(1).to(dev)

        def u(a, b):
            return float(((a.double() - b.double()).abs() / (b.double().abs() * 2.0 ** -23)).max())
        res[name] = {"ours_vs_truth_ulp": {"mean": u(ms[..., 0], mu64), "std": u(ms[..., 1], sd64)},
                     "torch_device_vs_truth_ulp": {"mean": u(t_mean, mu64), "std": u(t_std, sd64)},
                     "torch_cpu_vs_truth_ulp": {"mean": u(c_mean, mu64), "std": u(c_std, sd64)},
                     "ours_vs_torch_device_ulp": {"mean": u(ms[..., 0], t_mean), "std": u(ms[..., 1], t_std)},
                     "ours_bits_eq_round_of_fp64": bool(torch.equal(ms[..., 0], mu64.float()) and torch.equal(ms[..., 1], sd64.float()))}
        print("[cm] stats", name, res[name], flush=True)
    return res


# ------------------------------------------------------------------------------------------ 4. end to end
def end_to_end():
    res = {}
    for name, (F, H, W), k in (("4x540x960 k=1", (4, 540, 960), 1.0), ("3x270x480 k=0.35", (3, 270, 480), 0.35), ("2x2160x3840 k=1", (2, 2160, 3840), 1.0)):
        x = test_image(F, H, W, 31)
        ref = test_image(1, 300, 400, 32) * 0.7 + 0.1
        xd, refd = x.to(dev), ref.to(dev)
        want = R.color_match(xd, refd, k, 1)                                   # device oracle end to end (torch statistics)
        r = {}
        for mode in ("device", "fast"):
            got = ops.color_match(xd, refd, k, cm_math=mode)
            r[f"{mode}.vs_device_oracle"] = ulp_stats(got, want, unit_scale=True)
        # injected statistics: torch's own device statistics fed to our apply pass -> element-wise path only
        lab = R.kornia_rgb_to_lab(xd.permute(0, 3, 1, 2))
        mu, sd = R.lab_stats(lab)
        rmu, rsd = R.lab_stats(R.kornia_rgb_to_lab(refd.permute(0, 3, 1, 2)))
        ims = torch.stack([mu.flatten(1), sd.flatten(1)], dim=-1).contiguous()
        rms = torch.stack([rmu.flatten(1), rsd.flatten(1)], dim=-1).contiguous()
        for mode in ("device", "fast"):
            got = ops.colormatch_apply(xd, ims, rms, k, cm_math=mode)
            r[f"{mode}.injected_torch_stats"] = {"bit_equal": bool(torch.equal(got, want)), **ulp_stats(got, want, unit_scale=True)}
        # our statistics fed to the ORACLE's element-wise path: isolates the statistics difference
        oms = ops.finalize_stats(ops.lab_stats(xd))
        orms = ops.finalize_stats(ops.lab_stats(refd))
        o_mu, o_sd = oms[..., 0].view(F, 3, 1, 1), oms[..., 1].view(F, 3, 1, 1)
        r_mu, r_sd = orms[..., 0].view(1, 3, 1, 1), orms[..., 1].view(1, 3, 1, 1)
        alt = R.color_match_apply(lab, o_mu, o_sd, r_mu, r_sd, k).clamp(0, 1).permute(0, 2, 3, 1).contiguous()
        r["oracle_with_our_stats.vs_device_oracle"] = ulp_stats(alt, want, unit_scale=True)
        r["device.vs_oracle_with_our_stats"] = {"bit_equal": bool(torch.equal(ops.color_match(xd, refd, k, cm_math="device"), alt))}
        # the reference on the CPU (Sleef powf, IEEE division) against the reference on the device
        if H <= 540:
            cpu = R.color_match(x, ref, k, 1).to(dev)
            r["torch_cpu_oracle.vs_device_oracle"] = ulp_stats(cpu, want, unit_scale=True)
            r["fast.vs_cpu_oracle"] = ulp_stats(ops.color_match(xd, refd, k, cm_math="fast"), cpu, unit_scale=True)
            t64 = torch.from_numpy(truth64.color_match64(x.numpy(), ref.numpy(), k)).to(dev)
            for tag, v in (("device_oracle", want), ("cpu_oracle", cpu), ("hip_device", ops.color_match(xd, refd, k, cm_math="device")),
                           ("hip_fast", ops.color_match(xd, refd, k, cm_math="fast"))):
                r[f"{tag}.abs_err_vs_fp64_in_unit_ulps"] = float((v.double() - t64).abs().max() / 2.0 ** -23)
        res[name] = r
        for kk, vv in r.items():
            print("[cm] e2e", name, kk, vv, flush=True)
    return res


# ------------------------------------------------------------------------------------------ 5. zero-border stencils (conv2d / avg_pool2d on the device)
def zero_border_stencils():
    import torch.nn.functional as F
    res = {}
    x = test_image(2, 270, 480, 41).to(dev)
    nchw = x.permute(0, 3, 1, 2).contiguous()
    # nodes.py:248-257 laplacian (use_gpu): depthwise conv2d, zero padding
    kl = torch.tensor([[0, -1, 0], [-1, 4, -1], [0, -1, 0]], dtype=torch.float32, device=dev).view(1, 1, 3, 3).repeat(3, 1, 1, 1)
    edges = F.conv2d(nchw, kl, padding=1, groups=3)
    want = (nchw + 0.7 * edges).clamp(0, 1).permute(0, 2, 3, 1).contiguous()
    got = ops.stencil3x3(x, "laplacian", 0.7, zero_border=True)
    res["laplacian_zero.vs_conv2d_device"] = {"bit_equal": bool(torch.equal(got, want)), **ulp_stats(got, want, unit_scale=True)}
    kx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float32, device=dev).view(1, 1, 3, 3).repeat(3, 1, 1, 1)
    ky = torch.tensor([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], dtype=torch.float32, device=dev).view(1, 1, 3, 3).repeat(3, 1, 1, 1)
    gx, gy = F.conv2d(nchw, kx, padding=1, groups=3), F.conv2d(nchw, ky, padding=1, groups=3)
    ed = torch.sqrt(gx ** 2 + gy ** 2 + 1e-6)
    want = (nchw + 0.7 * ed).clamp(0, 1).permute(0, 2, 3, 1).contiguous()
    got = ops.stencil3x3(x, "sobel", 0.7, zero_border=True)
    res["sobel_zero.vs_conv2d_device"] = {"bit_equal": bool(torch.equal(got, want)), **ulp_stats(got, want, unit_scale=True)}
    blur = F.avg_pool2d(nchw, kernel_size=3, stride=1, padding=1)
    want = (nchw + 0.7 * (nchw - blur)).clamp(0, 1).permute(0, 2, 3, 1).contiguous()
    got = ops.stencil3x3(x, "unsharp", 0.7, zero_border=True)
    res["unsharp_zero.vs_avg_pool2d_device"] = {"bit_equal": bool(torch.equal(got, want)), **ulp_stats(got, want, unit_scale=True)}
    # the CPU conv2d of the same torch build, for reference
    edges_c = F.conv2d(nchw.cpu(), kl.cpu(), padding=1, groups=3)
    want_c = (nchw.cpu() + 0.7 * edges_c).clamp(0, 1).permute(0, 2, 3, 1).contiguous().to(dev)
    res["laplacian_zero.conv2d_cpu_vs_conv2d_device"] = ulp_stats(want_c, (nchw + 0.7 * edges).clamp(0, 1).permute(0, 2, 3, 1).contiguous(), unit_scale=True)
    # raw device conv2d / sqrt outputs of small frames for the offline summation-order search (tools/conv_order_search.py)
    cap = {}
    for tag, (Fn, H, W) in (("s", (2, 24, 40)), ("m", (1, 270, 480)), ("l", (1, 1080, 1920))):
        xi = test_image(Fn, H, W, 50 + H).to(dev)
        n2 = xi.permute(0, 3, 1, 2).contiguous()
        cap[f"{tag}_x"] = n2[:, :, :26, :42].cpu().numpy()
        cap[f"{tag}_lap"] = F.conv2d(n2, kl, padding=1, groups=3)[:, :, :24, :40].cpu().numpy()
        cap[f"{tag}_gx"] = F.conv2d(n2, kx, padding=1, groups=3)[:, :, :24, :40].cpu().numpy()
        cap[f"{tag}_gy"] = F.conv2d(n2, ky, padding=1, groups=3)[:, :, :24, :40].cpu().numpy()
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "conv_capture.npz"), **cap)
    for k, v in res.items():
        print("[cm]", k, v, flush=True)
    return res


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    for name, fn in (("pieces", pieces), ("lab_transforms", lab_transforms), ("statistics", statistics), ("end_to_end", end_to_end),
                     ("zero_border_stencils", zero_border_stencils)):
        try:
            OUT[name] = fn()
        except Exception as exc:      # keep the other sections
            import traceback
            OUT[name] = {"error": f"{type(exc).__name__}: {exc}", "trace": traceback.format_exc()[-1500:]}
            print("[cm] section", name, "FAILED", exc, flush=True)
    OUT["device"] = torch.cuda.get_device_properties(0).name
    OUT["torch"] = torch.__version__
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"cm_parity_{tag}.json"), "w") as fh:
        json.dump(OUT, fh, indent=1)
    print("[cm] wrote", f"gpurun_out/cm_parity_{tag}.json")


if __name__ == "__main__":
    main()
