"""float64 evaluations of the colour-match math (TEST INFRASTRUCTURE ONLY).

Used to *bound* fp32 results where bit-equality with the reference is not definable:
per-frame mean / unbiased std over millions of pixels is order dependent in fp32, and
``powf`` differs by an ulp between libraries (torch-CPU Sleef vs ROCm ocml).  The bar the tests
apply is SURVEY.md section 7 hard-part 4: |ours - truth64| <= |reference_fp32 - truth64| + tol.
Constants are the fp32 constants of oracle.restated (kornia's Python floats rounded to fp32
where they meet fp32 tensors) evaluated in float64.
"""
from __future__ import annotations

import numpy as np

from .restated import D65_WHITE, RGB2XYZ, XYZ2RGB


def _f32(v):
    return np.float64(np.float32(v))


def rgb_to_lab64(rgb_nhwc: np.ndarray) -> np.ndarray:
    """[...,3] float32/64 -> Lab float64, same formula as oracle.restated.kornia_rgb_to_lab."""
    x = rgb_nhwc.astype(np.float64)
    lin = np.where(x > _f32(0.04045), np.power((x + _f32(0.055)) / _f32(1.055), _f32(2.4)), x / _f32(12.92))
    r, g, b = lin[..., 0], lin[..., 1], lin[..., 2]
    xyz = [(_f32(RGB2XYZ[i][0]) * r + _f32(RGB2XYZ[i][1]) * g + _f32(RGB2XYZ[i][2]) * b) / _f32(D65_WHITE[i])
           for i in range(3)]
    thr = _f32(0.008856)
    f = [np.where(t > thr, np.power(np.maximum(t, thr), _f32(1 / 3.0)), _f32(7.787) * t + _f32(4.0 / 29.0))
         for t in xyz]
    return np.stack([_f32(116.0) * f[1] - 16.0, 500.0 * (f[0] - f[1]), 200.0 * (f[1] - f[2])], axis=-1)


def lab_to_rgb64(lab: np.ndarray) -> np.ndarray:
    L, a, b = lab[..., 0], lab[..., 1], lab[..., 2]
    fy = (L + 16.0) / 116.0
    fx = a / 500.0 + fy
    fz = np.maximum(fy - b / 200.0, 0.0)
    out = []
    for f, w in zip((fx, fy, fz), D65_WHITE):
        t = np.where(f > _f32(0.2068966), f * f * f, (f - _f32(4.0 / 29.0)) / _f32(7.787))
        out.append(t * _f32(w))
    x, y, z = out
    lin = [_f32(XYZ2RGB[i][0]) * x + _f32(XYZ2RGB[i][1]) * y + _f32(XYZ2RGB[i][2]) * z for i in range(3)]
    thr = _f32(0.0031308)
    rgb = [np.where(c > thr, _f32(1.055) * np.power(np.maximum(c, thr), _f32(1 / 2.4)) - _f32(0.055), _f32(12.92) * c)
           for c in lin]
    return np.clip(np.stack(rgb, axis=-1), 0.0, 1.0)


def lab_stats64(lab: np.ndarray):
    """lab [F,H,W,3] float64 -> mean [F,3], unbiased std + 1e-5 (as fp32 constant) [F,3]."""
    F_ = lab.shape[0]
    flat = lab.reshape(F_, -1, 3)
    n = flat.shape[1]
    mean = flat.mean(axis=1)
    var = ((flat - mean[:, None, :]) ** 2).sum(axis=1) / max(n - 1, 1) if n > 1 else np.full_like(mean, np.nan)
    return mean, np.sqrt(var) + _f32(1e-5)


def color_match64(images: np.ndarray, reference: np.ndarray, k: float) -> np.ndarray:
    """Whole colour match in float64 (per-frame statistics, reference batch 1 or F)."""
    lab = rgb_to_lab64(images)
    mu, sd = lab_stats64(lab)
    rmu, rsd = lab_stats64(rgb_to_lab64(reference))
    if rmu.shape[0] == 1:
        rmu = np.repeat(rmu, lab.shape[0], axis=0)
        rsd = np.repeat(rsd, lab.shape[0], axis=0)
    m = (lab - mu[:, None, None, :]) / sd[:, None, None, :] * rsd[:, None, None, :] + rmu[:, None, None, :]
    blended = _f32(k) * m + _f32(1.0 - k) * lab
    return lab_to_rgb64(blended)
