"""CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Every function states the reference lines it follows (paths relative to
``/root/reference``).  The restatement keeps the reference's *operation order and
rounding points* (one fp32 rounding per tensor op, Python scalars rounded once
to fp32 when they meet an fp32 tensor), because that is what "parity" means for
the HIP kernels; it is organised differently from the reference (noise is an
explicit argument, border mode is an argument, stats are a separate step) so
that each stage can be checked in isolation.

All functions take / return ``torch.float32`` CPU tensors in ComfyUI's IMAGE
convention ``[F, H, W, C]`` unless noted.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Film grain
# --------------------------------------------------------------------------------------

#: per-channel gains applied to the raw normal noise (nodes.py:53-54; G is left at 1)
GRAIN_GAIN_R = 2.0
GRAIN_GAIN_B = 3.0


def grain_apply(x: torch.Tensor, noise: torch.Tensor, intensity: float, saturation_mix: float) -> torch.Tensor:
    """Arithmetic of one grain chunk given the raw N(0,1) ``noise`` (nodes.py:53-60;
    identical in VRGDG_LUTVideoTools.py:272-277 and
    VRGDG_StandaloneVideoEnhancerNodes.py:274-277).

    g_c = fl(fl(S*k_c*n_c) ...): the order is: scale R,B in place; gray = n_G (unscaled);
    g = S*n + T*gray with S=(float)s, T=(float)(1.0-s) [double subtraction first];
    out = clamp(x + g*I, 0, 1).
    """
    n = noise.clone()
    n[..., 0] *= GRAIN_GAIN_R
    n[..., 2] *= GRAIN_GAIN_B
    gray = n[..., 1:2].expand(*n.shape[:-1], 3)
    mixed = saturation_mix * n + (1.0 - saturation_mix) * gray
    return (x + mixed * intensity).clamp(0.0, 1.0)


def fast_film_grain(images, grain_intensity, saturation_mix, batch_size, noise_fn=None):
    """FastFilmGrain.apply_grain (nodes.py:41-66).  ``noise_fn(chunk_start, shape)`` supplies
    the N(0,1) tensor of each ``batch_size`` chunk (default: ``torch.randn`` from the global
    CPU generator, which is what ``torch.randn_like`` consumes on CPU)."""
    step = batch_size if batch_size > 0 else images.shape[0]  # nodes.py:46
    outs = []
    for i in range(0, images.shape[0], step):
        chunk = images[i:i + step]
        noise = torch.randn(chunk.shape, dtype=chunk.dtype) if noise_fn is None else noise_fn(i, tuple(chunk.shape))
        outs.append(grain_apply(chunk, noise, grain_intensity, saturation_mix))
    return torch.cat(outs, dim=0)


def film_grain_tensor(image, grain_intensity=0.04, saturation_mix=0.5, seed=None, noise=None):
    """_apply_film_grain_tensor (VRGDG_LUTVideoTools.py:262-277): clamps I and s to [0,1],
    optional seeded generator for the whole tensor."""
    intensity = max(0.0, min(1.0, float(grain_intensity)))
    saturation = max(0.0, min(1.0, float(saturation_mix)))
    if noise is None:
        gen = None
        if seed not in (None, ""):
            gen = torch.Generator(device=image.device)
            gen.manual_seed(int(seed))
        noise = torch.randn(image.shape, dtype=image.dtype, device=image.device, generator=gen)
    return grain_apply(image, noise, intensity, saturation)


def seeded_grain_frame_seed(seed: int, frame_start: int, offset: int) -> int:
    """Per-frame generator seed (VRGDG_StandaloneVideoEnhancerNodes.py:270)."""
    return (int(seed) + int(frame_start) + int(offset)) & 0x7FFFFFFF


def seeded_grain(images, intensity, saturation_mix, seed, frame_start, noise_fn=None):
    """_apply_seeded_grain (VRGDG_StandaloneVideoEnhancerNodes.py:262-278): one generator per
    frame seeded with (seed + frame_start + offset) & 0x7FFFFFFF; ``intensity <= 0`` is a no-op."""
    if intensity <= 0:
        return images
    frames = []
    for off in range(images.shape[0]):
        fseed = seeded_grain_frame_seed(seed, frame_start, off)
        if noise_fn is None:
            gen = torch.Generator(device=images.device)
            gen.manual_seed(fseed)
            n = torch.randn(images[off].shape, generator=gen, device=images.device, dtype=images.dtype)
        else:
            n = noise_fn(fseed, tuple(images[off].shape))
        frames.append(n)
    noise = torch.stack(frames, dim=0)
    return grain_apply(images, noise, intensity, saturation_mix)


# --------------------------------------------------------------------------------------
# 3D LUT
# --------------------------------------------------------------------------------------

def parse_cube_file(path: str) -> dict:
    """VRGDG_LUTS._parse_cube_file (VRGDG_IV_Adjustments.py:221-282).

    Returns ``{"size", "lut" [N,N,N,3] indexed [blue,green,red,rgb], "domain_min", "domain_max"}``.
    Rules kept: blank / ``#`` lines skipped; ``TITLE `` skipped; ``LUT_1D_SIZE`` -> ValueError;
    ``LUT_3D_SIZE n``; ``DOMAIN_MIN/MAX a b c``; every other line with exactly three
    whitespace-separated tokens is a data row (Python ``float`` -> fp32); lines with any other
    token count are ignored; count must equal N^3 * 3.
    """
    import os
    size = None
    dmin = np.zeros(3, dtype=np.float32)
    dmax = np.ones(3, dtype=np.float32)
    vals = []
    with open(path, "r", encoding="utf-8", errors="ignore") as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line[0] == "#":
                continue
            up = line.upper()
            if up.startswith("TITLE "):
                continue
            if up.startswith("LUT_1D_SIZE"):
                raise ValueError(f"1D LUTs are not supported: {os.path.basename(path)}")
            tok = line.split()
            if up.startswith("LUT_3D_SIZE"):
                if len(tok) != 2:
                    raise ValueError(f"Invalid LUT_3D_SIZE line in {path}")
                size = int(tok[1])
                continue
            if up.startswith("DOMAIN_MIN") or up.startswith("DOMAIN_MAX"):
                if len(tok) != 4:
                    raise ValueError(f"Invalid {tok[0]} line in {path}")
                arr = np.array([float(tok[1]), float(tok[2]), float(tok[3])], dtype=np.float32)
                if up.startswith("DOMAIN_MIN"):
                    dmin = arr
                else:
                    dmax = arr
                continue
            if len(tok) != 3:
                continue
            vals.extend(float(t) for t in tok)
    if size is None:
        raise ValueError(f"Missing LUT_3D_SIZE in {path}")
    want = size * size * size * 3
    if len(vals) != want:
        raise ValueError(f"Invalid LUT data length in {path}. Expected {want} floats, got {len(vals)}.")
    lut = torch.from_numpy(np.asarray(vals, dtype=np.float32).reshape(size, size, size, 3))
    return {"size": size, "lut": lut, "domain_min": torch.from_numpy(dmin), "domain_max": torch.from_numpy(dmax)}


def apply_cube_lut(image, lut, domain_min, domain_max):
    """VRGDG_LUTS._apply_cube_lut (VRGDG_IV_Adjustments.py:288-343).

    t = clamp((x-dmin)/max(dmax-dmin,1e-6),0,1); c = t*(N-1); i0=floor(c) (int64);
    i1=min(i0+1,N-1); f=c-i0; corners lut[b,g,r]; lerp order blue, green, red, each
    a*(1-f)+b*f; clamp(0,1); channels >= 3 pass through.
    """
    if image.ndim != 4 or image.shape[-1] < 3:
        raise ValueError("VRGDG_LUTS expects IMAGE input shaped like [batch, height, width, channels].")
    rgb = image[..., :3].to(torch.float32)
    span = torch.clamp(domain_max - domain_min, min=1e-6)
    t = torch.clamp((rgb - domain_min) / span, 0.0, 1.0)
    top = lut.shape[0] - 1
    c = t * top
    lo = torch.floor(c).long()
    hi = torch.clamp(lo + 1, max=top)
    frac = c - lo.float()
    r0, g0, b0 = lo[..., 0], lo[..., 1], lo[..., 2]
    r1, g1, b1 = hi[..., 0], hi[..., 1], hi[..., 2]
    fr, fg, fb = frac[..., 0:1], frac[..., 1:2], frac[..., 2:3]

    def along_blue(g, r):
        return lut[b0, g, r] * (1.0 - fb) + lut[b1, g, r] * fb

    def along_green(r):
        return along_blue(g0, r) * (1.0 - fg) + along_blue(g1, r) * fg

    out = torch.clamp(along_green(r0) * (1.0 - fr) + along_green(r1) * fr, 0.0, 1.0)
    if image.shape[-1] == 3:
        return out.to(image.dtype)
    full = image.clone()
    full[..., :3] = out.to(image.dtype)
    return full


def lut_blend_factor(strength) -> float:
    """blend = clamp(strength, 0, 10) / 10 in double (VRGDG_IV_Adjustments.py:355)."""
    return max(0.0, min(10.0, float(strength))) / 10.0


def apply_lut_with_strength(image, lut_data, strength):
    """VRGDG_LUTS.apply_lut minus device moves (VRGDG_IV_Adjustments.py:345-361), same as
    _apply_lut_tensor (VRGDG_LUTVideoTools.py:172-185)."""
    dmin = lut_data["domain_min"].to(image.dtype)
    dmax = lut_data["domain_max"].to(image.dtype)
    graded = apply_cube_lut(image, lut_data["lut"], dmin, dmax)
    blend = lut_blend_factor(strength)
    if blend <= 0.0:
        return image
    if blend < 1.0:
        return (image * (1.0 - blend)) + (graded * blend)
    return graded


# --------------------------------------------------------------------------------------
# 3x3 stencils (unsharp / laplacian / sobel)
# --------------------------------------------------------------------------------------

def _edge_padded(images: torch.Tensor) -> np.ndarray:
    arr = images.contiguous().numpy()
    return np.pad(arr, ((0, 0), (1, 1), (1, 1), (0, 0)), mode="edge")


def _taps(p):
    """The nine shifted views p[dy][dx] (dy,dx in 0..2) of an edge/zero padded NHWC array."""
    H = p.shape[1] - 2
    W = p.shape[2] - 2
    return [[p[:, dy:dy + H, dx:dx + W] for dx in range(3)] for dy in range(3)]


def unsharp(images, strength, use_gpu=False):
    """FastUnsharpSharpen.apply_unsharp (nodes.py:156-209).

    use_gpu=False (default): numpy, edge-replicate pad, nine-term row-major left-assoc sum, /9.0.
    use_gpu=True: avg_pool2d(k=3,s=1,p=1) == zero pad, raster-order sum of in-bounds taps, /9.
    Then out = clip(x + strength*(x-blur), 0, 1).
    """
    if use_gpu:
        x = images.permute(0, 3, 1, 2)
        blur = F.avg_pool2d(x, kernel_size=3, stride=1, padding=1)
        return (x + strength * (x - blur)).clamp(0.0, 1.0).permute(0, 2, 3, 1)
    img = images.contiguous().numpy()
    t = _taps(_edge_padded(images))
    acc = t[0][0] + t[0][1]
    for dy, dx in ((0, 2), (1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2)):
        acc = acc + t[dy][dx]
    blur = acc / 9.0
    out = img + strength * (img - blur)
    np.clip(out, 0.0, 1.0, out=out)
    return torch.from_numpy(out)


def laplacian(images, strength, use_gpu=False):
    """FastLaplacianSharpen.apply_laplacian (nodes.py:234-289).

    CPU: lap = W + N + S + E - 4*x (that order), replicate border, out = clip(x + s*lap).
    GPU flag: depthwise conv2d with [[0,-1,0],[-1,4,-1],[0,-1,0]], zero pad, out = clip(x + s*edges)
    (opposite sign convention -- reproduced, not fixed).
    """
    if use_gpu:
        x = images.permute(0, 3, 1, 2)
        C = x.shape[1]
        k = torch.tensor([[0, -1, 0], [-1, 4, -1], [0, -1, 0]], dtype=torch.float32).expand(C, 1, 3, 3)
        edges = F.conv2d(x, k, padding=1, groups=C)
        return (x + strength * edges).clamp(0.0, 1.0).permute(0, 2, 3, 1)
    img = images.contiguous().numpy()
    t = _taps(_edge_padded(images))
    lap = t[1][0] + t[0][1] + t[2][1] + t[1][2] - 4.0 * img
    out = img + strength * lap
    np.clip(out, 0.0, 1.0, out=out)
    return torch.from_numpy(out)


def sobel(images, strength, use_gpu=False):
    """FastSobelSharpen.apply_sobel (nodes.py:314-384).

    CPU: gx, gy in the reference's term order, edges = sqrt(gx*gx + gy*gy), replicate border.
    GPU flag: conv2d zero pad, edges = sqrt(gx*gx + gy*gy + 1e-6).
    """
    if use_gpu:
        x = images.permute(0, 3, 1, 2)
        C = x.shape[1]
        kx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float32).expand(C, 1, 3, 3)
        ky = torch.tensor([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], dtype=torch.float32).expand(C, 1, 3, 3)
        gx = F.conv2d(x, kx, padding=1, groups=C)
        gy = F.conv2d(x, ky, padding=1, groups=C)
        edges = torch.sqrt(gx * gx + gy * gy + 1e-6)
        return (x + strength * edges).clamp(0.0, 1.0).permute(0, 2, 3, 1)
    img = images.contiguous().numpy()
    t = _taps(_edge_padded(images))
    gx = (-t[0][0] - 2 * t[1][0] - t[2][0] + t[0][2] + 2 * t[1][2] + t[2][2])
    gy = (-t[0][0] - 2 * t[0][1] - t[0][2] + t[2][0] + 2 * t[2][1] + t[2][2])
    edges = np.sqrt(gx * gx + gy * gy)
    out = img + strength * edges
    np.clip(out, 0.0, 1.0, out=out)
    return torch.from_numpy(out)


# Explicit-order restatements of the zero-pad ("use_gpu") stencils.  torch's conv2d does not
# define its accumulation order (MIOpen / oneDNN pick their own); the HIP kernels use raster
# (kh,kw) order over the non-zero taps, which these functions state explicitly so that the
# kernels can be checked bit-for-bit, while the conv2d versions above are checked to a few ulp.

def _zero_padded(images: torch.Tensor) -> np.ndarray:
    arr = images.contiguous().numpy()
    return np.pad(arr, ((0, 0), (1, 1), (1, 1), (0, 0)), mode="constant")


def laplacian_zero_raster(images, strength):
    img = images.contiguous().numpy()
    t = _taps(_zero_padded(images))
    edges = (((-t[0][1]) - t[1][0]) + 4.0 * t[1][1]) - t[1][2] - t[2][1]
    out = img + np.float32(strength) * edges
    np.clip(out, 0.0, 1.0, out=out)
    return torch.from_numpy(out)


def sobel_zero_raster(images, strength):
    img = images.contiguous().numpy()
    t = _taps(_zero_padded(images))
    gx = ((((-t[0][0]) + t[0][2]) - 2.0 * t[1][0]) + 2.0 * t[1][2]) - t[2][0] + t[2][2]
    gy = ((((-t[0][0]) - 2.0 * t[0][1]) - t[0][2]) + t[2][0]) + 2.0 * t[2][1] + t[2][2]
    edges = np.sqrt(gx * gx + gy * gy + np.float32(1e-6))
    out = img + np.float32(strength) * edges
    np.clip(out, 0.0, 1.0, out=out)
    return torch.from_numpy(out)


# --------------------------------------------------------------------------------------
# kornia.color Lab transforms -- RESTATED, PARITY UNPINNED (kornia is an unpinned external
# dependency of the reference: requirements.txt:1, call sites nodes.py:98,108,115).
# Published algorithm of kornia.color.{rgb_to_lab, lab_to_rgb, rgb_to_linear_rgb,
# linear_rgb_to_rgb, rgb_to_xyz, xyz_to_rgb} (kornia >= 0.6), NCHW layout [..., 3, H, W].
# --------------------------------------------------------------------------------------

D65_WHITE = (0.95047, 1.0, 1.08883)
RGB2XYZ = ((0.412453, 0.357580, 0.180423),
           (0.212671, 0.715160, 0.072169),
           (0.019334, 0.119193, 0.950227))
XYZ2RGB = ((3.2404813432005266, -1.5371515162713185, -0.4985363261688878),
           (-0.9692549499965682, 1.8759900014898907, 0.0415559265582928),
           (0.0556466391351772, -0.2040413383665112, 1.0573110696453443))


def kornia_rgb_to_lab(image: torch.Tensor) -> torch.Tensor:
    lin = torch.where(image > 0.04045, torch.pow((image + 0.055) / 1.055, 2.4), image / 12.92)
    r, g, b = lin[..., 0, :, :], lin[..., 1, :, :], lin[..., 2, :, :]
    xyz = torch.stack([RGB2XYZ[i][0] * r + RGB2XYZ[i][1] * g + RGB2XYZ[i][2] * b for i in range(3)], -3)
    white = torch.tensor(D65_WHITE, device=xyz.device, dtype=xyz.dtype)[..., :, None, None]
    xn = torch.div(xyz, white)
    thr = 0.008856
    f = torch.where(xn > thr, torch.pow(xn.clamp(min=thr), 1 / 3.0), 7.787 * xn + 4.0 / 29.0)
    fx, fy, fz = f[..., 0, :, :], f[..., 1, :, :], f[..., 2, :, :]
    return torch.stack([(116.0 * fy) - 16.0, 500.0 * (fx - fy), 200.0 * (fy - fz)], dim=-3)


def kornia_lab_to_rgb(image: torch.Tensor, clip: bool = True) -> torch.Tensor:
    L, a, b_ = image[..., 0, :, :], image[..., 1, :, :], image[..., 2, :, :]
    fy = (L + 16.0) / 116.0
    fx = (a / 500.0) + fy
    fz = (fy - (b_ / 200.0)).clamp(min=0.0)
    f = torch.stack([fx, fy, fz], dim=-3)
    xyz = torch.where(f > 0.2068966, torch.pow(f, 3.0), (f - 4.0 / 29.0) / 7.787)
    white = torch.tensor(D65_WHITE, device=xyz.device, dtype=xyz.dtype)[..., :, None, None]
    xyz = xyz * white
    x, y, z = xyz[..., 0, :, :], xyz[..., 1, :, :], xyz[..., 2, :, :]
    lin = torch.stack([XYZ2RGB[i][0] * x + XYZ2RGB[i][1] * y + XYZ2RGB[i][2] * z for i in range(3)], dim=-3)
    thr = 0.0031308
    rgb = torch.where(lin > thr, 1.055 * torch.pow(lin.clamp(min=thr), 1 / 2.4) - 0.055, 12.92 * lin)
    if clip:
        rgb = torch.clamp(rgb, min=0.0, max=1.0)
    return rgb


# --------------------------------------------------------------------------------------
# Colour match
# --------------------------------------------------------------------------------------

def lab_stats(lab_nchw: torch.Tensor):
    """Per-(frame, channel) mean and unbiased std (+1e-5) over H*W (nodes.py:99-100, 109-110)."""
    mean = lab_nchw.mean(dim=[2, 3], keepdim=True)
    std = lab_nchw.std(dim=[2, 3], keepdim=True) + 1e-5
    return mean, std


def color_match_apply(lab_nchw, img_mean, img_std, ref_mean, ref_std, match_strength):
    """matched=(lab-mu)/sigma*sigma_ref+mu_ref ; blended=k*matched+(1-k)*lab ; Lab->RGB
    (nodes.py:112-115)."""
    matched = (lab_nchw - img_mean) / img_std * ref_std + ref_mean
    blended = match_strength * matched + (1.0 - match_strength) * lab_nchw
    return kornia_lab_to_rgb(blended)


def color_match(images, reference_image, match_strength, batch_size):
    """ColorMatchToReference.match_color (nodes.py:91-124).  Returns a contiguous NHWC tensor
    (the reference returns a permuted view of NCHW memory; values are identical)."""
    x = images.permute(0, 3, 1, 2)
    ref = reference_image.permute(0, 3, 1, 2)
    ref_mean, ref_std = lab_stats(kornia_rgb_to_lab(ref))
    outs = []
    for i in range(0, x.shape[0], batch_size):
        lab = kornia_rgb_to_lab(x[i:i + batch_size])
        mean, std = lab_stats(lab)
        outs.append(color_match_apply(lab, mean, std, ref_mean, ref_std, match_strength))
    out = torch.cat(outs, dim=0).clamp(0.0, 1.0)
    return out.permute(0, 2, 3, 1).contiguous()


# --------------------------------------------------------------------------------------
# 13-slider "Adjust" 
# --------------------------------------------------------------------------------------

ADJUST_FIELDS = {
    "temperature": (-100.0, 100.0), "tint": (-100.0, 100.0), "saturation": (-100.0, 100.0), "exposure": (-100.0, 100.0),
    "contrast": (-100.0, 100.0), "highlights": (-100.0, 100.0), "shadows": (-100.0, 100.0), "whites": (-100.0, 100.0),
    "blacks": (-100.0, 100.0), "sharpen": (0.0, 100.0), "clarity": (-100.0, 100.0), "vignette": (0.0, 100.0), "fade": (0.0, 100.0),
}


def normalize_adjust_settings(settings=None) -> dict:
    """_normalize_adjust_settings (VRGDG_LUTVideoTools.py:280-304): clamp every slider, bad values -> 0."""
    settings = settings if isinstance(settings, dict) else {}
    out = {"enabled": settings.get("enabled", True) is not False}
    for key, (lo, hi) in ADJUST_FIELDS.items():
        try:
            v = float(settings.get(key, 0.0))
        except Exception:
            v = 0.0
        out[key] = max(lo, min(hi, v))
    return out


def _luma(t, dim):
    r, g, b = t.narrow(dim, 0, 1), t.narrow(dim, 1, 1), t.narrow(dim, 2, 1)
    return (r * 0.2126) + (g * 0.7152) + (b * 0.0722)


def adjust_box_kernel(target: int, height: int, width: int) -> int:
    """kernel = min(target, largest odd <= H, largest odd <= W); < 3 disables the blur (:349-353)."""
    return min(int(target), height if height % 2 else height - 1, width if width % 2 else width - 1)


def adjust_tensor(image: torch.Tensor, settings=None, ieee_sqrt: bool = False) -> torch.Tensor:
    """_apply_adjust_tensor (VRGDG_LUTVideoTools.py:307-391), op for op: white balance shift, exposure, contrast,
    saturation, highlights / shadows / whites / blacks masks on the post-saturation luma, clarity (9x9 reflect box
    blur detail, mid-tone weighted), sharpen (3x3 replicate box blur detail x5), fade, vignette, clamp.

    ``ieee_sqrt``: this torch build's CPU ``torch.sqrt`` is not correctly rounded (it disagrees with the IEEE
    square root on ~0.55 % of fp32 inputs, by 1 ulp), while the device sqrt the reference gets on a GPU is.  With
    ``ieee_sqrt=True`` the vignette distance uses numpy's correctly rounded sqrt -- the only op that changes --
    which is what the HIP kernels are held to bit for bit; the default reproduces the reference on CPU exactly."""
    adj = normalize_adjust_settings(settings)
    source = image.clamp(0.0, 1.0)
    if not adj["enabled"]:
        return source
    out = source + torch.tensor(
        [adj["temperature"] / 400.0 - adj["tint"] / 900.0, adj["tint"] / 450.0, -adj["temperature"] / 400.0 - adj["tint"] / 900.0],
        dtype=source.dtype, device=source.device).view(1, 1, 1, 3)
    out = out * (2.0 ** (adj["exposure"] / 100.0))
    out = (out - 0.5) * (1.0 + adj["contrast"] / 100.0) + 0.5
    gray = _luma(out, 3).repeat(1, 1, 1, 3)
    out = gray + (out - gray) * (1.0 + adj["saturation"] / 100.0)
    luma = _luma(out, 3)
    out = out + torch.clamp((luma - 0.55) / 0.45, 0.0, 1.0) * (adj["highlights"] / 220.0)
    out = out + torch.clamp((0.45 - luma) / 0.45, 0.0, 1.0) * (adj["shadows"] / 220.0)
    out = out + torch.clamp((luma - 0.75) / 0.25, 0.0, 1.0) * (adj["whites"] / 240.0)
    out = out + torch.clamp((0.25 - luma) / 0.25, 0.0, 1.0) * (adj["blacks"] / 240.0)
    clarity = adj["clarity"] / 100.0
    sharpen = adj["sharpen"] / 100.0
    if abs(clarity) > 0.001 or sharpen > 0.001:
        x = out.permute(0, 3, 1, 2)
        H, W = int(x.shape[2]), int(x.shape[3])
        if abs(clarity) > 0.001:
            k = adjust_box_kernel(9, H, W)
            blur = x if k < 3 else F.avg_pool2d(F.pad(x, (k // 2,) * 4, mode="reflect"), kernel_size=k, stride=1)
            detail = x - blur
            mid = 1.0 - torch.clamp(torch.abs(_luma(x, 1) - 0.5) / 0.5, 0.0, 1.0)
            x = x + detail * clarity * 1.55 * (0.35 + mid * 0.65)
        if sharpen > 0.001:
            fine = F.avg_pool2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), kernel_size=3, stride=1)
            x = x + (x - fine) * sharpen * 5.0
        out = x.permute(0, 2, 3, 1)
    fade = adj["fade"] / 100.0
    if fade > 0.0:
        out = out * (1.0 - fade * 0.35) + fade * 0.18
    vig = adj["vignette"] / 100.0
    if vig > 0.0:
        H, W = out.shape[1], out.shape[2]
        yy = torch.linspace(-1.0, 1.0, H, dtype=out.dtype, device=out.device).view(1, H, 1, 1)
        xx = torch.linspace(-1.0, 1.0, W, dtype=out.dtype, device=out.device).view(1, 1, W, 1)
        d2 = (xx * xx) + (yy * yy)
        dist = torch.from_numpy(np.sqrt(d2.numpy())) if (ieee_sqrt and not d2.is_cuda) else torch.sqrt(d2)
        out = out * (1.0 - torch.clamp((dist - 0.35) / 1.05, 0.0, 1.0) * vig * 0.75)
    return out.clamp(0.0, 1.0)


# --------------------------------------------------------------------------------------
# uint8 BGR frames at the codec edge
# --------------------------------------------------------------------------------------

def frames_to_tensor(frames) -> torch.Tensor:
    """_frames_to_tensor (VRGDG_LUTVideoTools.py:736-743 == VRGDG_StandaloneVideoEnhancerNodes.py:311-316):
    BGR->RGB channel reversal (cv2.COLOR_BGR2RGB), stack, ``astype(float32) / 255.0``."""
    rgb = [np.ascontiguousarray(np.asarray(f)[..., ::-1]) for f in frames]
    return torch.from_numpy(np.stack(rgb, axis=0).astype(np.float32) / 255.0)


def tensor_to_frames(tensor: torch.Tensor):
    """_tensor_to_frames (:746-752 / :319-324): ``clip(x * 255.0, 0, 255).astype(uint8)`` (C truncation), then
    RGB->BGR per frame."""
    array = np.clip(tensor.detach().cpu().numpy() * 255.0, 0, 255).astype(np.uint8)
    return [np.ascontiguousarray(frame[..., ::-1]) for frame in array]


# --------------------------------------------------------------------------------------
# Opening colour match of a new clip: statistics -> 17^3 cube
# --------------------------------------------------------------------------------------

def image_stat_rgb(frame_u8: np.ndarray):
    """PIL.ImageStat.Stat(image).mean / .stddev of an HxWx3 uint8 frame, as ImageStat computes them: integer-valued
    sums (from the histogram) in double, ``mean = sum / n``, ``var = (sum2 - sum**2.0 / n) / n``, ``sqrt``."""
    a = np.asarray(frame_u8).reshape(-1, 3).astype(np.uint64)
    n = a.shape[0]
    mean, std = [], []
    for c in range(3):
        s = float(int(a[:, c].sum()))
        s2 = float(int((a[:, c] * a[:, c]).sum()))
        mean.append(s / n)
        std.append(math.sqrt((s2 - (s ** 2.0) / n) / n))
    return mean, std


def opening_match_terms(reference_stats, target_stats):
    """scales / offsets of VRGDG_WorkflowRunnerNodes.py:4386-4391: stddevs floored at 1, scale clamped to [0.25, 4]."""
    (rm, rs), (tm, ts) = reference_stats, target_stats
    rs = [max(1.0, float(v)) for v in rs]
    ts = [max(1.0, float(v)) for v in ts]
    scales = [max(0.25, min(4.0, rs[i] / ts[i])) for i in range(3)]
    offsets = [float(rm[i]) - float(tm[i]) * scales[i] for i in range(3)]
    return scales, offsets


def opening_match_cube_text(scales, offsets, size=17) -> str:
    """The ``.cube`` the reference hands to ffmpeg's lut3d (:4393-4405): red fastest, 8 decimals."""
    out = ['TITLE "VRGDG opening color match"\n', f"LUT_3D_SIZE {size}\nDOMAIN_MIN 0.0 0.0 0.0\nDOMAIN_MAX 1.0 1.0 1.0\n"]
    for blue in range(size):
        for green in range(size):
            for red in range(size):
                idx = (red, green, blue)
                v = [max(0.0, min(1.0, ((idx[i] / (size - 1)) * 255.0 * scales[i] + offsets[i]) / 255.0)) for i in range(3)]
                out.append(f"{v[0]:.8f} {v[1]:.8f} {v[2]:.8f}\n")
    return "".join(out)


def opening_match_weight(frame_index: int, fps: float, strength: float, fade_seconds: float) -> float:
    """ffmpeg blend weight ``max(0, min(1, strength * (1 - T / fade)))`` at T = frame_index / fps (:4407); the
    expression text carries strength and fade with six decimals, so those are the values ffmpeg evaluates."""
    s6, f6 = float(f"{strength:.6f}"), float(f"{fade_seconds:.6f}")
    return max(0.0, min(1.0, s6 * (1.0 - (frame_index / fps) / f6)))


# --------------------------------------------------------------------------------------
# Sequential composition used by the fused-chain parity tests
# --------------------------------------------------------------------------------------

def chain(images, noise=None, grain=None, lut=None, colormatch=None, sharpen=None):
    """grain -> LUT -> colour match -> unsharp applied one after the other, each stage exactly
    as its node would.  ``grain``=(I, s), ``lut``=(lut_data,
    strength), ``colormatch``=(reference_image, k), ``sharpen``=(strength, use_gpu)."""
    y = images
    if grain is not None:
        y = grain_apply(y, noise, grain[0], grain[1])
    if lut is not None:
        y = apply_lut_with_strength(y, lut[0], lut[1])
    if colormatch is not None:
        y = color_match(y, colormatch[0], colormatch[1], 1)
    if sharpen is not None:
        y = unsharp(y, sharpen[0], sharpen[1])
        if not y.is_contiguous():
            y = y.contiguous()
    return y
