"""Tensor-level operators over device-resident fp32 ``[F, H, W, C]`` frames.

Each function enqueues hand-written HIP kernels (libvrgdg_hip.so, C ABI in include/vrgdg_hip.h) on the
current torch HIP stream and returns a new tensor; inputs are never modified.  Scalars that the reference
computes in Python doubles are computed here in double and rounded to fp32 exactly once, as torch does
when a Python float meets an fp32 tensor.

PyTorch is used for device memory, streams and the generator state only -- there is no torch arithmetic
on the data path and no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import functools
import math
import os
import threading
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _hip, rng

_F3 = C.c_float * 3

#: Arithmetic policy of the Lab transforms / statistics transfer (include/vrgdg_hip.h, enum vrg_cm_math).
#: "device" (default): each element-wise op is the one torch-ROCm runs for it on this GPU (bit-equal to the reference's
#: colour match executed on the MI355X, given the same statistics); "fast": table-driven powers, a few ulp away, 1.2-1.4x faster.
CM_MATH = {"device": _hip.CM_MATH_DEVICE, "fast": _hip.CM_MATH_FAST}


def default_cm_math() -> int:
    name = os.environ.get("VRGDG_CM_MATH", "device").strip().lower()
    if name not in CM_MATH:
        raise ValueError(f"VRGDG_CM_MATH must be one of {sorted(CM_MATH)}, got {name!r}")
    return CM_MATH[name]


def _cm_math(value) -> int:
    if value is None:
        return default_cm_math()
    if isinstance(value, str):
        return CM_MATH[value]
    return int(value)


#: Where the colour statistics come from.  "device" (default with the device arithmetic): torch-ROCm's own fp32 reductions replayed
#: bit for bit (csrc/vrg_torch_stats.hip) -- with it the whole colour match equals the reference run on this GPU; the value depends on
#: the node's batch_size like the reference's does.  "fp64": fp64-accumulated (n, mean, M2), rounded once -- closer to the exact
#: statistics than either reference, independent of batch_size, mergeable across GPUs (sharding.py), and what the fast policy uses.
CM_STATS = ("device", "fp64")


_NO_DEVICE_STATS_WARNED = set()
_STATE_LOCK = threading.RLock()     # re-entrant: HipEvent.__del__ takes it and may run in a GC pass triggered while it is held          # guards the small per-process caches below (first-use races between host threads)


def device_stats_supported(device) -> bool:
    """Can csrc/vrg_torch_stats.hip replay torch's reductions on this GPU?  torch's setReduceConfig takes another branch
    (force_splitting_output) on devices with fewer than 100 CUs -- e.g. one CPX partition of an MI355X; vrg_lab_stats_torch_f32 returns
    VRG_ERR_UNSUPPORTED there."""
    return torch.cuda.get_device_properties(device).multi_processor_count >= 100


def _cm_stats(value, cm_math, device=None) -> str:
    """Resolve the statistics policy.  An explicit value is taken as it is; the DEFAULT ("device" with the device arithmetic) falls
    back to "fp64" with a one-time warning on a GPU whose reduction geometry is not replayed (`device` given and unsupported)."""
    if value is None:
        value = os.environ.get("VRGDG_CM_STATS", "").strip().lower() or ("device" if _cm_math(cm_math) == _hip.CM_MATH_DEVICE else "fp64")
        if value == "device" and device is not None and torch.device(device).type == "cuda" and not device_stats_supported(device):
            key = torch.device(device).index
            with _STATE_LOCK:
                first = key not in _NO_DEVICE_STATS_WARNED
                _NO_DEVICE_STATS_WARNED.add(key)
            if first:
                import warnings
                warnings.warn(f"comfyui-vrgamedevgirl_amd: {torch.cuda.get_device_name(device)} reports "
                              f"{torch.cuda.get_device_properties(device).multi_processor_count} CUs; torch's reduction geometry is replayed for "
                              "devices with >= 100 CUs only -- colour statistics fall back to the fp64 form (a few ulp from the reference "
                              "run on this GPU instead of bit-equal to it)", RuntimeWarning)
            value = "fp64"
    if value not in CM_STATS:
        raise ValueError(f"cm_stats must be one of {CM_STATS}, got {value!r}")
    return value


def _on_device(fn):
    """Run `fn` with the device of its first tensor argument current: the kernels are enqueued on torch's current
    stream of the current device and the C side asks hipGetDevice(), so a tensor on cuda:1 while cuda:0 is current
    would otherwise get device-1 pointers on a device-0 stream."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        first = next((a for a in args if isinstance(a, torch.Tensor)), None)
        if first is None:
            first = next((a for a in kwargs.values() if isinstance(a, torch.Tensor)), None)
        if first is not None and first.is_cuda and first.device.index != torch.cuda.current_device():
            with torch.cuda.device(first.device):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapped


def _f32(v: float) -> float:
    """Python double -> nearest fp32 (as a Python float), the rounding torch applies to a scalar operand."""
    return float(np.float32(v))


def _check_frames(t: torch.Tensor, name="images", channels: Optional[int] = None, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or t.ndim != 4:
        raise ValueError(f"{name} must be a [frames, height, width, channels] tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {str(dtype).replace('torch.', '')}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (ops.* are device-resident; nodes.* move data for you)")
    if channels is not None and t.shape[-1] != channels:
        raise ValueError(f"{name} must have {channels} channels, got {t.shape[-1]}")
    return t if t.is_contiguous() else t.contiguous()


def _check_side(t: torch.Tensor, name: str, like: torch.Tensor, shape_tail=None, dtype=torch.float32):
    """A small device-resident operand of a kernel (statistics rows, LUT record table): the kernels read it through a raw
    pointer, so it must be contiguous, of the stated type and on the frames' GPU."""
    if not isinstance(t, torch.Tensor) or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous {str(dtype).replace('torch.', '')} tensor")
    if t.device != like.device:
        raise RuntimeError(f"{name} lives on {t.device}, the frames on {like.device}")
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"{name} must be shaped [n, {', '.join(str(v) for v in shape_tail)}], got {tuple(t.shape)}")
    return t


def _out_like(x: torch.Tensor, out: Optional[torch.Tensor]) -> torch.Tensor:
    """The destination of an operator: a fresh tensor, or the caller's (contiguous, shaped and typed like the frames, same device, not
    the frames themselves -- the stencils read neighbours of what they write)."""
    if out is None:
        return torch.empty_like(x)
    if out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous() or out.device != x.device:
        raise ValueError("out must be a contiguous tensor shaped and typed like the frames on the same device")
    if out.data_ptr() == x.data_ptr() and x.numel():
        raise ValueError("out must not alias the input frames")
    return out


# ------------------------------------------------------------------------------------------------
# noise descriptors
# ------------------------------------------------------------------------------------------------

@dataclass
class NoisePlan:
    """How the frames of a call map onto torch.randn calls: `chunk_frames` frames per call."""
    chunk_frames: int
    stream: rng.ChunkedStream
    chunk0: int = 0

    def desc(self, chunk0: Optional[int] = None) -> _hip.NoiseDesc:
        s = self.stream
        return _hip.NoiseDesc(seed0=s.seed, seed_stride=s.seed_stride, offset0=s.offset0, offset_stride=s.offset_stride,
                              chunk0=self.chunk0 if chunk0 is None else chunk0, chunk_frames=self.chunk_frames,
                              grid_threads=s.grid_threads)


def oversize_chunks(frames: int, frame_numel: int, chunk_frames: int) -> bool:
    """Does the reference's chunk loop draw a torch.randn that ATen has to split (more than 2^29 elements)?"""
    step = min(chunk_frames if chunk_frames > 0 else frames, frames)
    return step * frame_numel > rng.MAX_CHUNK_NUMEL


def plan_noise(frames: int, frame_numel: int, chunk_frames: int, device, generator=None):
    """Reserve the generator range FastFilmGrain's chunk loop would consume (nodes.py:46-51).
    Returns (plan for the full chunks or None, plan for the ragged tail chunk or None, n_full_chunks)."""
    step = chunk_frames if chunk_frames > 0 else frames
    step = min(step, frames)
    n_full, tail = divmod(frames, step)
    main = tail_plan = None
    if n_full:
        main = NoisePlan(step, rng.reserve(step * frame_numel, n_full, device, generator))
    if tail:
        tail_plan = NoisePlan(tail, rng.reserve(tail * frame_numel, 1, device, generator))
    return main, tail_plan, n_full


def slice_plans(plans, first_frame: int, frames: int):
    """The part of a whole-batch reservation (`plan_noise`) that covers frames [first_frame, first_frame + frames): what `plan_noise` would
    have returned for that piece had the pieces been reserved one after the other -- a batch's chunks are consecutive generator ranges, so
    reserving it at once (when the node is called) and slicing (when a piece runs) consumes the generator exactly as the reference's
    chunk loop does.  `first_frame` must be a multiple of the chunk size."""
    main, tail, n_full = plans
    step = main.chunk_frames if main is not None else (tail.chunk_frames if tail is not None else 1)
    full_end = n_full * step if main is not None else 0
    if first_frame % step and first_frame < full_end:
        raise ValueError("a piece of a grain batch must start on a noise chunk")
    end = first_frame + frames
    sub_main, sub_n = None, 0
    if main is not None and first_frame < full_end:
        sub_n = (min(end, full_end) - first_frame) // step
        if sub_n:
            sub_main = NoisePlan(step, main.stream, main.chunk0 + first_frame // step)
    covered = first_frame + sub_n * step if first_frame < full_end else first_frame
    sub_tail = None
    if covered < end:
        if tail is None or covered != full_end or end - covered != tail.chunk_frames:
            raise ValueError("a piece of a grain batch must be made of whole noise chunks")
        sub_tail = tail
    return sub_main, sub_tail, sub_n


# ------------------------------------------------------------------------------------------------
# grain
# ------------------------------------------------------------------------------------------------

def torch_stream_noise(frames: int, frame_numel: int, plan: NoisePlan, device) -> torch.Tensor:
    """The raw N(0,1) stream of `frames` frames under `plan` (test / debug entry)."""
    out = torch.empty((frames, frame_numel), dtype=torch.float32, device=device)
    d = plan.desc()
    _hip.check(_hip.lib().vrg_noise_f32(_hip.ptr(out), frames, frame_numel, C.byref(d), _hip.current_stream()), "vrg_noise_f32")
    return out


def _grain_call(x, out, f0, nf, plan: NoisePlan, I32, S32, T32):
    F, H, W, _ = x.shape
    fe = H * W * 3
    d = plan.desc()
    base_in = x.data_ptr() + f0 * fe * 4
    base_out = out.data_ptr() + f0 * fe * 4
    _hip.check(_hip.lib().vrg_grain_f32(C.c_void_p(base_in), C.c_void_p(base_out), nf, H, W, I32, S32, T32, C.byref(d),
                                       _hip.current_stream()), "vrg_grain_f32")


def _film_grain_oversize(x, out, grain_intensity, saturation_mix, chunk_frames, generator):
    """RNG chunks beyond 2^29 elements (batch_size 0 or >= 22 4K / >= 87 1080p frames): torch runs such a randn as several
    kernels over 32-bit indexable sub-ranges, each with its own grid and generator offset (rng.reserve_split).  The
    sub-ranges are not frame -- not even pixel -- aligned, so the chunk's noise is materialised leaf by leaf with the
    stream kernel (bit-identical to torch.randn) and applied with the injected-noise grain kernel: three passes over
    the chunk instead of one, for a case the reference handles at one tenth of the speed."""
    F = x.shape[0]
    fe = x[0].numel()
    step = min(chunk_frames if chunk_frames > 0 else F, F)
    lib = _hip.lib()
    I32, S32, T32 = _f32(grain_intensity), _f32(saturation_mix), _f32(1.0 - saturation_mix)
    noise = torch.empty((step * fe,), dtype=torch.float32, device=x.device)
    for f0 in range(0, F, step):
        nf = min(step, F - f0)
        numel = nf * fe
        if numel > rng.MAX_CHUNK_NUMEL:
            split = rng.reserve_split(numel, x.device, generator)
            seed, leaves = split.seed, split.leaves
        else:           # ragged tail chunk below the limit: one ordinary randn
            st = rng.reserve(numel, 1, x.device, generator)
            seed, leaves = st.seed, [(0, numel, st.grid_threads, st.offset0)]
        for start, size, G, off in leaves:
            d = _hip.NoiseDesc(seed0=seed, seed_stride=0, offset0=off, offset_stride=0, chunk0=0, chunk_frames=1, grid_threads=G)
            _hip.check(lib.vrg_noise_f32(C.c_void_p(noise.data_ptr() + 4 * start), 1, size, C.byref(d), _hip.current_stream()), "vrg_noise_f32")
        _hip.check(lib.vrg_grain_injected_f32(C.c_void_p(x.data_ptr() + 4 * f0 * fe), _hip.ptr(noise), C.c_void_p(out.data_ptr() + 4 * f0 * fe),
                                              nf * fe // 3, I32, S32, T32, _hip.current_stream()), "vrg_grain_injected_f32")
    return out


@_on_device
def film_grain(images: torch.Tensor, grain_intensity: float, saturation_mix: float, chunk_frames: int = 0,
               generator: Optional[torch.Generator] = None, plans=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Film grain with in-register Philox noise, bit-identical to the reference run on this GPU with the same
    generator state: chunks of `chunk_frames` frames each draw one ``torch.randn`` (0 = one draw for all)."""
    x = _check_frames(images, channels=3)
    F, H, W, _ = x.shape
    out = _out_like(x, out)
    if F == 0:
        return out
    fe = H * W * 3
    if plans is None and oversize_chunks(F, fe, chunk_frames):
        return _film_grain_oversize(x, out, grain_intensity, saturation_mix, chunk_frames, generator)
    main, tail, n_full = plans if plans is not None else plan_noise(F, fe, chunk_frames, x.device, generator)
    I32, S32, T32 = _f32(grain_intensity), _f32(saturation_mix), _f32(1.0 - saturation_mix)
    done = 0
    if main is not None:
        _grain_call(x, out, 0, n_full * main.chunk_frames, main, I32, S32, T32)
        done = n_full * main.chunk_frames
    if tail is not None:
        _grain_call(x, out, done, tail.chunk_frames, tail, I32, S32, T32)
    return out


@_on_device
def film_grain_seeded_frames(images: torch.Tensor, grain_intensity: float, saturation_mix: float, seed: int,
                             frame_start: int = 0) -> torch.Tensor:
    """Per-frame seeded grain: frame i uses generator seed (seed + frame_start + i) & 0x7FFFFFFF, offset 0
    (VRGDG_StandaloneVideoEnhancerNodes.py:262-278) -- independent of how frames are batched or sharded."""
    x = _check_frames(images, channels=3)
    F, H, W, _ = x.shape
    out = torch.empty_like(x)
    if F == 0 or grain_intensity <= 0:
        return x.clone() if F else out
    fe = H * W * 3
    I32, S32, T32 = _f32(grain_intensity), _f32(saturation_mix), _f32(1.0 - saturation_mix)
    first = int(seed) + int(frame_start)
    f = 0
    while f < F:   # split where the 31-bit mask wraps (practically never)
        s0 = (first + f) & 0x7FFFFFFF
        run = min(F - f, 0x80000000 - s0)
        plan = NoisePlan(1, rng.per_frame_seeded(fe, s0, x.device))
        _grain_call(x, out, f, run, plan, I32, S32, T32)
        f += run
    return out


@_on_device
def sharpen_then_seeded_grain(images: torch.Tensor, sharpen_strength: float, zero_border: bool, grain_intensity: float,
                              saturation_mix: float, seed: int, frame_start: int = 0) -> torch.Tensor:
    """Unsharp, then per-frame-seeded grain: the effect order of the stand-alone enhancer's _apply_effects_batch
    (VRGDG_StandaloneVideoEnhancerNodes.py:278-294) as ONE pass over the frames (vrg_sharpen_grain_f32: 24 B/px).  Frame sizes the
    fused kernel does not take (width not a multiple of 4 or below 344) run the two kernels; the result is the same bits either way
    (tests/test_gpu_parity.py::test_fused_sharpen_then_seeded_grain_equals_the_two_kernels)."""
    if isinstance(images, torch.Tensor) and images.dtype == torch.uint8:
        return _sharpen_then_seeded_grain_u8(images, sharpen_strength, zero_border, grain_intensity, saturation_mix, seed, frame_start)
    x = _check_frames(images, channels=3)
    F, H, W, _ = x.shape
    if F == 0 or sharpen_strength <= 0 or grain_intensity <= 0:
        y = stencil3x3(x, "unsharp", sharpen_strength, zero_border) if (F and sharpen_strength > 0) else x
        return film_grain_seeded_frames(y, grain_intensity, saturation_mix, seed, frame_start) if (F and grain_intensity > 0) else y
    fe = H * W * 3
    I32, S32, T32 = _f32(grain_intensity), _f32(saturation_mix), _f32(1.0 - saturation_mix)
    out = torch.empty_like(x)
    lib = _hip.lib()
    first = int(seed) + int(frame_start)
    f = 0
    while f < F:   # split where the 31-bit mask wraps (practically never)
        s0 = (first + f) & 0x7FFFFFFF
        run = min(F - f, 0x80000000 - s0)
        d = NoisePlan(1, rng.per_frame_seeded(fe, s0, x.device)).desc()
        st = lib.vrg_sharpen_grain_f32(C.c_void_p(x.data_ptr() + 4 * f * fe), C.c_void_p(out.data_ptr() + 4 * f * fe), run, H, W,
                                       _f32(sharpen_strength), _hip.BORDER_ZERO if zero_border else _hip.BORDER_REPLICATE, I32, S32, T32,
                                       C.byref(d), _hip.current_stream())
        if st == _hip.VRG_ERR_UNSUPPORTED:
            if f != 0:
                raise RuntimeError("vrg_sharpen_grain_f32 refused a later run of a batch it had accepted")
            return film_grain_seeded_frames(stencil3x3(x, "unsharp", sharpen_strength, zero_border), grain_intensity, saturation_mix,
                                            seed, frame_start)
        _hip.check(st, "vrg_sharpen_grain_f32")
        f += run
    return out


def _sharpen_then_seeded_grain_u8(frames_bgr, sharpen_strength, zero_border, grain_intensity, saturation_mix, seed, frame_start):
    """Decoded uint8 B,G,R frames in and out: ``_tensor_to_frames(_apply_effects_batch(_frames_to_tensor(frames)))`` of the enhancer's
    render loop (VRGDG_StandaloneVideoEnhancerNodes.py:417-421) as one kernel (vrg_sharpen_grain_u8, 3 + 3 B/px; any width, height and
    alignment since round 5).  A batch of fewer than four bytes, or a call with one of the two effects off, takes the converter -> fp32 ->
    converter route: the same bytes."""
    x = _check_frames(frames_bgr, "frames", channels=3, dtype=torch.uint8)
    F, H, W, _ = x.shape
    if F == 0:
        return x.clone()

    def through_fp32():
        y = frames_u8_to_f32(x)
        if sharpen_strength > 0:
            y = stencil3x3(y, "unsharp", sharpen_strength, zero_border)
        if grain_intensity > 0:
            y = film_grain_seeded_frames(y, grain_intensity, saturation_mix, seed, frame_start)
        return f32_to_frames_u8(y)

    if sharpen_strength <= 0 or grain_intensity <= 0:
        return through_fp32()
    fe = H * W * 3
    I32, S32, T32 = _f32(grain_intensity), _f32(saturation_mix), _f32(1.0 - saturation_mix)
    out = torch.empty_like(x)
    lib = _hip.lib()
    first = int(seed) + int(frame_start)
    f = 0
    while f < F:   # split where the 31-bit mask wraps (practically never)
        s0 = (first + f) & 0x7FFFFFFF
        run = min(F - f, 0x80000000 - s0)
        d = NoisePlan(1, rng.per_frame_seeded(fe, s0, x.device)).desc()
        st = lib.vrg_sharpen_grain_u8(C.c_void_p(x.data_ptr() + f * fe), C.c_void_p(out.data_ptr() + f * fe), run, H, W,
                                      _f32(sharpen_strength), _hip.BORDER_ZERO if zero_border else _hip.BORDER_REPLICATE, I32, S32, T32,
                                      C.byref(d), _hip.current_stream())
        if st == _hip.VRG_ERR_UNSUPPORTED:
            if f != 0:
                raise RuntimeError("vrg_sharpen_grain_u8 refused a later run of a batch it had accepted")
            return through_fp32()
        _hip.check(st, "vrg_sharpen_grain_u8")
        f += run
    return out


@_on_device
def film_grain_injected(images: torch.Tensor, noise: torch.Tensor, grain_intensity: float, saturation_mix: float) -> torch.Tensor:
    """Grain arithmetic with caller-supplied N(0,1) noise (same shape): the noise-injection parity form."""
    x = _check_frames(images, channels=3)
    n = _check_frames(noise, "noise", channels=3)
    if n.shape != x.shape:
        raise ValueError("noise must have the shape of images")
    out = torch.empty_like(x)
    px = x.numel() // 3
    _hip.check(_hip.lib().vrg_grain_injected_f32(_hip.ptr(x), _hip.ptr(n), _hip.ptr(out), px, _f32(grain_intensity),
                                                _f32(saturation_mix), _f32(1.0 - saturation_mix), _hip.current_stream()),
               "vrg_grain_injected_f32")
    return out


# ------------------------------------------------------------------------------------------------
# 3D LUT
# ------------------------------------------------------------------------------------------------

@dataclass
class DeviceLut:
    table: torch.Tensor        # record table on the device: (N-1)^2 * N records of 12 fp32 (vrg_lut_prepare_f32)
    size: int                  # N
    domain_min: tuple
    domain_max: tuple
    nodes: Optional[torch.Tensor] = None   # the parsed [N,N,N,3] table itself on the device (ffmpeg-style lookup only)


def upload_lut(lut_data: dict, device) -> DeviceLut:
    """Upload a parsed .cube table ([N,N,N,3], index [b][g][r]) and rewrite it on the device into the
    cell-major form the kernels read (values copied verbatim)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is not None and device.index != torch.cuda.current_device():
        with torch.cuda.device(device):
            return upload_lut(lut_data, device)
    raw = lut_data["lut"].to(device=device, dtype=torch.float32).contiguous()
    if raw.ndim != 4 or raw.shape[-1] != 3 or not (raw.shape[0] == raw.shape[1] == raw.shape[2]):
        raise ValueError("LUT table must be [N, N, N, 3]")
    n = int(raw.shape[0])
    count = int(_hip.lib().vrg_lut_cells_floats(n))
    if count <= 0:
        raise ValueError(f"unsupported LUT size {n} (2..256)")
    cells = torch.empty((count,), dtype=torch.float32, device=device)
    _hip.check(_hip.lib().vrg_lut_prepare_f32(_hip.ptr(raw), n, _hip.ptr(cells), _hip.current_stream()), "vrg_lut_prepare_f32")
    dmin = tuple(float(v) for v in lut_data["domain_min"].to(torch.float32).cpu().tolist())
    dmax = tuple(float(v) for v in lut_data["domain_max"].to(torch.float32).cpu().tolist())
    return DeviceLut(cells, n, dmin, dmax, raw)


def blend_terms(strength: float):
    """(blend_mode, B, 1-B) from the node's 0..10 strength (VRGDG_IV_Adjustments.py:355-359).
    mode 0 = return the input, 1 = LUT only, 2 = x*(1-B) + y*B."""
    blend = max(0.0, min(10.0, float(strength))) / 10.0
    if blend <= 0.0:
        return 0, 0.0, 1.0
    if blend < 1.0:
        return 2, _f32(blend), _f32(1.0 - blend)
    return 1, 1.0, 0.0


@_on_device
def lut3d(image: torch.Tensor, lut: DeviceLut, strength: float = 10.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if image.ndim != 4 or image.shape[-1] < 3:
        raise ValueError("VRGDG_LUTS expects IMAGE input shaped like [batch, height, width, channels].")
    x = _check_frames(image, "image")
    mode, B, omB = blend_terms(strength)
    if mode == 0:
        if out is not None:
            _out_like(x, out).copy_(x)
            return out
        return x
    _check_side(lut.table, "LUT record table", x)
    out = _out_like(x, out)
    px = x.numel() // x.shape[-1]
    if px == 0:
        return out
    _hip.check(_hip.lib().vrg_lut3d_f32(_hip.ptr(x), _hip.ptr(out), px, x.shape[-1], _hip.ptr(lut.table), lut.size,
                                       _F3(*lut.domain_min), _F3(*lut.domain_max), mode, B, omB, _hip.current_stream()),
               "vrg_lut3d_f32")
    return out


@_on_device
def lut3d_ffmpeg_u8(frames_bgr: torch.Tensor, lut: DeviceLut, weights=None) -> torch.Tensor:
    """ffmpeg's ``lut3d`` (tetrahedral) and, with `weights` (one blend weight per frame), its ``blend`` expression
    ``A*(1-w)+B*w`` on decoded uint8 B,G,R frames: the per-pixel step of the reference's opening colour match as ffmpeg
    performs it.  Restated from ffmpeg's published sources, parity unpinned (csrc/vrg_lut_tetra.hip)."""
    x = _check_frames(frames_bgr, "frames", channels=3, dtype=torch.uint8)
    if lut.nodes is None:
        raise ValueError("this DeviceLut carries no node table (use ops.upload_lut)")
    _check_side(lut.nodes, "LUT node table", x)
    F, H, W, _ = x.shape
    out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    wt = None
    if weights is not None:
        wt = torch.tensor([float(w) for w in weights], dtype=torch.float64, device=x.device)
        if wt.numel() != F:
            raise ValueError(f"{wt.numel()} blend weights for {F} frames")
    _hip.check(_hip.lib().vrg_lut3d_tetra_u8(_hip.ptr(x), _hip.ptr(out), F, H * W, _hip.ptr(lut.nodes), lut.size,
                                            _F3(*lut.domain_min), _F3(*lut.domain_max), _hip.ptr(wt) if wt is not None else None,
                                            _hip.current_stream()), "vrg_lut3d_tetra_u8")
    return out


# ------------------------------------------------------------------------------------------------
# 3x3 stencils
# ------------------------------------------------------------------------------------------------

_STENCIL = {"unsharp": _hip.STENCIL_UNSHARP, "laplacian": _hip.STENCIL_LAPLACIAN, "sobel": _hip.STENCIL_SOBEL}


@_on_device
def stencil3x3(images: torch.Tensor, op: str, strength: float, zero_border: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """unsharp / laplacian / sobel.  zero_border=False is the reference's default numpy path (edge replicate);
    True is its ``use_gpu`` path (zero padding, and for laplacian/sobel the conv2d sign/epsilon conventions)."""
    x = _check_frames(images)
    out = _out_like(x, out)
    F, H, W, Cn = x.shape
    if x.numel() == 0:
        return out
    _hip.check(_hip.lib().vrg_stencil3x3_f32(_hip.ptr(x), _hip.ptr(out), F, H, W, Cn, _STENCIL[op],
                                            _hip.BORDER_ZERO if zero_border else _hip.BORDER_REPLICATE, _f32(strength),
                                            _hip.current_stream()), "vrg_stencil3x3_f32")
    return out


# ------------------------------------------------------------------------------------------------
# 13-slider Adjust (video routes)
# ------------------------------------------------------------------------------------------------

def adjust_terms(adjust: dict, device_math: bool = False) -> "_hip.AdjustDesc":
    """Slider arithmetic of _apply_adjust_tensor (VRGDG_LUTVideoTools.py:309-315 ff.) in Python doubles, rounded
    once to fp32 -- what torch does with a Python scalar operand.  `adjust` is a normalized settings dict.
    `device_math`: reproduce the reference run with device="cuda" (its `/ 0.45`, `/ 1.05` become multiplications by the
    fp32-rounded reciprocal, as torch evaluates tensor / scalar on a GPU) instead of with device="cpu" (IEEE quotients)."""
    d = _hip.AdjustDesc()
    d.div_mode = _hip.ADJUST_DIV_DEVICE if device_math else _hip.ADJUST_DIV_IEEE
    d.enabled = 1 if adjust["enabled"] else 0
    t, ti = adjust["temperature"], adjust["tint"]
    d.shift[0] = _f32(t / 400.0 - ti / 900.0)
    d.shift[1] = _f32(ti / 450.0)
    d.shift[2] = _f32(-t / 400.0 - ti / 900.0)
    d.exposure = _f32(2.0 ** (adjust["exposure"] / 100.0))
    d.contrast = _f32(1.0 + (adjust["contrast"] / 100.0))
    d.saturation = _f32(1.0 + (adjust["saturation"] / 100.0))
    d.highlights = _f32(adjust["highlights"] / 220.0)
    d.shadows = _f32(adjust["shadows"] / 220.0)
    d.whites = _f32(adjust["whites"] / 240.0)
    d.blacks = _f32(adjust["blacks"] / 240.0)
    clarity = adjust["clarity"] / 100.0
    sharpen = adjust["sharpen"] / 100.0
    d.has_clarity = 1 if abs(clarity) > 0.001 else 0
    d.clarity = _f32(clarity)
    d.has_sharpen = 1 if sharpen > 0.001 else 0
    d.sharpen = _f32(sharpen)
    fade = adjust["fade"] / 100.0
    d.has_fade = 1 if fade > 0.0 else 0
    d.fade_mul = _f32(1.0 - fade * 0.35)
    d.fade_add = _f32(fade * 0.18)
    vignette = adjust["vignette"] / 100.0
    d.has_vignette = 1 if vignette > 0.0 else 0
    d.vignette = _f32(vignette)
    return d


@_on_device
def adjust(images: torch.Tensor, terms: "_hip.AdjustDesc", out: torch.Tensor | None = None,
           workspace: torch.Tensor | None = None) -> torch.Tensor:
    """Run the Adjust kernels on ``[F,H,W,3]`` frames: fp32 R,G,B tensors, or uint8 B,G,R decoded frames (then the
    result is uint8 B,G,R as well, equal to convert -> adjust -> convert).  `workspace` (fp32, same shape) is used
    only when clarity and sharpen are both active; it is allocated when not supplied."""
    u8 = isinstance(images, torch.Tensor) and images.dtype == torch.uint8       # decoded BGR frames (video routes)
    x = _check_frames(images, dtype=torch.uint8 if u8 else torch.float32)
    if x.shape[-1] != 3:
        raise ValueError("adjust expects 3-channel frames")
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous() or out.device != x.device:
        raise ValueError("out must be a contiguous tensor shaped and typed like the frames on the same device")
    F, H, W, _ = x.shape
    if x.numel() == 0:
        return out
    tmp = None
    if terms.enabled and terms.has_clarity and terms.has_sharpen:
        tmp = workspace if workspace is not None else torch.empty(x.shape, dtype=torch.float32, device=x.device)
        if tmp.shape != x.shape or tmp.dtype != torch.float32 or not tmp.is_contiguous() or tmp.device != x.device:
            raise ValueError("adjust workspace must be a contiguous float32 tensor shaped like the frames")
    fn, name = (_hip.lib().vrg_adjust_u8, "vrg_adjust_u8") if u8 else (_hip.lib().vrg_adjust_f32, "vrg_adjust_f32")
    _hip.check(fn(_hip.ptr(x), _hip.ptr(out), _hip.ptr(tmp) if tmp is not None else None, F, H, W, C.byref(terms),
                  _hip.current_stream()), name)
    return out


# ------------------------------------------------------------------------------------------------
# uint8 BGR frames <-> fp32 RGB tensors (video I/O edge)
# ------------------------------------------------------------------------------------------------

@_on_device
def frames_u8_to_f32(frames_bgr: torch.Tensor) -> torch.Tensor:
    """``[F,H,W,3]`` uint8 B,G,R on the GPU -> fp32 R,G,B in [0,1] (``/ 255.0``)."""
    x = _check_frames(frames_bgr, "frames", channels=3, dtype=torch.uint8)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if x.numel():
        _hip.check(_hip.lib().vrg_u8bgr_to_f32rgb(_hip.ptr(x), _hip.ptr(out), x.numel() // 3, _hip.current_stream()),
                   "vrg_u8bgr_to_f32rgb")
    return out


@_on_device
def f32_to_frames_u8(images: torch.Tensor) -> torch.Tensor:
    """fp32 R,G,B -> ``clip(x * 255, 0, 255)`` truncated to uint8, B,G,R order."""
    x = _check_frames(images, channels=3)
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    if x.numel():
        _hip.check(_hip.lib().vrg_f32rgb_to_u8bgr(_hip.ptr(x), _hip.ptr(out), x.numel() // 3, _hip.current_stream()),
                   "vrg_f32rgb_to_u8bgr")
    return out


# ------------------------------------------------------------------------------------------------
# colour match
# ------------------------------------------------------------------------------------------------

def _stats_scratch(frames: int, device) -> torch.Tensor:
    nbytes = int(_hip.lib().vrg_lab_stats_scratch_bytes(frames))
    return torch.empty((max(nbytes, 8) // 8,), dtype=torch.float64, device=device)


@_on_device
def lab_stats(images: torch.Tensor, cm_math=None) -> torch.Tensor:
    """Per-frame Lab statistics as fp64 ``[F, 3, 3]`` = (n, mean, M2) per channel L,a,b."""
    x = _check_frames(images, channels=3)
    F, H, W, _ = x.shape
    stats = torch.empty((F, 3, 3), dtype=torch.float64, device=x.device)
    if F == 0:
        return stats
    scratch = _stats_scratch(F, x.device)
    _hip.check(_hip.lib().vrg_lab_stats_f32(_hip.ptr(x), F, H, W, _hip.ptr(stats), _hip.ptr(scratch), _cm_math(cm_math),
                                           _hip.current_stream()), "vrg_lab_stats_f32")
    return stats


@_on_device
def finalize_stats(stats: torch.Tensor) -> torch.Tensor:
    """(n, mean, M2) fp64 -> fp32 ``[F, 3, 2]`` = (mean, unbiased std + 1e-5) (nodes.py:99-100, 109-110)."""
    F = stats.shape[0]
    ms = torch.empty((F, 3, 2), dtype=torch.float32, device=stats.device)
    if F:
        _hip.check(_hip.lib().vrg_lab_stats_finalize(_hip.ptr(stats), _hip.ptr(ms), F, _hip.current_stream()),
                   "vrg_lab_stats_finalize")
    return ms


def _chunk_runs(frames: int, chunks) -> list:
    """`chunks`: frames per reference call (an int: the node's batch_size, last call = remainder) or the explicit list of call
    sizes.  Returns [(first_frame, frames, call_size)] runs of equally sized calls."""
    if isinstance(chunks, int):
        if chunks < 1:
            raise ValueError("colour-match chunk must be >= 1")
        return [(0, frames, chunks)] if frames else []
    sizes = [int(c) for c in chunks]
    if any(c < 1 for c in sizes) or sum(sizes) != frames:
        raise ValueError("colour-match chunk sizes must be positive and add up to the number of frames")
    runs, f = [], 0
    for c in sizes:
        if runs and runs[-1][2] == c and runs[-1][1] % c == 0:
            runs[-1] = (runs[-1][0], runs[-1][1] + c, c)
        else:
            runs.append((f, c, c))
        f += c
    return runs


_TS_CHECKED = {}             # device index -> True (replay == this torch build) / False


def _selfcheck_planes(device):
    """Probe tensors of the self-check: small planes for the generic (one workgroup per plane) form -- an aligned one and one whose
    H*W is odd, so that planes start off a vector boundary -- and, for the whole-frame forms the video-sized calls take, one plane
    per batch_size class (block (256,2) for 1, (128,4) for 2, (64,8) for >= 3) shaped like a 4K row band: long enough (>= 2 rounds of
    2048 pixels plus a ragged tail) to run the same thread loop, lane tree and warp tree as a full frame."""
    g = torch.Generator().manual_seed(20260926)
    small = (torch.rand((3, 24, 40, 3), generator=g) * 100.0 - 30.0).to(device)
    odd = (torch.rand((4, 23, 41, 3), generator=g) * 100.0 - 30.0).to(device)
    band = (torch.rand((6, 4, 3840, 3), generator=g) * 120.0 - 40.0).to(device)          # H*W = 15,360: 7.5 rounds
    many = (torch.rand((66, 4, 3840, 3), generator=g) * 120.0 - 40.0).to(device)        # > 64 frames: the one-workgroup-per-frame form
    return [(small, (1, 3)), (odd, (1, 2, 4)), (band, (1, 2, 3, 6)), (many, (1,))]


def device_stats_selfcheck(device, force: bool = False) -> bool:
    """Once per device and process (a few ms): the replayed reductions against THIS torch build's `mean` / `std` over the call
    geometries that matter -- every batch_size class of the whole-frame kernels, an unaligned plane, ragged last calls.  The replay
    follows torch 2.10.0+rocm7.0's reduce_kernel; another build may pick another geometry or contraction -- then the node is no longer
    bit-equal to the reference run under that build (it stays within the statistics band, ~20 ulp(1.0)).  Returns whether the replay
    matched; a mismatch warns once and is reported by `device_stats_status()` so that a host can surface it next to its results."""
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    with _STATE_LOCK:
        if key in _TS_CHECKED and not force:
            return _TS_CHECKED[key]
    ok = True
    with torch.cuda.device(key):
        for probe, call_sizes in _selfcheck_planes(torch.device("cuda", key)):
            F, H, W, _ = probe.shape
            for calls in call_sizes:
                want = []
                for i in range(0, F, calls):
                    t = probe[i:i + calls].permute(0, 3, 1, 2).contiguous()
                    want.append(torch.stack([t.mean(dim=[2, 3]), t.std(dim=[2, 3])], dim=-1))
                want = torch.cat(want, dim=0)
                got = torch.empty((F, 3, 2), dtype=torch.float32, device=probe.device)
                _hip.check(_hip.lib().vrg_lab_stats_torch_f32(_hip.ptr(probe), F, H, W, calls, _hip.ptr(got), _f32(0.0), _hip.current_stream()),
                           "vrg_lab_stats_torch_f32")
                ok = ok and torch.equal(got, want)
                # the forms production takes (lab_stats_device): with a scratch buffer, the throughput entry and the latency entry
                # (one accumulator per lane, <= 2 frames per call)
                nbytes = int(_hip.lib().vrg_lab_stats_torch_scratch_bytes(F))
                if nbytes:
                    scratch = torch.empty((nbytes + 15) // 16 * 4, dtype=torch.float32, device=probe.device)
                    for name in ("vrg_lab_stats_torch_ws_f32", "vrg_lab_stats_torch_lat_f32"):
                        got2 = torch.empty((F, 3, 2), dtype=torch.float32, device=probe.device)
                        _hip.check(getattr(_hip.lib(), name)(_hip.ptr(probe), F, H, W, calls, _hip.ptr(got2), _f32(0.0), _hip.ptr(scratch), nbytes,
                                                             _hip.current_stream()), name)
                        ok = ok and torch.equal(got2, want)
    with _STATE_LOCK:
        first = key not in _TS_CHECKED
        _TS_CHECKED[key] = ok
    if not ok and first:
        import warnings
        warnings.warn(f"comfyui-vrgamedevgirl_amd: torch {torch.__version__} reduces mean()/std() in another order than the one this "
                      "library replays (torch 2.10.0+rocm7.0): ColorMatchToReference stays within a few ulp of the reference instead of "
                      "bit-equal to it (ops.device_stats_status())", RuntimeWarning)
    return ok


def device_stats_status(device=None) -> dict:
    """What a host can log next to a colour-match result: which statistics the default policy uses on `device` and whether the
    replay of torch's reductions was verified against this torch build in this process."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if not device_stats_supported(device):
        return {"cm_stats": "fp64", "bit_equal_to_torch": False, "reason": "fewer than 100 CUs: torch's reduction geometry is not replayed"}
    ok = device_stats_selfcheck(device)
    return {"cm_stats": "device", "bit_equal_to_torch": bool(ok), "torch": torch.__version__,
            "reason": None if ok else "this torch build reduces in another order than the replayed one (torch 2.10.0+rocm7.0)"}


def _device_stats_selfcheck(device) -> None:
    device_stats_selfcheck(device)
    if os.environ.get("VRGDG_SELFCHECK", "1") != "0":
        toolchain_selfcheck(device)


# ------------------------------------------------------------------------------------------------
# Toolchain coupling, checked at first use: two fast forms rest on measurements of ONE ROCm build --
#   * dev_pow_ziv's rounding test carries half-widths calibrated against this build's ocml logarithm (tools/ziv_calibration.json);
#     another ocml could make it accept a wrongly rounded power;
#   * the steady rows of the wave march issue their LUT gathers as inline-assembly LDS-DMA with hand-counted `vmcnt` waits
#     (csrc/vrg_march.hip); a compiler that schedules other memory operations between them breaks the count.
# The first colour match / fused chain on a device compares (a) dev_pow_ziv at its three call sites with torch.pow -- ocml's powf --
# on that device over 3 x 2^20 arguments of the call sites' domains, the neighbourhood of 1.0 included, and (b) the march with the
# LDS-tile kernels (vrg_chain_desc.variant 2 against 1) on a geometry whose rows are steady.  (b) failing switches the automatic
# choice to the tile kernels for the process (same results, slower) and warns; (a) failing warns and is reported by
# `toolchain_status()` -- the powers then differ from the reference's by an ulp in rare lanes.
# ------------------------------------------------------------------------------------------------
_TOOLCHAIN = {}              # device index -> {"pow": bool, "march": bool}
_POW_SITES = ((12, 2.4, 0.0625, 2.0), (13, 1.0 / 2.4, 0.0031308, 4.0), (14, 1.0 / 3.0, 0.008856, 4.0))     # (vrg_debug_cm_math op, y, domain) per call site


def _pow_probe_arguments(lo: float, hi: float, device) -> torch.Tensor:
    a, b = np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32)
    n = 1 << 20
    bits = (int(a) + (np.arange(n, dtype=np.uint64) * (int(b) - int(a)) // (n - 1))).astype(np.uint32)
    near1 = (np.float32(1.0).view(np.uint32) + np.arange(-4096, 4096, dtype=np.int64)).astype(np.uint32)      # saturated pixels: x ~ 1
    return torch.from_numpy(np.concatenate([bits, near1]).view(np.float32)).to(device)


_TOOLCHAIN_RUNNING = set()   # device indices whose probes are running right now (on whatever thread)


def toolchain_selfcheck(device, force: bool = False) -> dict:
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    with _STATE_LOCK:
        if key in _TOOLCHAIN and not force:
            return _TOOLCHAIN[key]
        if key in _TOOLCHAIN_RUNNING:
            # the probes below run fused_chain themselves (re-entrancy), and another host thread may arrive while they run: both get the
            # SAFE answer for this one call -- the tile kernels, same bits -- and nothing is recorded before the probes have finished
            return {"pow": True, "march": False}
        if torch.cuda.is_current_stream_capturing():
            return {"pow": True, "march": False}          # the probes synchronise: not inside a graph capture; checked at the next plain call
        _TOOLCHAIN_RUNNING.add(key)
    dev = torch.device("cuda", key)
    res = {"pow": True, "march": True}
    try:
        with torch.cuda.device(key):
            lib = _hip.lib()
            for site, (op, y, lo, hi) in enumerate(_POW_SITES):
                x = _pow_probe_arguments(lo, hi, dev)
                got = torch.empty_like(x)
                _hip.check(lib.vrg_selfcheck_pow_f32(_hip.ptr(x), _hip.ptr(got), x.numel(), site, _hip.current_stream()), "vrg_selfcheck_pow_f32")
                res["pow"] = res["pow"] and bool(torch.equal(got, torch.pow(x, y)))
            g = torch.Generator(device=dev).manual_seed(20260926)
            frames = torch.rand((2, 512, 512, 3), generator=g, device=dev)       # one RNG chunk of 2 frames: full-grid Philox geometry, steady rows
            axis = np.linspace(0.0, 1.0, 33, dtype=np.float32)
            bb, gg, rr = np.meshgrid(axis, axis, axis, indexing="ij")             # [b][g][r] -> (r, g, b): a smooth non-identity grade
            table = np.stack([rr ** 1.25, 0.9 * gg + 0.1 * bb, bb * bb], axis=-1).astype(np.float32)
            lut = upload_lut({"lut": torch.from_numpy(table), "domain_min": torch.zeros(3), "domain_max": torch.ones(3)}, dev)
            outs = []
            for variant in (2, 1):
                gen = torch.Generator(device=dev).manual_seed(7)
                spec = ChainSpec(grain=(0.04, 0.5, 2), lut=(lut, 10.0), sharpen=("unsharp", 0.5, False), variant=variant)
                outs.append(fused_chain(frames, spec, generator=gen))
            res["march"] = bool(torch.equal(outs[0], outs[1]))
    except Exception as exc:             # a probe that cannot run (out of memory, a launch error) proves nothing: take the safe kernels, say so
        res = {"pow": False, "march": False, "error": f"{type(exc).__name__}: {exc}"}
    finally:
        with _STATE_LOCK:
            _TOOLCHAIN_RUNNING.discard(key)
            _TOOLCHAIN[key] = res
    if not (res["pow"] and res["march"]):
        import warnings
        what = []
        if "error" in res:
            what.append(f"the first-use self-check could not run ({res['error']}): fused chains take the tile kernels in this process")
        else:
            if not res["march"]:
                what.append("the wave-march kernel disagrees with the tile kernels (hand-counted waits of its LDS-DMA gathers): fused chains take the "
                            "tile kernels in this process")
            if not res["pow"]:
                what.append("dev_pow_ziv disagrees with this ROCm's powf (its rounding test is calibrated against ROCm 7.0's ocml): colour match may "
                            "differ from the reference by an ulp in rare pixels")
        warnings.warn("comfyui-vrgamedevgirl_amd: built or run with another toolchain than the one its fast forms were measured on -- "
                      + "; ".join(what) + " (ops.toolchain_status())", RuntimeWarning)
    return res


def toolchain_status(device=None) -> dict:
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    res = toolchain_selfcheck(device)
    return {"pow_equals_ocml": res["pow"], "march_equals_tile_kernels": res["march"], "error": res.get("error"), "hip": torch.version.hip,
            "torch": torch.__version__}


def _auto_variant(device) -> int:
    """vrg_chain_desc.variant for the automatic choice: 0, or 1 (tile kernels) where the march failed its self-check on this device."""
    if os.environ.get("VRGDG_SELFCHECK", "1") == "0":
        return 0
    return 0 if toolchain_selfcheck(device)["march"] else 1


@_on_device
def lab_stats_device(lab: torch.Tensor, chunks=1, eps: float = 1e-5, out: Optional[torch.Tensor] = None, latency_form: bool = False) -> torch.Tensor:
    """fp32 ``[F, 3, 2]`` = (mean, unbiased std + eps) of an interleaved Lab image ``[F,H,W,3]`` with the BITS
    ``lab.mean(dim=[2,3])`` / ``lab.std(dim=[2,3]) + 1e-5`` have when torch evaluates them on this GPU for ``chunks`` frames per
    call (nodes.py:99-100, 109-110).  include/vrgdg_hip.h: vrg_lab_stats_torch_f32.  `latency_form`: allow the one-accumulator-per-lane
    kernels for calls of at most two frames (vrg_lab_stats_torch_lat_f32) -- faster on an otherwise idle GPU, slower beside a full-size
    pass; same bits."""
    x = _check_frames(lab, "lab", channels=3)
    F, H, W, _ = x.shape
    ms = out if out is not None else torch.empty((F, 3, 2), dtype=torch.float32, device=x.device)
    if out is not None:
        _check_side(out, "statistics output", x, (3, 2))
        if int(out.shape[0]) != F:
            raise ValueError("statistics output must hold one [3, 2] record per frame")
    fe = H * W * 3
    if F:
        _device_stats_selfcheck(x.device)
    lib = _hip.lib()
    for f0, nf, c in _chunk_runs(F, chunks):
        nbytes = int(lib.vrg_lab_stats_torch_scratch_bytes(nf))      # small batches: the half-block form wants a scratch buffer (0: another form)
        scratch = torch.empty((nbytes + 15) // 16 * 4, dtype=torch.float32, device=x.device) if nbytes else None
        entry = lib.vrg_lab_stats_torch_lat_f32 if latency_form else lib.vrg_lab_stats_torch_ws_f32
        _hip.check(entry(C.c_void_p(x.data_ptr() + f0 * fe * 4), nf, H, W, c, C.c_void_p(ms.data_ptr() + f0 * 24), _f32(eps),
                         _hip.ptr(scratch) if scratch is not None else None, nbytes, _hip.current_stream()),
                   "vrg_lab_stats_torch_lat_f32" if latency_form else "vrg_lab_stats_torch_ws_f32")
    return ms


#: steps of at most this many frames take the latency form of the reference frame's statistics (measured: 4 / 8 / 16 frames per step
#: 2.0 / 2.8 / 4.5 ms against 2.4 / 3.4 / 5.0; 32 / 64 frames 8.4 / 16.8 against 8.0 / 14.4 -- profiles/r04_frames_table_*.json)
SMALL_STEP_FRAMES = 16


def reference_stats(reference_image: torch.Tensor, cm_math=None, cm_stats=None, step_frames: Optional[int] = None) -> torch.Tensor:
    """fp32 ``[R, 3, 2]`` (mean, std + 1e-5) of the reference frame(s) (nodes.py:98-100): one reduction call over the whole
    reference batch with the device statistics.  `step_frames`: how many frames the step these statistics belong to processes, if the
    caller knows -- small steps leave the GPU idle and take the latency form of the reduction (SMALL_STEP_FRAMES)."""
    ref = _check_frames(reference_image, "reference_image", channels=3)
    if _cm_stats(cm_stats, cm_math, ref.device) == "fp64":
        return finalize_stats(lab_stats(ref, cm_math))
    lab = torch.empty_like(ref)
    Rn, H, W, _ = ref.shape
    if Rn:
        d = _chain_desc(ChainSpec(cm_math=cm_math), None, [], ref)
        _hip.check(_hip.lib().vrg_chain_stats_lab_f32(_hip.ptr(ref), _hip.ptr(lab), Rn, H, W, C.byref(d), None, None, _hip.current_stream()),
                   "vrg_chain_stats_lab_f32")        # the Lab image only
    return lab_stats_device(lab, max(int(Rn), 1), latency_form=step_frames is not None and int(step_frames) <= SMALL_STEP_FRAMES)


_SIDE_STREAMS = {}


def _side_stream(device) -> "torch.cuda.Stream":
    """One extra stream per device for work that is off the critical path of the frame passes: the statistics of the reference frame.
    HIGH priority: its kernels are a handful of latency-bound workgroups; on a normal-priority stream they queue behind the thousands of
    workgroups of pass 1 (measured: 32 x 4K frames per step 11.7 ms = the passes' 9.4 ms + the reference statistics' 2.2 ms, nothing
    hidden), a high-priority queue lets the dispatcher slot them in as soon as they are submitted."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    with _STATE_LOCK:
        st = _SIDE_STREAMS.get(key)
        if st is None:
            st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=key, priority=-1)
    return st


@_on_device
def reference_stats_async(reference_image: torch.Tensor, cm_math=None, cm_stats=None, step_frames: Optional[int] = None):
    """reference_stats on the side stream: returns (ref_ms, event).  The reference frame's statistics are one latency-bound chain
    (16,200 dependent Welford updates per accumulator for a 4K frame: 1.5-2 ms on a handful of workgroups) that only the APPLY pass
    needs -- give `event` to ChainSpec.cm_ref_event and pass 1 of the batch runs meanwhile."""
    main = torch.cuda.current_stream()
    side = _side_stream(reference_image.device)
    side.wait_stream(main)                         # the reference frame may have been produced on the caller's stream
    with torch.cuda.stream(side):
        ref_ms = reference_stats(reference_image, cm_math, cm_stats, step_frames=step_frames)
        ev = torch.cuda.Event()
        ev.record(side)
    ref_ms.record_stream(main)
    reference_image.record_stream(side)
    return ref_ms, ev


def merge_stats(parts: torch.Tensor) -> torch.Tensor:
    """Chan/Golub/LeVeque merge of (n, mean, M2) triples along dim 0, in index order (deterministic):
    used to combine the slices of one reference frame reduced on different GPUs.  ``parts``: [R, ..., 3]."""
    acc = parts[0].clone()
    for r in range(1, parts.shape[0]):
        nb, mb, m2b = parts[r][..., 0], parts[r][..., 1], parts[r][..., 2]
        na, ma, m2a = acc[..., 0], acc[..., 1], acc[..., 2]
        n = na + nb
        delta = mb - ma
        mean = ma + delta * (nb / n)
        m2 = m2a + m2b + delta * delta * (na * nb / n)
        acc = torch.stack([n, mean, m2], dim=-1)
    return acc


@_on_device
def colormatch_apply(images: torch.Tensor, img_ms: torch.Tensor, ref_ms: torch.Tensor, match_strength: float,
                     cm_math=None) -> torch.Tensor:
    x = _check_frames(images, channels=3)
    F, H, W, _ = x.shape
    out = torch.empty_like(x)
    if F == 0:
        return out
    _check_side(ref_ms, "ref_ms", x, (3, 2))
    _check_side(img_ms, "img_ms", x, (3, 2))
    if int(img_ms.shape[0]) != F:
        raise ValueError(f"img_ms holds {int(img_ms.shape[0])} frames of statistics for {F} frames")
    R = int(ref_ms.shape[0])
    if R != 1 and F % R != 0:
        raise RuntimeError(f"The size of tensor a ({F}) must match the size of tensor b ({R}) at non-singleton dimension 0")
    _hip.check(_hip.lib().vrg_colormatch_apply_f32(_hip.ptr(x), _hip.ptr(out), F, H, W, _hip.ptr(img_ms), _hip.ptr(ref_ms), R,
                                                  _f32(match_strength), _f32(1.0 - match_strength), _cm_math(cm_math),
                                                  _hip.current_stream()),
               "vrg_colormatch_apply_f32")
    return out


@_on_device
def color_match(images: torch.Tensor, reference_image: torch.Tensor, match_strength: float,
                ref_ms: Optional[torch.Tensor] = None, cache_lab: bool = True, cm_math=None, cm_chunk=1, cm_stats=None,
                ref_event=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-frame Lab mean/std transfer to the reference frame(s) (nodes.py:91-124).  Two passes over HBM:
    statistics (which also stores the Lab image when cache_lab) and apply; see fused_chain.  `cm_chunk`: frames per statistics
    call of the reference (the node's batch_size, or the list of call sizes) -- it shapes the device statistics like it shapes
    the reference's (CM_STATS).  `ref_event`: the event of reference_stats_async when `ref_ms` comes from it."""
    x = _check_frames(images, channels=3)
    stats = _cm_stats(cm_stats, cm_math, x.device)
    if ref_ms is None:
        ref_ms = reference_stats(reference_image.to(x.device), cm_math, stats)
    if cache_lab or stats == "device":
        return fused_chain(x, ChainSpec(colormatch=(ref_ms, match_strength), cm_math=cm_math, cm_chunk=cm_chunk, cm_stats=stats,
                                        cm_ref_event=ref_event), out=out)
    if ref_event is not None:
        torch.cuda.current_stream().wait_event(ref_event)
    img_ms = finalize_stats(lab_stats(x, cm_math))
    res = colormatch_apply(x, img_ms, ref_ms, match_strength, cm_math)
    if out is not None:
        _out_like(x, out).copy_(res)
        return out
    return res


# ------------------------------------------------------------------------------------------------
# fused chain
# ------------------------------------------------------------------------------------------------

@dataclass
class ChainSpec:
    """grain -> LUT -> colour match -> sharpen; None disables a stage."""
    grain: Optional[tuple] = None          # (intensity, saturation_mix, chunk_frames)
    lut: Optional[tuple] = None            # (DeviceLut, strength 0..10)
    colormatch: Optional[tuple] = None     # (ref_ms [R,3,2] fp32 device tensor, match_strength)
    sharpen: Optional[tuple] = None        # (op name, strength, zero_border)
    variant: int = 0
    cm_math: object = None                 # None = default_cm_math(); "device" / "fast" (colour-match arithmetic policy)
    cm_chunk: object = 1                   # frames per statistics call of the reference (batch_size), or the list of call sizes
    cm_stats: object = None                # None = by cm_math; "device" (torch's reductions, bit for bit) / "fp64" (CM_STATS)
    cm_ref_event: object = None            # torch.cuda.Event after which ref_ms is valid (reference_stats_async); waited for before pass 2


def _chain_desc(spec: ChainSpec, plan: Optional[NoisePlan], keep, like: Optional[torch.Tensor] = None):
    d = _hip.ChainDesc()
    stages = 0
    if spec.grain is not None:
        I, s, _ = spec.grain
        stages |= _hip.STAGE_GRAIN
        d.intensity, d.sat, d.one_minus_sat = _f32(I), _f32(s), _f32(1.0 - s)
        d.noise = plan.desc()
    if spec.lut is not None:
        lut, strength = spec.lut
        mode, B, omB = blend_terms(strength)
        if mode != 0:
            if like is not None:
                _check_side(lut.table, "LUT record table", like)
            stages |= _hip.STAGE_LUT
            d.lut = lut.table.data_ptr(); d.lut_size = lut.size
            d.domain_min = _F3(*lut.domain_min); d.domain_max = _F3(*lut.domain_max)
            d.blend_mode, d.blend, d.one_minus_blend = mode, B, omB
            keep.append(lut.table)
    if spec.colormatch is not None:
        ref_ms, k = spec.colormatch
        if like is not None:
            _check_side(ref_ms, "colormatch reference statistics", like, (3, 2))
        stages |= _hip.STAGE_COLORMATCH
        d.ref_ms = ref_ms.data_ptr(); d.ref_frames = int(ref_ms.shape[0])
        d.k, d.one_minus_k = _f32(k), _f32(1.0 - k)
        keep.append(ref_ms)
    if spec.sharpen is not None:
        op, strength, zero = spec.sharpen
        stages |= _hip.STAGE_SHARPEN
        d.stencil_op = _STENCIL[op]
        d.border = _hip.BORDER_ZERO if zero else _hip.BORDER_REPLICATE
        d.strength = _f32(strength)
    d.stages = stages
    d.variant = spec.variant
    d.cm_math = _cm_math(spec.cm_math)
    return d


@_on_device
def chain_stats(images: torch.Tensor, spec: ChainSpec, generator: Optional[torch.Generator] = None, plans=None,
                lab_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Colour-match pass 1 on its own: fp64 (n, mean, M2) of Lab(grain -> LUT (images)) per frame, optionally
    storing that Lab image.  (Test / analysis entry; fused_chain runs the same kernels.)"""
    x = _check_frames(images, channels=3)
    F, H, W, _ = x.shape
    fe = H * W * 3
    segments = [(0, F, None)]
    if spec.grain is not None:
        main, tail, n_full = plans if plans is not None else plan_noise(F, fe, spec.grain[2], x.device, generator)
        segments = []
        if main is not None:
            segments.append((0, n_full * main.chunk_frames, main))
        if tail is not None:
            segments.append((F - tail.chunk_frames, tail.chunk_frames, tail))
    stats = torch.empty((F, 3, 3), dtype=torch.float64, device=x.device)
    lib = _hip.lib()
    for f0, nf, plan in segments:
        keep = []
        d = _chain_desc(ChainSpec(grain=spec.grain, lut=spec.lut, variant=spec.variant, cm_math=spec.cm_math), plan, keep, x)
        nbytes = int(lib.vrg_chain_stats_scratch_bytes(nf, H, W, C.byref(d)))
        scratch = torch.empty((max(nbytes, 8) + 7) // 8, dtype=torch.float64, device=x.device)
        src = C.c_void_p(x.data_ptr() + f0 * fe * 4)
        dst = C.c_void_p(lab_out.data_ptr() + f0 * fe * 4) if lab_out is not None else C.c_void_p(0)
        _hip.check(lib.vrg_chain_stats_lab_f32(src, dst, nf, H, W, C.byref(d), C.c_void_p(stats.data_ptr() + f0 * 72),
                                              _hip.ptr(scratch), _hip.current_stream()), "vrg_chain_stats_lab_f32")
    return stats



@_on_device
def fused_chain(images: torch.Tensor, spec: ChainSpec, generator: Optional[torch.Generator] = None, plans=None,
                out: Optional[torch.Tensor] = None, kernel_events: Optional[list] = None,
                lab_workspace: Optional[torch.Tensor] = None, cache_lab: bool = True) -> torch.Tensor:
    """One pass over HBM for grain -> LUT -> colour match -> 3x3 sharpen (colour match adds one statistics
    pass).  Bit-identical to applying the stand-alone operators in that order.

    `out` may supply the destination (bench: no allocation in the timed region); `kernel_events`, if a list,
    receives (name, start, stop, frames) tuples -- HipEvent pairs on the current stream bracketing the
    statistics pass ("stats") and the fused apply pass ("apply") of every segment.

    Chains with a colour-match stage run as two passes: (1) grain -> LUT -> Lab, reduced to per-frame statistics
    and (cache_lab=True) stored as a Lab image, (2) match -> Lab->RGB -> sharpen on that Lab image.  With
    cache_lab=False pass 2 re-evaluates grain/LUT/Lab from the input instead (36 B/px instead of 48 B/px of HBM
    traffic, but the gathers and powers twice); both forms give bit-identical results.  `lab_workspace` may
    supply the fp32 buffer (same shape as images).  With the device statistics (CM_STATS, the default) the statistics are torch's
    own reductions over the stored Lab image, per `spec.cm_chunk` frames: that form always keeps the Lab image (cache_lab is moot)."""
    u8 = isinstance(images, torch.Tensor) and images.dtype == torch.uint8       # decoded BGR frames (video routes)
    x = _check_frames(images, channels=3, dtype=torch.uint8 if u8 else torch.float32)
    if u8 and spec.colormatch is not None:
        raise ValueError("uint8 frames: colour match is not part of the video routes (convert to fp32 first)")
    esize = 1 if u8 else 4
    F, H, W, _ = x.shape
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous() or out.device != x.device:
        raise ValueError("out must be a contiguous tensor shaped and typed like images on the same device")
    if F == 0:
        return out
    fe = H * W * 3
    if spec.grain is not None and plans is not None and any(p is not None and p.chunk_frames * fe > rng.MAX_CHUNK_NUMEL for p in plans[:2]):
        raise ValueError("fused_chain: a caller-supplied noise plan cannot describe an RNG chunk of more than 2^29 elements (torch splits "
                         "such a randn into several kernels, rng.reserve_split); pass the generator instead of `plans`")
    if spec.grain is not None and plans is None and oversize_chunks(F, fe, spec.grain[2]):
        # RNG chunks that torch itself splits (see _film_grain_oversize): grain as its own pass, then the rest of the chain
        import dataclasses
        if u8:
            # decoded uint8 frames: the same detour on the fp32 image (u8 -> f32 -> chain -> u8 equals the uint8 kernels byte for byte,
            # tests/test_gpu_parity.py); costs two fp32 copies of the chunk on top of the noise buffer -- a > 0.5 G-element chunk of
            # uint8 video is not a route the reference takes at speed either
            res = fused_chain(frames_u8_to_f32(x), spec, generator=generator, kernel_events=kernel_events, cache_lab=cache_lab)
            out.copy_(f32_to_frames_u8(res))
            return out
        grained = film_grain(x, spec.grain[0], spec.grain[1], spec.grain[2], generator=generator)
        rest = dataclasses.replace(spec, grain=None)
        if rest.lut is None and rest.colormatch is None and rest.sharpen is None:
            out.copy_(grained)
            return out
        return fused_chain(grained, rest, out=out, kernel_events=kernel_events, lab_workspace=lab_workspace, cache_lab=cache_lab)
    segments = [(0, F, None)]
    if spec.grain is not None:
        main, tail, n_full = plans if plans is not None else plan_noise(F, fe, spec.grain[2], x.device, generator)
        segments = []
        if main is not None:
            segments.append((0, n_full * main.chunk_frames, main))
        if tail is not None:
            segments.append((F - tail.chunk_frames, tail.chunk_frames, tail))
    lib = _hip.lib()
    st = _hip.current_stream()
    if spec.variant == 0:
        v = _auto_variant(x.device)            # first use on a device: the toolchain self-check (a few ms, once)
        if v:
            import dataclasses
            spec = dataclasses.replace(spec, variant=v)
    device_stats = spec.colormatch is not None and _cm_stats(spec.cm_stats, spec.cm_math, x.device) == "device"
    lab_full = img_ms_full = None
    if device_stats:
        # The reference's statistics are torch reductions over each batch_size call: they need the Lab image of WHOLE calls, so
        # pass 1 runs for every segment first, then the reductions over the batch, then pass 2 (cache_lab is implied).
        # (Four schedules that overlap the passes over frame ranges were built and measured slower in round 3; they live in
        # tools/experiments/ with their A/B logs, LABNOTES.md section "schedules".)
        if lab_workspace is not None:
            if lab_workspace.shape != x.shape or lab_workspace.dtype != torch.float32 or not lab_workspace.is_contiguous():
                raise ValueError("lab_workspace must be a contiguous float32 tensor shaped like images")
            lab_full = lab_workspace
        else:
            lab_full = torch.empty((F, H, W, 3), dtype=torch.float32, device=x.device)
        for f0, nf, plan in segments:
            keep = []
            d = _chain_desc(spec, plan, keep, x)
            if kernel_events is not None:
                s0, s1 = HipEvent(), HipEvent()
                s0.record()
            _hip.check(lib.vrg_chain_stats_lab_f32(C.c_void_p(x.data_ptr() + f0 * fe * 4), C.c_void_p(lab_full.data_ptr() + f0 * fe * 4), nf, H, W,
                                                  C.byref(d), None, None, st), "vrg_chain_stats_lab_f32")       # the Lab image only
            if kernel_events is not None:
                s1.record()
                kernel_events.append(("stats", s0, s1, nf))
        if kernel_events is not None:
            t0, t1 = HipEvent(), HipEvent()
            t0.record()
        img_ms_full = lab_stats_device(lab_full, spec.cm_chunk)
        if kernel_events is not None:
            t1.record()
            kernel_events.append(("tstats", t0, t1, F))
    if spec.colormatch is not None and spec.cm_ref_event is not None:
        torch.cuda.current_stream().wait_event(spec.cm_ref_event)          # reference_stats_async: ref_ms is valid from here on
    for f0, nf, plan in segments:
        keep = []
        d = _chain_desc(spec, plan, keep, x)
        if d.stages == 0:
            out[f0:f0 + nf] = x[f0:f0 + nf]
            continue
        src = C.c_void_p(x.data_ptr() + f0 * fe * esize)
        dst = C.c_void_p(out.data_ptr() + f0 * fe * esize)
        if d.stages & _hip.STAGE_COLORMATCH:
            if d.ref_frames != 1 and (nf % d.ref_frames or f0 % d.ref_frames):
                raise RuntimeError("reference_image batch must be 1 or divide the frame batch")
            if device_stats:
                src = C.c_void_p(lab_full.data_ptr() + f0 * fe * 4)
                d.stages = (d.stages & _hip.STAGE_SHARPEN) | _hip.STAGE_COLORMATCH | _hip.STAGE_FROM_LAB
                d.img_ms = img_ms_full.data_ptr() + f0 * 24
            else:
                stats = torch.empty((nf, 3, 3), dtype=torch.float64, device=x.device)
                nbytes = int(lib.vrg_chain_stats_scratch_bytes(nf, H, W, C.byref(d)))
                scratch = torch.empty((max(nbytes, 8) + 7) // 8, dtype=torch.float64, device=x.device)
                if kernel_events is not None:
                    s0, s1 = HipEvent(), HipEvent()
                    s0.record()
                if cache_lab:
                    if lab_workspace is not None:
                        if lab_workspace.shape != x.shape or lab_workspace.dtype != torch.float32 or not lab_workspace.is_contiguous():
                            raise ValueError("lab_workspace must be a contiguous float32 tensor shaped like images")
                        lab = lab_workspace[f0:f0 + nf]
                    else:
                        lab = torch.empty((nf, H, W, 3), dtype=torch.float32, device=x.device)
                    keep.append(lab)
                    _hip.check(lib.vrg_chain_stats_lab_f32(src, _hip.ptr(lab), nf, H, W, C.byref(d), _hip.ptr(stats), _hip.ptr(scratch), st),
                               "vrg_chain_stats_lab_f32")
                    src = C.c_void_p(lab.data_ptr())
                    d.stages = (d.stages & _hip.STAGE_SHARPEN) | _hip.STAGE_COLORMATCH | _hip.STAGE_FROM_LAB
                else:
                    _hip.check(lib.vrg_chain_stats_f32(src, nf, H, W, C.byref(d), _hip.ptr(stats), _hip.ptr(scratch), st),
                               "vrg_chain_stats_f32")
                if kernel_events is not None:
                    s1.record()
                    kernel_events.append(("stats", s0, s1, nf))
                img_ms = finalize_stats(stats)
                d.img_ms = img_ms.data_ptr()
                keep.append(img_ms)
        if kernel_events is not None:
            e0, e1 = HipEvent(), HipEvent()
            e0.record()
        if u8:
            _hip.check(lib.vrg_fused_chain_u8(src, dst, nf, H, W, C.byref(d), st), "vrg_fused_chain_u8")
        else:
            _hip.check(lib.vrg_fused_chain_f32(src, dst, nf, H, W, C.byref(d), st), "vrg_fused_chain_f32")
        if kernel_events is not None:
            e1.record()
            kernel_events.append(("apply", e0, e1, nf))
    return out


def fused_stages(frames: torch.Tensor, first_frame: int, stages: dict, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """What `_devices` runs when consecutive nodes of this pack were called on one another's results (deferred graph fusion): the
    recorded nodes -- ``stages[kind]`` for kind in grain / lut / colormatch / sharpen, each the parameters its node was called with -- as
    ONE fused chain over frames [first_frame, first_frame + n) of the ORIGINAL input.  Bit-identical to the nodes run one after the
    other (tests/test_gpu_parity.py): grain draws from the generator range its node reserved when it was called (``plans``, sliced per
    piece), the colour statistics are reduced over the node's batch_size calls (``calls_of``), the reference statistics come from the
    side stream they were queued on at call time (``ref_event``).  Reference call order: nodes.py:41-66 -> VRGDG_IV_Adjustments.py:345-361
    -> nodes.py:91-124 -> nodes.py:156-209."""
    n = int(frames.shape[0])
    g, l, c, sh = (stages.get(k) for k in ("grain", "lut", "colormatch", "sharpen"))
    spec = ChainSpec(grain=(g["I"], g["s"], g["step"]) if g else None,
                     lut=(l["lut"], l["strength"]) if l else None,
                     colormatch=(c["ref_ms"], c["k"]) if c else None,
                     sharpen=(sh["op"], sh["strength"], sh["zero"]) if sh else None,
                     cm_chunk=c["calls_of"](first_frame, n) if c else 1,
                     cm_ref_event=c["ref_event"] if c else None)
    plans = slice_plans(g["plans"], first_frame, n) if g else None
    return fused_chain(frames, spec, plans=plans, out=out)


# ------------------------------------------------------------------------------------------------
# timing helper (HIP events on the stream the kernels run on)
# ------------------------------------------------------------------------------------------------

class HipEvent:
    """A hipEvent_t on the kernels' stream.  Handles are pooled per process: creating and destroying an event costs a driver call each
    (bench.py brackets every pass of every step), recording a pooled one does not."""
    _pool = {}          # device index -> free handles (an event belongs to the device it was created on)

    def __init__(self):
        self._dev = torch.cuda.current_device()
        with _STATE_LOCK:
            free = HipEvent._pool.get(self._dev)
            self._ev = free.pop() if free else None
        if self._ev is None:
            self._ev = C.c_void_p()
            _hip.check(_hip.lib().vrg_event_create(C.byref(self._ev)), "vrg_event_create")

    def record(self):
        _hip.check(_hip.lib().vrg_event_record(self._ev, _hip.current_stream()), "vrg_event_record")

    def elapsed_ms(self, stop: "HipEvent") -> float:
        ms = C.c_float()
        _hip.check(_hip.lib().vrg_event_elapsed_ms(self._ev, stop._ev, C.byref(ms)), "vrg_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self._ev:
                with _STATE_LOCK:
                    free = HipEvent._pool.setdefault(self._dev, [])
                    if len(free) < 256:
                        free.append(self._ev)
                        self._ev = None
                if self._ev:
                    _hip.lib().vrg_event_destroy(self._ev)
        except Exception:
            pass
