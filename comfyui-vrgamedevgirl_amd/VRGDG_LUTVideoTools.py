"""Tensor helpers the pack's video routes call between decode and encode
(VRGDG_LUTVideoTools.py:172-185, 262-277 of the reference): the second call site of the LUT and grain math.
The cv2 / ffmpeg media loop around them is out of scope (codec-bound)."""
from __future__ import annotations

import torch

from . import ops
from .VRGDG_IV_Adjustments import LUTS_DIR, VRGDG_LUTS  # noqa: F401  (names the routes import)
from ._devices import compute_device


def _on_gpu(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_cuda else t.to(compute_device())


def _apply_lut_tensor(image_tensor, lut_name, strength, device):
    """LUT + strength blend for one decoded batch; result stays on the GPU (the caller converts to uint8)."""
    lut_data = VRGDG_LUTS._load_lut(lut_name)
    src = _on_gpu(image_tensor).to(torch.float32)
    return ops.lut3d(src, ops.upload_lut(lut_data, src.device), strength)


def _apply_film_grain_tensor(image_tensor, grain_intensity=0.04, saturation_mix=0.5, device="cpu", seed=None):
    """Grain for one decoded batch: intensity / saturation clamped to [0,1]; with a seed the whole batch is one
    ``torch.randn`` draw from a fresh generator seeded with it (callers pass seed + frame_offset per batch)."""
    intensity = max(0.0, min(1.0, float(grain_intensity)))
    saturation = max(0.0, min(1.0, float(saturation_mix)))
    src = _on_gpu(image_tensor).to(torch.float32)
    gen = None
    if seed not in (None, ""):
        gen = torch.Generator(device=src.device)
        gen.manual_seed(int(seed))
    return ops.film_grain(src, intensity, saturation, chunk_frames=0, generator=gen)


# bounds of the 13 sliders (VRGDG_LUTVideoTools.py:282-297 of the reference)
_ADJUST_BOUNDS = {
    "temperature": (-100.0, 100.0), "tint": (-100.0, 100.0), "saturation": (-100.0, 100.0),
    "exposure": (-100.0, 100.0), "contrast": (-100.0, 100.0), "highlights": (-100.0, 100.0),
    "shadows": (-100.0, 100.0), "whites": (-100.0, 100.0), "blacks": (-100.0, 100.0),
    "sharpen": (0.0, 100.0), "clarity": (-100.0, 100.0), "vignette": (0.0, 100.0), "fade": (0.0, 100.0),
}


def _slider_value(raw) -> float:
    try:
        return float(raw)
    except Exception:               # None, "", a list ...: the reference reads such a slider as 0
        return 0.0


def _normalize_adjust_settings(settings=None):
    """The routes' settings contract (reference :280-304; fixtures tests/golden/adjust_normalized.json): anything but a dict means
    defaults, an unparsable slider is 0, every slider is clamped to its range (`max(lo, min(hi, v))`: a NaN clamps to `hi`), and
    ``enabled`` is true unless it is literally ``False``."""
    given = settings if isinstance(settings, dict) else {}
    out = {"enabled": given.get("enabled", True) is not False}
    out.update((name, max(lo, min(hi, _slider_value(given.get(name, 0.0))))) for name, (lo, hi) in _ADJUST_BOUNDS.items())
    return out


def _is_gpu_device(device) -> bool:
    """Does the reference, called with this `device`, do its arithmetic on a GPU?  (The routes pass "cuda" when one is
    available; all compute here is on the MI355X either way -- the argument only selects which of the reference's two
    arithmetics, torch-CPU or torch-GPU, is reproduced bit for bit.)"""
    return str(getattr(device, "type", device) or "cpu").strip().lower().split(":")[0] not in ("cpu", "")


def _apply_adjust_tensor(image_tensor, settings=None, device="cpu"):
    """13-slider Adjust for one decoded batch (:307-391 of the reference): one or two HIP passes
    (csrc/vrg_adjust.hip); result stays on the GPU."""
    adjust = _normalize_adjust_settings(settings)
    src = _on_gpu(image_tensor).to(torch.float32)
    return ops.adjust(src, ops.adjust_terms(adjust, device_math=_is_gpu_device(device)))


# ------------------------------------------------------------------------------------------------
# uint8 BGR frames at the codec edge (:736-752, :1365-1386 of the reference)
# ------------------------------------------------------------------------------------------------

def _stack_frames(frames) -> torch.Tensor:
    """List of decoded ``HxWx3`` uint8 B,G,R frames (cv2.VideoCapture.read) -> one uint8 tensor on the GPU:
    3 B/px over PCIe instead of the 12 B/px of an fp32 tensor.  Each frame is uploaded straight from the decoder's
    buffer into its slot of the batch (no host-side np.stack: that single-threaded copy cost 5x the DMA)."""
    import numpy as np
    arrays = [np.asarray(f) for f in frames]
    if not arrays:
        raise ValueError("frames must not be empty")
    shape = arrays[0].shape
    for a in arrays:
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[-1] != 3 or a.shape != shape:
            raise ValueError("frames must be HxWx3 uint8 arrays of one size")
    batch = torch.empty((len(arrays),) + tuple(shape), dtype=torch.uint8, device=compute_device())
    for i, a in enumerate(arrays):
        if not (a.flags.c_contiguous and a.flags.writeable):
            a = np.array(a, order="C")           # torch.from_numpy wants a writable, dense buffer
        batch[i].copy_(torch.from_numpy(a), non_blocking=True)
    return batch


def _unstack_frames(frames_u8: torch.Tensor):
    """GPU uint8 batch -> list of HxWx3 numpy frames (views of one page-locked download; the caching host allocator
    makes that buffer free after the first batch, and the DMA does not page-fault a fresh array)."""
    host = torch.empty(frames_u8.shape, dtype=torch.uint8, pin_memory=True)
    host.copy_(frames_u8, non_blocking=True)
    torch.cuda.current_stream(frames_u8.device).synchronize()
    array = host.numpy()
    return [array[i] for i in range(array.shape[0])]


def _frames_to_tensor(frames):
    """BGR uint8 frames -> fp32 RGB ``[F,H,W,3]`` in [0,1] (``astype(float32) / 255.0``); result on the GPU."""
    return ops.frames_u8_to_f32(_stack_frames(frames))


def _tensor_to_frames(tensor):
    """fp32 RGB tensor -> list of BGR uint8 frames (``clip(x * 255, 0, 255).astype(uint8)``: truncation)."""
    return _unstack_frames(ops.f32_to_frames_u8(_on_gpu(tensor.detach()).to(torch.float32)))


def _process_video_batch(batch, writer, lut_name, strength, target_device):
    """Decode batch -> LUT -> encoder, with both conversions inside the LUT kernel (uint8 in, uint8 out)."""
    lut_data = VRGDG_LUTS._load_lut(lut_name)
    src = _stack_frames(batch)
    out = ops.fused_chain(src, ops.ChainSpec(lut=(ops.upload_lut(lut_data, src.device), strength)))
    for frame in _unstack_frames(out):
        writer.write(frame)
    return len(batch)


def _process_film_grain_batch(batch, writer, grain_intensity, saturation_mix, target_device, seed=None):
    intensity = max(0.0, min(1.0, float(grain_intensity)))
    saturation = max(0.0, min(1.0, float(saturation_mix)))
    src = _stack_frames(batch)
    gen = None
    if seed not in (None, ""):
        gen = torch.Generator(device=src.device)
        gen.manual_seed(int(seed))
    out = ops.fused_chain(src, ops.ChainSpec(grain=(intensity, saturation, 0)), generator=gen)
    for frame in _unstack_frames(out):
        writer.write(frame)
    return len(batch)


def _process_adjust_batch(batch, writer, settings, target_device):
    src = _stack_frames(batch)
    out = ops.adjust(src, ops.adjust_terms(_normalize_adjust_settings(settings), device_math=_is_gpu_device(target_device)))
    for frame in _unstack_frames(out):
        writer.write(frame)
    return len(batch)
