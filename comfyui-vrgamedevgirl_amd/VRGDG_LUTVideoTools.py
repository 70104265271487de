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
