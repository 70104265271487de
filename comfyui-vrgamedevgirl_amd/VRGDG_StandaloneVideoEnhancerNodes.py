"""Per-batch effect helpers of the stand-alone video enhancer
(VRGDG_StandaloneVideoEnhancerNodes.py:233-308 of the reference): unsharp, then per-frame-seeded grain.
Per-frame seeding makes the result independent of batch boundaries and of how frames are sharded across GPUs
(the property the reference tests at tests/test_standalone_video_enhancer.py:39-61).  Job control, segment
files and the ffmpeg mux around these helpers are out of scope."""
from __future__ import annotations

import torch

from . import ops
from ._devices import compute_device


def _apply_unsharp(images, strength, use_gpu):
    if strength <= 0:
        return images
    return ops.stencil3x3(images, "unsharp", strength, zero_border=bool(use_gpu))


def _apply_seeded_grain(images, intensity, saturation_mix, seed, frame_start):
    if intensity <= 0:
        return images
    return ops.film_grain_seeded_frames(images, intensity, saturation_mix, seed, frame_start)


def _apply_effects_batch(images, settings, frame_start=0):
    """sharpen (optional) then seeded grain (optional).  A CPU tensor comes back as a CPU tensor like in the
    reference; a GPU tensor (what this module's _frames_to_tensor produces) stays on the GPU so that the enhancer loop
    decode -> _frames_to_tensor -> _process_with_retry -> _tensor_to_frames crosses PCIe once each way, in uint8."""
    use_gpu_flag = bool(settings.get("use_gpu", True))
    batch = images if images.is_cuda else images.to(compute_device())
    batch = batch.to(torch.float32)
    sharpen, grain = bool(settings.get("sharpen_enabled", True)), bool(settings.get("grain_enabled", False))
    strength, intensity = float(settings.get("sharpen_strength", 0.5)), float(settings.get("grain_intensity", 0.04))
    if sharpen and grain and strength > 0 and intensity > 0:
        # both effects: one pass over the frames (the same bits as the two helpers below, one after the other)
        batch = ops.sharpen_then_seeded_grain(batch, strength, use_gpu_flag, intensity, float(settings.get("saturation_mix", 0.5)),
                                              int(settings.get("seed", 42)), int(frame_start))
    else:
        if sharpen:
            batch = _apply_unsharp(batch, strength, use_gpu_flag)
        if grain:
            batch = _apply_seeded_grain(batch, intensity, float(settings.get("saturation_mix", 0.5)), int(settings.get("seed", 42)),
                                        int(frame_start))
    return batch.detach() if images.is_cuda else batch.detach().cpu()


def _process_with_retry(images, settings, frame_start):
    """Halve the batch on device OOM (same contract as the reference: returns (frames, smallest batch used))."""
    try:
        return _apply_effects_batch(images, settings, frame_start), len(images)
    except RuntimeError as exc:
        if "out of memory" not in str(exc).lower() or len(images) <= 1:
            raise
        torch.cuda.empty_cache()
        mid = max(1, len(images) // 2)
        left, ls = _process_with_retry(images[:mid], settings, frame_start)
        right, rs = _process_with_retry(images[mid:], settings, frame_start + mid)
        return torch.cat((left, right), dim=0), min(ls, rs)


def _frames_to_tensor(frames):
    """BGR uint8 frames -> fp32 RGB in [0,1] (:311-316 of the reference); 3 B/px cross PCIe, the conversion runs
    on the GPU and the tensor stays there for _process_with_retry."""
    from .VRGDG_LUTVideoTools import _frames_to_tensor as impl
    return impl(frames)


def _tensor_to_frames(tensor):
    """fp32 RGB -> list of BGR uint8 frames (:319-324 of the reference)."""
    from .VRGDG_LUTVideoTools import _tensor_to_frames as impl
    return impl(tensor)
