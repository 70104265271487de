"""Per-batch effect helpers of the stand-alone video enhancer
(VRGDG_StandaloneVideoEnhancerNodes.py:233-308 of the reference): unsharp, then per-frame-seeded grain.
Per-frame seeding makes the result independent of batch boundaries and of how frames are sharded across GPUs
(the property the reference tests at tests/test_standalone_video_enhancer.py:39-61).  Job control, segment
files and the ffmpeg mux around these helpers are out of scope."""
from __future__ import annotations

import torch

from . import ops
from ._devices import compute_device


class DecodedFrames:
    """What this module's ``_frames_to_tensor`` hands to the render loop (reference :417-421:
    ``tensor = _frames_to_tensor(frames); enhanced, n = _process_with_retry(tensor, settings, i); _tensor_to_frames(enhanced)``):
    the decoded batch as ONE uint8 B,G,R tensor on the GPU, with the ``/ 255`` conversion DEFERRED.  ``_apply_effects_batch`` runs
    the whole loop body on it as one uint8 -> uint8 kernel (ops.sharpen_then_seeded_grain: 3 + 3 B/px instead of the 54 B/px of
    converter, fp32 effects, converter) and ``_tensor_to_frames`` only downloads the bytes; the results are byte-identical to the
    reference's fp32 route.  Anything else that wants the reference's fp32 tensor calls ``float_tensor()``."""

    def __init__(self, frames_u8: torch.Tensor):
        self.u8 = frames_u8

    def __len__(self):
        return int(self.u8.shape[0])

    def __getitem__(self, index):
        if not isinstance(index, slice):
            raise TypeError("DecodedFrames supports batch slices only")
        return DecodedFrames(self.u8[index])

    @property
    def shape(self):
        return self.u8.shape

    @property
    def is_cuda(self):
        return self.u8.is_cuda

    def float_tensor(self) -> torch.Tensor:
        """fp32 R,G,B [F,H,W,3] in [0,1]: ``np.stack(rgb).astype(float32) / 255.0`` (reference :311-316), on the GPU."""
        return ops.frames_u8_to_f32(self.u8)

    @staticmethod
    def cat(parts):
        return DecodedFrames(torch.cat([p.u8 for p in parts], dim=0))


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
    if isinstance(images, DecodedFrames):
        # decoded uint8 frames: the conversions and both effects in one pass (or the converter route for what that kernel refuses)
        sharpen_on, grain_on = bool(settings.get("sharpen_enabled", True)), bool(settings.get("grain_enabled", False))
        out = ops.sharpen_then_seeded_grain(images.u8, float(settings.get("sharpen_strength", 0.5)) if sharpen_on else 0.0, use_gpu_flag,
                                            float(settings.get("grain_intensity", 0.04)) if grain_on else 0.0,
                                            float(settings.get("saturation_mix", 0.5)), int(settings.get("seed", 42)), int(frame_start))
        return DecodedFrames(out)
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
        if isinstance(left, DecodedFrames):
            return DecodedFrames.cat((left, right)), min(ls, rs)
        return torch.cat((left, right), dim=0), min(ls, rs)


def _frames_to_tensor(frames):
    """BGR uint8 frames -> the batch on the GPU, 3 B/px over PCIe, as ``DecodedFrames`` (the ``/ 255`` of reference :311-316 happens
    inside the effects kernel; ``.float_tensor()`` is the reference's fp32 tensor)."""
    from .VRGDG_LUTVideoTools import _stack_frames
    return DecodedFrames(_stack_frames(frames))


def _tensor_to_frames(tensor):
    """fp32 RGB (or the DecodedFrames ``_apply_effects_batch`` returned) -> list of BGR uint8 frames (:319-324 of the reference)."""
    if isinstance(tensor, DecodedFrames):
        from .VRGDG_LUTVideoTools import _unstack_frames
        return _unstack_frames(tensor.u8)
    from .VRGDG_LUTVideoTools import _tensor_to_frames as impl
    return impl(tensor)
