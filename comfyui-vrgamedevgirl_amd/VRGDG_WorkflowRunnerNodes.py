"""Opening colour match of a new clip (``_apply_scene_start_color_match``, VRGDG_WorkflowRunnerNodes.py:4334-4432 of
the reference): match the first frame of a new clip to the last frame of the previous one with a per-channel
gain / offset LUT, and fade the correction out over the first seconds.

The reference drives ffmpeg for everything around the numbers (frame extraction to PNG, ``lut3d`` + ``blend`` filter
graph, re-encode).  What is *its own arithmetic* -- the channel statistics (PIL.ImageStat), scales / offsets, the 17^3
``.cube`` it writes and the blend-weight expression -- is reproduced here exactly, with the statistics as one exact
integer reduction on the GPU (``vrg_u8_channel_sums``).  The per-pixel part that ffmpeg performs (LUT lookup + fade
blend) runs on the decoded uint8 frames, with either arithmetic (``filter_arithmetic``):
  "nodes"   this package's LUT stage (trilinear, fp32 blend ``x*(1-w) + y*w``, one quantisation at the end): bit-exact
            against the oracle composition of the reference's own LUT node arithmetic;
  "ffmpeg"  ffmpeg's filters as published (tetrahedral ``lut3d`` with its 8-bit truncation, then the ``blend`` expression in
            double, truncated): a restatement of a third-party implementation that is absent here -- parity unpinned, and
            the yuv420p <-> RGB conversions ffmpeg inserts around the filters are not part of it.
The LUT is affine per channel below the clamp, so the two interpolations agree up to the 8-bit rounding rule.
The media plumbing (paths, ffmpeg, thumbnails) is out of scope."""
from __future__ import annotations

import math
import os
import tempfile

import numpy as np
import torch

from . import _hip, cube, ops
from ._devices import compute_device

CUBE_SIZE = 17


def _as_gpu_u8(frames) -> torch.Tensor:
    """One HxWx3 frame, a list of frames or an [F,H,W,3] array / tensor of uint8 -> [F,H,W,3] uint8 on the GPU."""
    if isinstance(frames, torch.Tensor):
        t = frames if frames.ndim == 4 else frames.unsqueeze(0)
        if t.dtype != torch.uint8 or t.shape[-1] != 3:
            raise ValueError("frames must be uint8 HxWx3")
        return t.to(compute_device()).contiguous()
    a = np.asarray(frames)
    if a.ndim == 3:
        a = a[None]
    if a.dtype != np.uint8 or a.ndim != 4 or a.shape[-1] != 3:
        raise ValueError("frames must be uint8 HxWx3")
    return torch.from_numpy(np.ascontiguousarray(a)).to(compute_device())


def _frame_channel_stats(frame_bgr):
    """``ImageStat.Stat(image.convert("RGB")).mean[:3] / .stddev[:3]`` (:4382-4389) of one decoded B,G,R frame, in
    R,G,B order: exact 64-bit sums on the GPU, then ImageStat's own double arithmetic
    (``mean = sum / n``, ``var = (sum2 - sum**2.0 / n) / n``, ``stddev = sqrt(var)``)."""
    x = _as_gpu_u8(frame_bgr)
    if x.shape[0] != 1:
        raise ValueError("one frame expected")
    _, H, W, _ = x.shape
    sums = torch.empty((1, 3, 2), dtype=torch.int64, device=x.device)
    _hip.check(_hip.lib().vrg_u8_channel_sums(_hip.ptr(x), 1, H, W, _hip.ptr(sums), _hip.current_stream()), "vrg_u8_channel_sums")
    s = sums.cpu().tolist()[0]
    n = H * W
    mean, std = [], []
    for c in (2, 1, 0):                                   # memory order is B,G,R
        total, total2 = float(s[c][0]), float(s[c][1])
        mean.append(total / n)
        std.append(math.sqrt((total2 - (total ** 2.0) / n) / n))
    return mean, std


def _opening_color_match_terms(reference_stats, target_stats):
    """(:4386-4391) stddevs floored at 1, per-channel gain clamped to [0.25, 4], offset = ref_mean - tgt_mean * gain."""
    (rm, rs), (tm, ts) = reference_stats, target_stats
    reference_std = [max(1.0, float(v)) for v in rs[:3]]
    target_std = [max(1.0, float(v)) for v in ts[:3]]
    scales = [max(0.25, min(4.0, reference_std[i] / target_std[i])) for i in range(3)]
    offsets = [float(rm[i]) - float(tm[i]) * scales[i] for i in range(3)]
    return scales, offsets


def _opening_color_match_cube_text(scales, offsets, cube_size=CUBE_SIZE) -> str:
    """The ``.cube`` handed to lut3d (:4393-4405), character for character."""
    parts = ['TITLE "VRGDG opening color match"\n',
             f"LUT_3D_SIZE {cube_size}\nDOMAIN_MIN 0.0 0.0 0.0\nDOMAIN_MAX 1.0 1.0 1.0\n"]
    top = cube_size - 1
    for blue in range(cube_size):
        for green in range(cube_size):
            for red in range(cube_size):
                node = (red, green, blue)
                r, g, b = (max(0.0, min(1.0, ((node[i] / top) * 255.0 * scales[i] + offsets[i]) / 255.0)) for i in range(3))
                parts.append(f"{r:.8f} {g:.8f} {b:.8f}\n")
    return "".join(parts)


def _opening_color_match_weight(frame_index, fps, strength, fade_seconds) -> float:
    """``max(0, min(1, strength * (1 - T / fade)))`` at ``T = frame_index / fps`` (:4407); the filter text carries
    strength and fade with six decimals, so those are the values that count."""
    s6, f6 = float(f"{strength:.6f}"), float(f"{fade_seconds:.6f}")
    return max(0.0, min(1.0, s6 * (1.0 - (frame_index / float(fps)) / f6)))


def _apply_scene_start_color_match_frames(frames, reference_frame, fps, fade_seconds=1.0, strength=0.85, target_frame=None,
                                          first_frame_index=0, filter_arithmetic="nodes"):
    """Frame-level form of the reference routine: `frames` = decoded B,G,R uint8 frames of the new clip starting at
    `first_frame_index`, `reference_frame` = last frame of the previous clip, `target_frame` = first frame of the new
    clip (default: ``frames[0]``).  Returns ``(frames_out, info)`` with the same clamps as the reference
    (fade 0.05..30 s, strength 0..1; strength 0 -> unchanged, ``applied: False``)."""
    if filter_arithmetic not in ("nodes", "ffmpeg"):
        raise ValueError("filter_arithmetic must be 'nodes' or 'ffmpeg'")
    fade_seconds = max(0.05, min(30.0, float(fade_seconds or 1.0)))
    strength = max(0.0, min(1.0, float(strength or 0.85)))
    batch = _as_gpu_u8(frames)
    if strength <= 0.0:
        return [f for f in batch.cpu().numpy()], {"applied": False, "reason": "strength is zero"}
    ref_stats = _frame_channel_stats(reference_frame)
    tgt_stats = _frame_channel_stats(batch[0] if target_frame is None else target_frame)
    scales, offsets = _opening_color_match_terms(ref_stats, tgt_stats)
    text = _opening_color_match_cube_text(scales, offsets)
    with tempfile.TemporaryDirectory() as tmp:               # through the parser: the values lut3d reads are the 8-decimal ones
        path = os.path.join(tmp, "opening.cube")
        with open(path, "w", encoding="utf-8", newline="\n") as handle:
            handle.write(text)
        lut_data = cube.parse_cube_file(path)
    dev_lut = ops.upload_lut(lut_data, batch.device)
    weights = [_opening_color_match_weight(first_frame_index + i, fps, strength, fade_seconds) for i in range(batch.shape[0])]
    if filter_arithmetic == "ffmpeg":
        out = ops.lut3d_ffmpeg_u8(batch, dev_lut, weights)        # w = 0 leaves a frame unchanged: A*1 + B*0
    else:
        out = batch.clone()
        for i, w in enumerate(weights):
            if w > 0.0:
                ops.fused_chain(batch[i:i + 1], ops.ChainSpec(lut=(dev_lut, 10.0 * w)), out=out[i:i + 1])
    info = {"applied": True, "scales": scales, "offsets": offsets, "weights": weights, "cube_text": text,
            "reference_mean": ref_stats[0], "reference_std": ref_stats[1], "target_mean": tgt_stats[0], "target_std": tgt_stats[1]}
    return [f for f in out.cpu().numpy()], info
