"""Device policy shared by the node classes: where to compute, where ComfyUI wants results, and how to
stream a CPU-resident batch through the GPU in bounded pieces."""
from __future__ import annotations

import torch

#: upper bound of one host->device staging transfer (bytes of fp32 frames)
STAGE_BYTES = 1 << 30


def compute_device() -> torch.device:
    """ComfyUI's torch device (comfy.model_management.get_torch_device, as nodes.py:41 of the reference does);
    outside ComfyUI the current HIP device.  There is no CPU fallback."""
    dev = None
    try:
        import comfy.model_management as mm  # type: ignore
        dev = torch.device(mm.get_torch_device())
    except Exception:
        dev = None
    if dev is None or dev.type != "cuda":
        if not torch.cuda.is_available():
            raise RuntimeError("comfyui-vrgamedevgirl_amd: no AMD GPU visible to PyTorch-ROCm; these nodes are "
                               "MI355X-native and have no CPU path")
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def intermediate_device() -> torch.device:
    try:
        import comfy.model_management as mm  # type: ignore
        return torch.device(mm.intermediate_device())
    except Exception:
        return torch.device("cpu")


def frame_groups(n_frames: int, frame_bytes: int, multiple_of: int = 1):
    """Yield (start, stop) frame ranges of at most STAGE_BYTES, each a multiple of `multiple_of` frames
    (except the last)."""
    if n_frames <= 0:
        return
    per = max(1, STAGE_BYTES // max(frame_bytes, 1))
    per = max(multiple_of, (per // multiple_of) * multiple_of)
    for s in range(0, n_frames, per):
        yield s, min(n_frames, s + per)
