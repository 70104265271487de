"""TEST INFRASTRUCTURE ONLY -- the oracle composition of the fused chains at the geometry the benchmark runs.

``headline_chain`` evaluates grain -> LUT -> colour match -> unsharp the way the reference's nodes would, one after the other
(nodes.py:41-66, VRGDG_IV_Adjustments.py:288-361, nodes.py:91-124, nodes.py:182-209), for WHOLE RNG chunks of a job-wide noise
stream:
  * the noise of chunk c is ``torch.randn(chunk shape, device=dev, generator=g)`` with g at (seed, offset0 + c * offset_stride) --
    torch's own Philox stream on the GPU, what ``torch.randn_like`` draws there;
  * grain, LUT and unsharp are the CPU restatements pinned to the reference's fixtures (oracle/restated.py);
  * colour match is the restated formulas evaluated by torch ON THE DEVICE (what the reference computes when ComfyUI hands it the GPU).
Used by tests/test_gpu_parity.py (bench-geometry tests) and by bench.py's --verify leg (outside the timed region, as the checker).
Nothing here is imported by the product.
"""
from __future__ import annotations

import torch

from . import restated as R


def chunk_noise(shape, dev, seed: int, offset: int) -> torch.Tensor:
    """torch.randn of one chunk on the device with the generator at (seed, offset); returned on the CPU."""
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    g.set_offset(int(offset))
    return torch.randn(shape, device=dev, generator=g).cpu()


def headline_chain(x_cpu: torch.Tensor, dev, *, stages, stream=None, chunk0: int = 0, chunk_frames: int = 4, grain=(0.04, 0.5),
                   lut_cpu=None, lut_strength: float = 10.0, reference_dev=None, match_strength: float = 1.0, cm_batch: int = 1,
                   unsharp=(0.5, False)) -> torch.Tensor:
    """`x_cpu`: [n * chunk_frames, H, W, 3] fp32 CPU frames = the job's RNG chunks chunk0 .. chunk0 + n - 1; `stream`: the job-wide
    noise description (seed, offset0, offset_stride: comfyui-vrgamedevgirl_amd/rng.py ChunkedStream); `stages`: names out of
    ("grain", "lut", "colormatch", "sharpen").  Returns the CPU result."""
    y = x_cpu
    if "grain" in stages:
        if y.shape[0] % chunk_frames:
            raise ValueError("whole RNG chunks only")

        def noise_fn(first_frame, shape):
            c = chunk0 + first_frame // chunk_frames
            return chunk_noise(shape, dev, stream.seed, stream.offset0 + c * stream.offset_stride)

        y = R.fast_film_grain(y, grain[0], grain[1], chunk_frames, noise_fn=noise_fn)
    if "lut" in stages:
        y = R.apply_lut_with_strength(y, lut_cpu, lut_strength)
    if "colormatch" in stages:
        y = R.color_match(y.to(dev), reference_dev, match_strength, cm_batch).cpu()
    if "sharpen" in stages:
        y = R.unsharp(y, unsharp[0], unsharp[1])
        if not y.is_contiguous():
            y = y.contiguous()
    return y
