"""ComfyUI node classes of the per-pixel video post-processing path, MI355X-native.

Operator surface (mapping keys, INPUT_TYPES widget specs, RETURN_TYPES, FUNCTION, CATEGORY, DESCRIPTION,
argument names and order) is the reference's -- nodes.py:18-384 and :1881-1933 there -- so these classes
drop into any existing graph.  The bodies only move frames and call the HIP operators in ``ops``.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import ops
from ._devices import Stage, compute_device, defer, frame_groups, intermediate_device, on_device, stream_frames

_IMAGE = ("IMAGE",)


def _float_widget(default, lo, hi, step):
    return ("FLOAT", {"default": default, "min": lo, "max": hi, "step": step})


def _frame_numel(images: torch.Tensor) -> int:
    """Elements of one frame, from the shape alone (indexing a result of a previous node of this pack would download it: _devices.LazyFrames)."""
    n = 1
    for d in images.shape[1:]:
        n *= int(d)
    return n


def _frames_bytes(images: torch.Tensor) -> int:
    return _frame_numel(images) * 4 if images.shape[0] else 0


#: False selects the plain sequential upload / run / download per group (kept for A/B measurements)
PIPELINED = True


def _run_grouped(images, fn, multiple_of=1, fn_for_device=None, kind=None, fuse=None, stage_for_device=None):
    """Stream `images` through the GPU in bounded groups; `fn(gpu_frames, first_frame, out=None)` returns GPU frames.  `fn_for_device(device)`
    builds that callable for one of several GPUs (VRGDG_DEVICES, _devices.stream_frames); `fn` is the compute device's.  `kind` / `fuse`:
    what this node is to ops.fused_chain (grain / lut / colormatch / sharpen and its parameters) -- with them the call may be DEFERRED and
    fused with the nodes of this pack that follow it in the graph (_devices.defer)."""
    dev = compute_device()
    out_dev = intermediate_device()
    # (`stage_for_device(device) -> (fn, fuse)`: the stage for another GPU of VRGDG_DEVICES -- nodes whose stage holds device-resident operands)
    stage = Stage(kind, fn, multiple_of, fuse, for_device=stage_for_device) if kind is not None else None
    if images.is_cuda:
        if stage is not None:
            res = defer(images, dev, stage, out_dev)
            if res is not None:
                return res
        return fn(images.to(dev), 0).to(out_dev)
    if images.dtype != torch.float32:
        images = images.float()
    if PIPELINED and out_dev.type == "cpu" and images.shape[0] > 0:
        # CPU in, CPU out (ComfyUI's default): H2D, kernels and D2H overlapped on three streams
        return stream_frames(images, fn, multiple_of, fn_for_device=fn_for_device, stage=stage)
    pieces = []
    for s, e in frame_groups(images.shape[0], _frames_bytes(images), multiple_of):
        pieces.append(fn(images[s:e].to(dev), s).to(out_dev))
    if not pieces:
        return torch.empty_like(images, device=out_dev)
    return pieces[0] if len(pieces) == 1 else torch.cat(pieces, dim=0)


class FastFilmGrain:
    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "images": _IMAGE,
            "grain_intensity": _float_widget(0.04, 0.001, 1.0, 0.001),
            "saturation_mix": _float_widget(0.5, 0.0, 1.0, 0.01),
            "batch_size": ("INT", {"default": 4, "min": 0, "max": 500, "step": 1}),
        }}

    RETURN_TYPES = _IMAGE
    FUNCTION = "apply_grain"
    CATEGORY = "video/enhancement"
    DESCRIPTION = "Adds lightweight film grain"

    def apply_grain(self, images, grain_intensity, saturation_mix, batch_size):
        # `batch_size` frames share one torch.randn draw from the device's global generator, exactly like
        # the reference's chunk loop; 0 means one draw for the whole batch.
        frames = int(images.shape[0])
        step = batch_size if batch_size > 0 else max(frames, 1)
        primary = compute_device()
        many = images.ndim == 4 and frames > 0 and images.shape[-1] == 3 and not ops.oversize_chunks(frames, _frame_numel(images), step)
        if not many:
            # RNG chunks torch itself splits (> 2^29 elements), an empty batch, frames the kernels refuse: the generator is consumed inside
            # ops.film_grain, piece by piece, in submission order
            def run_now(gpu_frames, _first, out=None):
                return ops.film_grain(gpu_frames, grain_intensity, saturation_mix, chunk_frames=step, out=out)
            return (_run_grouped(images, run_now, multiple_of=step),)
        # The noise of the WHOLE batch is reserved here, when the node is called -- on the host, from the PRIMARY device's generator, the one
        # the reference's torch.randn calls would consume, chunk after chunk (nodes.py:46-51) -- and sliced per piece when the piece runs:
        # the generator moves exactly as the reference's loop moves it whether the pieces run now, later (deferred graph fusion) or on
        # another GPU (VRGDG_DEVICES).
        plans = ops.plan_noise(frames, _frame_numel(images), step, primary)

        def run(gpu_frames, first, out=None):
            return ops.film_grain(gpu_frames, grain_intensity, saturation_mix, chunk_frames=step,
                                  plans=ops.slice_plans(plans, first, int(gpu_frames.shape[0])), out=out)

        fuse = {"I": grain_intensity, "s": saturation_mix, "step": step, "plans": plans}
        return (_run_grouped(images, run, multiple_of=step, fn_for_device=lambda _device: run, kind="grain", fuse=fuse),)


class ColorMatchToReference:
    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "images": _IMAGE,
            "reference_image": _IMAGE,
            "match_strength": _float_widget(1.0, 0.0, 1.0, 0.01),
            "batch_size": ("INT", {"default": 1, "min": 1, "max": 500, "step": 1}),
        }}

    RETURN_TYPES = _IMAGE
    FUNCTION = "match_color"
    CATEGORY = "video/enhancement"
    DESCRIPTION = "Matches the color tone of input image to a reference image using LAB mean/std alignment"

    def match_color(self, images, reference_image, match_strength, batch_size):
        dev = compute_device()
        ref = on_device(reference_image, dev)            # (a result of this pack that is still in HBM is used where it is)
        n_ref = int(ref.shape[0])
        frames = int(images.shape[0])
        expand = None
        if n_ref != 1:
            # the reference broadcasts the [n_ref,3,1,1] reference statistics against every batch_size chunk (nodes.py:112):
            # a chunk of n_ref frames pairs frame i with reference i; a chunk of ONE frame broadcasts the other way and
            # yields n_ref frames (that frame matched to every reference); any other size is torch's broadcasting error
            sizes = [min(batch_size, frames - i) for i in range(0, frames, batch_size)]
            bad = next((b for b in sizes if b not in (1, n_ref)), None)
            if bad is not None:
                raise RuntimeError(f"The size of tensor a ({bad}) must match the size of tensor b ({n_ref}) at "
                                   "non-singleton dimension 0")
            if any(b == 1 for b in sizes):
                expand, f = [], 0
                for b in sizes:
                    expand += [f] * n_ref if b == 1 else list(range(f, f + b))
                    f += b
        # side stream: only the apply pass of the first group waits for it.  Host frames: the GPU idles while the first piece crosses
        # PCIe, so the reduction takes its latency form; device frames: only steps of few frames leave it the room (ops.SMALL_STEP_FRAMES)
        step_frames = frames if images.is_cuda else 0
        ref_ms, ref_ready = ops.reference_stats_async(ref, step_frames=step_frames)
        # The statistics of a chunk are ONE torch reduction call over its frames (nodes.py:109-110): the call sizes shape the
        # device statistics (ops.CM_STATS), so every piece streamed through the GPU is made of whole calls.
        if n_ref == 1:
            calls_of = lambda first, n: batch_size          # pieces start on multiples of batch_size; the last holds the remainder
            group = batch_size
        else:
            sizes = [min(batch_size, frames - i) for i in range(0, frames, batch_size)]
            calls = [c for b in sizes for c in ([1] * n_ref if b == 1 else [b])]       # a broadcast single frame: n_ref calls of one frame
            starts = [0]
            for c in calls:
                starts.append(starts[-1] + c)
            first_call = {f: i for i, f in enumerate(starts)}

            def calls_of(first, n):
                i, got, out = first_call[first], 0, []
                while got < n:
                    out.append(calls[i]); got += calls[i]; i += 1
                return out
            group = n_ref

        def run(gpu_frames, first, out=None):
            return ops.color_match(gpu_frames, None, match_strength, ref_ms=ref_ms, cm_chunk=calls_of(first, int(gpu_frames.shape[0])),
                                   ref_event=ref_ready, out=out)

        per_device = {}

        def stats_on(device):
            if device not in per_device:
                per_device[device] = ops.reference_stats_async(ref.to(device), step_frames=step_frames)   # every GPU reduces the reference frame itself: same bits
            return per_device[device]

        def run_on(device):
            if device == dev:
                return run
            ms_d, ready_d = stats_on(device)

            def run_d(gpu_frames, first, out=None):
                return ops.color_match(gpu_frames, None, match_strength, ref_ms=ms_d, cm_chunk=calls_of(first, int(gpu_frames.shape[0])),
                                       ref_event=ready_d, out=out)
            return run_d

        def stage_on(device):
            ms_d, ready_d = stats_on(device)
            return run_on(device), {"ref_ms": ms_d, "k": match_strength, "calls_of": calls_of, "ref_event": ready_d}

        fuse = None
        if expand is not None:
            images = images[torch.tensor(expand, dtype=torch.long, device=images.device)]
        elif images.ndim == 4 and images.shape[-1] == 3:
            fuse = {"ref_ms": ref_ms, "k": match_strength, "calls_of": calls_of, "ref_event": ref_ready}
        return (_run_grouped(images, run, multiple_of=group, fn_for_device=run_on, kind="colormatch" if fuse else None, fuse=fuse,
                             stage_for_device=stage_on),)


class _Sharpen:
    _OP = ""
    _MAX = 2.0
    _RGB_ONLY_ON_GPU_FLAG = True

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "images": _IMAGE,
            "strength": _float_widget(0.5, 0.0, cls._MAX, 0.01),
            "use_gpu": ("BOOLEAN", {"default": False}),
        }}

    RETURN_TYPES = _IMAGE
    CATEGORY = "video/enhancement"

    def _apply(self, images: torch.Tensor, strength: float, use_gpu: bool) -> Tuple[torch.Tensor]:
        # use_gpu selects the reference's *border / sign semantics* (False: edge-replicate numpy path,
        # True: zero-padded avg_pool2d / conv2d path); both run on the MI355X here.
        if use_gpu and self._RGB_ONLY_ON_GPU_FLAG and images.shape[-1] != 3:
            raise RuntimeError(f"Given groups=3, expected input to have 3 channels, but got {images.shape[-1]} channels instead")

        def run(gpu_frames, _first, out=None):
            return ops.stencil3x3(gpu_frames, self._OP, strength, zero_border=bool(use_gpu), out=out)

        fuse = {"op": self._OP, "strength": strength, "zero": bool(use_gpu)} if images.ndim == 4 and images.shape[-1] == 3 else None
        return (_run_grouped(images, run, fn_for_device=lambda _device: run, kind="sharpen" if fuse else None, fuse=fuse),)


class FastUnsharpSharpen(_Sharpen):
    _OP = "unsharp"
    _MAX = 10.0
    _RGB_ONLY_ON_GPU_FLAG = False
    FUNCTION = "apply_unsharp"
    DESCRIPTION = "Unsharp mask (CPU default, optional GPU path)."

    def apply_unsharp(self, images, strength, use_gpu):
        return self._apply(images, strength, use_gpu)


class FastLaplacianSharpen(_Sharpen):
    _OP = "laplacian"
    FUNCTION = "apply_laplacian"
    DESCRIPTION = "Laplacian sharpen (CPU default, optional GPU)."

    def apply_laplacian(self, images, strength, use_gpu):
        return self._apply(images, strength, use_gpu)


class FastSobelSharpen(_Sharpen):
    _OP = "sobel"
    FUNCTION = "apply_sobel"
    DESCRIPTION = "Sobel sharpen (CPU default, optional GPU)."

    def apply_sobel(self, images, strength, use_gpu):
        return self._apply(images, strength, use_gpu)


NODE_CLASS_MAPPINGS = {
    "FastFilmGrain": FastFilmGrain,
    "ColorMatchToReference": ColorMatchToReference,
    "FastUnsharpSharpen": FastUnsharpSharpen,
    "FastLaplacianSharpen": FastLaplacianSharpen,
    "FastSobelSharpen": FastSobelSharpen,
}

NODE_DISPLAY_NAME_MAPPINGS = {
    "FastFilmGrain": "\U0001F39E\uFE0F Fast Film Grain",
    "ColorMatchToReference": "\U0001F3A8 Color Match To Reference",
    "FastUnsharpSharpen": "\U0001F3AF Fast Unsharp Sharpen",
    "FastLaplacianSharpen": "\U0001F300 Fast Laplacian Sharpen",
    "FastSobelSharpen": "\U0001F4CF Fast Sobel Sharpen",
}
