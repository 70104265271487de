"""Host-side bookkeeping of torch's HIP Philox generator, so that the in-register noise of the grain
kernels is the very stream ``torch.randn`` would have produced on this device with the same generator
state (the reference draws its grain with ``torch.randn_like`` / ``torch.randn(generator=...)``:
nodes.py:51, VRGDG_LUTVideoTools.py:267-271, VRGDG_StandaloneVideoEnhancerNodes.py:268-271).

Mirrors ``calc_execution_policy`` and ``distribution_nullary_kernel``
(torch/include/ATen/native/hip/DistributionTemplates.h:52-66, 118-140): one ``randn(numel)`` call uses
G = min(CUs * (max_threads_per_CU / 256), ceil(numel / 256)) * 256 Philox subsequences and advances the
generator offset by ((numel - 1) // (4 G) + 1) * 4.
"""
from __future__ import annotations

import threading
from dataclasses import dataclass

import torch

#: torch takes the generator's mutex around "read the Philox state, advance the offset" of every randn; the reservations
#: below do the same read-modify-write from Python (nodes and routes may enter from different host threads)
_RESERVE_LOCK = threading.Lock()

BLOCK = 256
UNROLL = 4
#: ATen runs a randn whose output cannot be indexed with 32-bit BYTE offsets (TensorIteratorBase::can_use_32bit_indexing:
#: 1 + (numel - 1) * 4 <= INT32_MAX, i.e. numel <= 2^29) as several kernels over sub-ranges (`split_32bit`); the in-register
#: noise of the fused kernels covers chunks up to this size, larger chunks go through `reserve_split`.
MAX_CHUNK_NUMEL = 2 ** 29


@dataclass(frozen=True)
class DeviceGeometry:
    cu_count: int
    max_threads_per_cu: int

    @property
    def max_grid(self) -> int:
        return self.cu_count * (self.max_threads_per_cu // BLOCK)


_geom_cache = {}


def device_geometry(device=None) -> DeviceGeometry:
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    g = _geom_cache.get(idx)
    if g is None:
        props = torch.cuda.get_device_properties(idx)
        g = DeviceGeometry(int(props.multi_processor_count), int(props.max_threads_per_multi_processor))
        _geom_cache[idx] = g
    return g


def grid_threads(numel: int, geom: DeviceGeometry) -> int:
    grid = (numel + BLOCK - 1) // BLOCK
    return min(geom.max_grid, grid) * BLOCK


def counter_offset(numel: int, G: int) -> int:
    return ((numel - 1) // (G * UNROLL) + 1) * 4


@dataclass
class ChunkedStream:
    """Noise description for `n_chunks` equal chunks (+ how far the generator moved)."""
    seed: int
    offset0: int
    offset_stride: int
    seed_stride: int
    grid_threads: int
    chunk_numel: int


def split_32bit(numel: int, start: int = 0):
    """The sub-ranges ``TensorIteratorBase::with_32bit_indexing`` (SplitUntil32Bit, TensorIterator.cpp) cuts a contiguous
    fp32 tensor of `numel` elements into, in the order they are launched: the (single, coalesced) dimension is halved --
    first half floor(n/2), second half the rest -- depth first, until a piece can be indexed with 32-bit byte offsets.
    Returns [(start, size), ...] in memory order."""
    if numel <= MAX_CHUNK_NUMEL:
        return [(start, numel)]
    first = numel // 2
    return split_32bit(first, start) + split_32bit(numel - first, start + first)


@dataclass
class SplitStream:
    """One oversize ``randn(numel)`` call: the leaves ATen launches, each with its own grid and generator offset."""
    seed: int
    leaves: list            # [(start, size, grid_threads, offset), ...]
    chunk_numel: int


def reserve_split(numel: int, device, generator=None, geom: DeviceGeometry | None = None) -> SplitStream:
    """Generator bookkeeping of ONE ``torch.randn(numel)`` with numel > 2^29, exactly as ATen's
    ``distribution_nullary_kernel`` does it (DistributionTemplates.h:118-140): the outer call takes its Philox state --
    advancing the offset by the whole tensor's counter_offset, which is then never used -- before it notices that the
    iterator needs splitting, and every sub-iterator call takes (and advances by) its own."""
    geom = geom or device_geometry(device)
    gen = _generator_for(device, generator)
    with _RESERVE_LOCK:
        seed = int(gen.initial_seed())
        off = int(gen.get_offset())
        off += counter_offset(numel, grid_threads(numel, geom))          # the outer call's unused reservation
        leaves = []
        for start, size in split_32bit(numel):
            G = grid_threads(size, geom)
            leaves.append((start, size, G, off))
            off += counter_offset(size, G)
        gen.set_offset(off)
    return SplitStream(seed=seed & 0xFFFFFFFFFFFFFFFF, leaves=leaves, chunk_numel=numel)


def _generator_for(device, generator):
    if generator is not None:
        return generator
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return torch.cuda.default_generators[idx]


def reserve(numel: int, n_chunks: int, device, generator=None, geom: DeviceGeometry | None = None) -> ChunkedStream:
    """Reserve the generator range that `n_chunks` successive ``torch.randn(numel)`` calls would consume,
    advance the generator exactly as those calls would, and return the stream description."""
    if numel <= 0 or n_chunks <= 0:
        raise ValueError("numel and n_chunks must be positive")
    if numel > MAX_CHUNK_NUMEL:
        raise ValueError(f"a noise chunk of {numel} elements is split by torch into 32-bit indexable sub-ranges: use reserve_split")
    geom = geom or device_geometry(device)
    gen = _generator_for(device, generator)
    G = grid_threads(numel, geom)
    stride = counter_offset(numel, G)
    with _RESERVE_LOCK:
        seed = int(gen.initial_seed())
        off = int(gen.get_offset())
        gen.set_offset(off + stride * n_chunks)
    return ChunkedStream(seed=seed & 0xFFFFFFFFFFFFFFFF, offset0=off, offset_stride=stride, seed_stride=0,
                         grid_threads=G, chunk_numel=numel)


def per_frame_seeded(frame_numel: int, base_seed: int, device, geom: DeviceGeometry | None = None) -> ChunkedStream:
    """One generator per frame seeded base_seed + frame (offset 0): the stream of
    ``torch.Generator(device).manual_seed(seed + frame_start + offset)`` + ``torch.randn(frame.shape)``
    (VRGDG_StandaloneVideoEnhancerNodes.py:268-271).  The & 0x7FFFFFFF wrap of the reference is applied by
    the caller when it can occur inside a call."""
    if frame_numel > MAX_CHUNK_NUMEL:
        raise NotImplementedError("frame too large for the in-register noise path")
    geom = geom or device_geometry(device)
    G = grid_threads(frame_numel, geom)
    return ChunkedStream(seed=int(base_seed) & 0xFFFFFFFFFFFFFFFF, offset0=0, offset_stride=0, seed_stride=1,
                         grid_threads=G, chunk_numel=frame_numel)
