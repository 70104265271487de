"""Philox4x32-10 and the torch-HIP ``randn`` element mapping (TEST INFRASTRUCTURE ONLY).

The reference draws its grain from ``torch.randn_like`` / ``torch.randn(generator=...)``
(nodes.py:51; VRGDG_LUTVideoTools.py:271; VRGDG_StandaloneVideoEnhancerNodes.py:271).  On a
ROCm device that is ATen's ``distribution_elementwise_grid_stride_kernel``
(torch/include/ATen/native/hip/DistributionTemplates.h:52-99, 446-456) driving rocRAND's
Philox4x32-10 (rocrand/rocrand_philox4x32_10.h:150-310) and ``rocrand_normal4``
(rocrand/rocrand_normal.h:52-68, 259-265).  This module restates:

  * the integer part bit-exactly (counter/key schedule, ten rounds, element -> (thread, call,
    component) mapping, generator-offset bookkeeping incl. the >INT32 TensorIterator split);
  * the Box-Muller part in float64 (the device uses the hardware ``v_log_f32`` / ``v_sin_f32`` /
    ``v_cos_f32`` approximations, which cannot be reproduced on a CPU; the bit-exact check of
    the normals is therefore ``torch.randn(device="cuda")`` itself, in the ``-m gpu`` tests).
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

TORCH_BLOCK = 256            # block_size_bound (DistributionTemplates.h)
TORCH_UNROLL = 4             # float4 per hiprand_normal4 call
INT32_MAX = 2 ** 31 - 1


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten Philox rounds on uint32 numpy arrays (rocrand_philox4x32_10.h:270-303)."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    c3 = np.asarray(c3, dtype=np.uint64)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def torch_grid_threads(numel: int, cu_count: int, max_threads_per_cu: int = 2048) -> int:
    """G = grid.x * 256 of calc_execution_policy (DistributionTemplates.h:52-66)."""
    grid = (numel + TORCH_BLOCK - 1) // TORCH_BLOCK
    grid = min(cu_count * (max_threads_per_cu // TORCH_BLOCK), grid)
    return grid * TORCH_BLOCK


def torch_counter_offset(numel: int, G: int) -> int:
    """How far one randn call advances the generator's philox offset (same function)."""
    return ((numel - 1) // (G * TORCH_UNROLL) + 1) * 4


def split_32bit(numel: int, itemsize: int = 4):
    """Sub-ranges TensorIterator::with_32bit_indexing() yields for a contiguous 1-D output:
    halve (first = n//2) until ``numel <= INT32_MAX`` and the byte offset of the last element
    fits int32; in-order traversal (TensorIterator.cpp SplitUntil32Bit)."""
    def ok(n):
        return n <= INT32_MAX and (n - 1) * itemsize <= INT32_MAX
    out = []

    def rec(start, n):
        if ok(n):
            out.append((start, n))
            return
        first = n // 2
        rec(start, first)
        rec(start + first, n - first)
    rec(0, numel)
    return out


def torch_randn_plan(numel: int, philox_offset: int, cu_count: int, itemsize: int = 4):
    """Launch plan of one ``torch.randn(numel)`` on a HIP device.

    Returns (list of (start, length, G, offset_for_this_launch), new_generator_offset).  Every
    call to ``distribution_nullary_kernel`` -- including the outer one that only splits --
    consumes its own counter_offset (DistributionTemplates.h:118-140)."""
    if numel == 0:
        return [], philox_offset
    launches = []

    def rec(start, n, off):
        G = torch_grid_threads(n, cu_count)
        my_off = off
        off = off + torch_counter_offset(n, G)
        ok = n <= INT32_MAX and (n - 1) * itemsize <= INT32_MAX
        if ok:
            launches.append((start, n, G, my_off))
            return off
        first = n // 2
        # with_32bit_indexing() splits recursively *before* launching; each yielded sub-iter
        # then calls distribution_nullary_kernel itself.
        for (s, l) in split_32bit(n, itemsize):
            Gs = torch_grid_threads(l, cu_count)
            launches.append((start + s, l, Gs, off))
            off = off + torch_counter_offset(l, Gs)
        return off

    new_off = rec(0, numel, philox_offset)
    return launches, new_off


def torch_stream_uint32(numel: int, seed: int, philox_offset: int, G: int):
    """The raw uint32 each element li consumes *as a pair source*: returns arrays (a, b) such
    that element li's normal is box_muller(a, b) component (ii & 1), where
    idx = li % G, q = li // G, call k = q // 4, ii = q % 4; the call's four words are
    (x, y, z, w); ii in {0,1} uses (x, y), ii in {2,3} uses (z, w); even ii -> sin, odd -> cos."""
    li = np.arange(numel, dtype=np.uint64)
    idx = li % np.uint64(G)
    q = li // np.uint64(G)
    k = q // np.uint64(4)
    ii = (q % np.uint64(4)).astype(np.int64)
    ctr = np.uint64(philox_offset // 4) + k
    x, y, z, w = philox4x32_10(ctr & MASK32, ctr >> np.uint64(32), idx & MASK32, idx >> np.uint64(32),
                               seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = np.where(ii < 2, x, z)
    b = np.where(ii < 2, y, w)
    return a, b, ii


def box_muller_f64(a, b, ii):
    """float64 evaluation of rocrand's box_muller (rocrand_normal.h:52-68) for component ii&1."""
    two_m32 = np.float32(2.3283064e-10)
    two_pi_m32 = np.float32(1.46291807e-09)
    u = np.float64(two_m32) + a.astype(np.float32).astype(np.float64) * np.float64(two_m32)
    v = np.float64(two_pi_m32) + b.astype(np.float32).astype(np.float64) * np.float64(two_pi_m32)
    s = np.sqrt(-2.0 * np.log(u.astype(np.float32).astype(np.float64)))
    v32 = v.astype(np.float32).astype(np.float64)
    return np.where((ii & 1) == 0, np.sin(v32) * s, np.cos(v32) * s)


def torch_stream_normals_f64(numel: int, seed: int, philox_offset: int, G: int):
    a, b, ii = torch_stream_uint32(numel, seed, philox_offset, G)
    return box_muller_f64(a, b, ii)
