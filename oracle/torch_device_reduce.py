"""TEST INFRASTRUCTURE (oracle) -- torch-ROCm's `mean(dim=[2,3])` / `std(dim=[2,3])` of a contiguous fp32 `[b,3,H,W]` tensor on the
MI355X, restated in numpy so that the result is the same BITS the device returns.

The reference takes its colour statistics with those two calls (/root/reference/nodes.py:99-100,109-110).  On a GPU they are ATen's
`reduce_kernel` (torch/include/ATen/native/cuda/Reduce.cuh, the copy that ships with the installed torch 2.10.0+rocm7.0) with
`MeanOps<float,float,float,float>` (vt0 = 4, vectorised by 4) and `WelfordOps<float,float,int32,pair>` (vt0 = 2, vectorised by 2)
from ATen/native/SharedReduceOps.h.  fp32 addition is not associative, so the value depends on the launch geometry
(`setReduceConfig`, Reduce.cuh:1012-1180: a function of the number of outputs, the reduction length and the device: 256 CUs,
warp 64) and on the order in which per-thread accumulators, warp lanes and warps are combined.  This file follows that code path
for the shapes the path produces (TensorIterator view: 2 dims -- H*W contiguous reduced, b*3 kept):

  * per output ONE workgroup row of `block_width` lanes (x `block_height` rows when the reduction is split across warps);
    `ctas_per_output` is always 1 here (Reduce.cuh:1118-1135: iter.ndim() == 2 caps max_threads_per_mp at 256 on ROCm, so
    blocks_per_sm = 256 / 512 = 0 and the grid is never split) -- asserted below;
  * thread loop: `input_vectorized_thread_reduce_impl` (Reduce.cuh:498-556; unaligned head, vec-wide accumulators, tail) or
    `thread_reduce_impl` (:558-624; vt0 strided accumulators) below 128 elements;
  * `block_x_reduce` (:626-661; LDS tree down to 64 lanes, then shuffles with INCREASING offsets -- the USE_ROCM branch),
    `block_y_reduce` (:663-680);
  * `project`: mean = sum * float(num_outputs)/numel ; std = sqrt(m2 / (nf - 1)).

Compiler facts of libtorch_hip.so (hipcc, default -ffp-contract=fast-honor-pragmas, IEEE division / sqrt): checked against the
device by tools/probe_torch_reduce.py (ground truth collected on the MI355X: tests/golden/torch_reduce_truth.npz).
"""
from __future__ import annotations

import numpy as np

NUM_MP = 256          # multiProcessorCount of the MI355X
WARP = 64
MAX_THREADS = 512     # mnt_wrapper<float>::MAX_NUM_THREADS

f32 = np.float32


def _last_pow2(n: int) -> int:
    n |= n >> 1; n |= n >> 2; n |= n >> 4; n |= n >> 8; n |= n >> 16
    return max(1, n - (n >> 1))


def _div_up(a: int, b: int) -> int:
    return (a + b - 1) // b


class ReduceConfig:
    """setReduceConfig (Reduce.cuh:1012-1180) for a 2-dim iterator reducing its contiguous fastest dimension."""

    def __init__(self, num_outputs: int, num_inputs: int, vec: int):
        self.num_outputs, self.num_inputs, self.vec = num_outputs, num_inputs, vec
        dim0, dim1 = num_inputs, num_outputs
        self.vectorize = dim0 >= 128                     # reduction on the fastest dim, stride == sizeof(float), one reduce dim
        if self.vectorize:
            dim0 //= vec
        d0 = _last_pow2(dim0) if dim0 < MAX_THREADS else MAX_THREADS
        d1 = _last_pow2(dim1) if dim1 < MAX_THREADS else MAX_THREADS
        bw = min(d0, WARP)
        bh = min(d1, MAX_THREADS // bw)
        bw = min(d0, MAX_THREADS // bh)
        self.block_width, self.block_height = bw, bh
        self.step_input, self.step_output = bw, 1        # input_mult[0] = split_input(block_width)
        vpt = _div_up(num_inputs, self.step_input)
        self.split_warps = vpt >= min(bh * 16, 256)      # (force_splitting_output needs num_mp < 100)
        if self.split_warps:
            self.step_input *= bh                        # input_mult[1] = block_width
        else:
            self.step_output *= bh                       # output_mult[1] = 1
        grid_x = _div_up(num_outputs, self.step_output)
        # `grid.x == grid.y == grid.z == 1` as C parses it: ((x == y) == z) == 1 with y = z = 1
        single = (int(int(grid_x == 1) == 1) == 1)
        max_threads_per_mp = 2048 if single else 256     # iter.ndim() == 2
        target = NUM_MP * (max_threads_per_mp // (bw * bh))
        vpt = _div_up(num_inputs, self.step_input)
        self.ctas_per_output = 1
        if self.split_warps and vpt >= 256 and grid_x <= target:
            c1, c2, c3 = _div_up(target, grid_x), _div_up(vpt, 16), _div_up(vpt, 256)
            c = max(min(c1, c2), c3)
            if c > NUM_MP:
                c = NUM_MP
            elif c > _div_up(NUM_MP, 2):
                c = _div_up(NUM_MP, 2)
            elif c < 16:
                c = 1
            self.ctas_per_output = c
        if self.ctas_per_output != 1:
            raise NotImplementedError("global (multi-CTA) reduction: not a shape this path produces")

    def __repr__(self):
        return (f"ReduceConfig(outputs={self.num_outputs}, inputs={self.num_inputs}, vec={self.vec}, block=({self.block_width},"
                f"{self.block_height}), split_warps={self.split_warps}, vectorize={self.vectorize})")


def fma32(a, b, c):
    """RN32(a*b + c) exactly, for float32 arrays: the product is exact in float64; the float64 sum is turned into a
    round-to-odd value with the error of a two-sum, which then rounds correctly to 24 bits."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64); c = np.asarray(c, dtype=np.float64)
    p = a * b
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    bits = s.view(np.int64).copy() if s.ndim else np.array(s).view(np.int64).copy()
    fix = (err != 0) & ((bits & 1) == 0) & np.isfinite(s)
    up = (err > 0) == (s > 0)          # away from zero when the error has the sign of s
    bits = np.where(fix, np.where(up, bits + 1, bits - 1), bits)
    return bits.view(np.float64).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# ops (ATen/native/SharedReduceOps.h:92-141 WelfordOps, :143-172 MeanOps), on arrays
# ---------------------------------------------------------------------------------------------------------------------
class MeanOps:
    def __init__(self, factor):
        self.factor = f32(factor)

    def ident(self, shape):
        return np.zeros(shape, dtype=f32)

    def reduce(self, acc, x, mask=None, fused=True):
        new = acc + x
        return new if mask is None else np.where(mask, new, acc)

    def combine(self, a, b):
        return a + b

    def take(self, acc, index):
        return acc[index]

    def put(self, acc, index, val):
        acc[index] = val

    def project(self, acc):
        return acc * self.factor


class WelfordOps:
    """acc = (mean, m2, n, nf) stacked on the last axis as float32 (n is exact below 2^24 and only used through nf)."""

    def __init__(self, correction=1.0, contract=True):
        self.correction = f32(correction)
        self.contract = contract

    def ident(self, shape):
        return np.zeros(tuple(shape) + (3,), dtype=f32)          # mean, m2, nf

    def reduce(self, acc, x, mask=None, fused=True):
        mean, m2, nf = acc[..., 0], acc[..., 1], acc[..., 2]
        new_nf = nf + f32(1)                                      # float(n + 1): exact for n < 2^24
        delta = x - mean
        new_mean = mean + delta / new_nf
        new_delta = x - new_mean
        new_m2 = fma32(delta, new_delta, m2) if (self.contract and fused) else m2 + delta * new_delta
        new = np.stack([new_mean, new_m2, new_nf], axis=-1)
        return new if mask is None else np.where(mask[..., None], new, acc)

    def combine(self, a, b):
        am, a2, an = a[..., 0], a[..., 1], a[..., 2]
        bm, b2, bn = b[..., 0], b[..., 1], b[..., 2]
        delta = bm - am
        cnt = an + bn
        with np.errstate(invalid="ignore", divide="ignore"):
            nb = bn / cnt
        if self.contract:
            mean = fma32(delta, nb, am)
            m2 = fma32((delta * delta) * an, nb, a2 + b2)
        else:
            mean = am + delta * nb
            m2 = (a2 + b2) + ((delta * delta) * an) * nb
        out = np.stack([mean, m2, cnt], axis=-1)
        out = np.where((bn == 0)[..., None], a, out)
        out = np.where((an == 0)[..., None], b, out)
        return out

    def take(self, acc, index):
        return acc[index]

    def put(self, acc, index, val):
        acc[index] = val

    def project(self, acc):
        m2, nf = acc[..., 1], acc[..., 2]
        divisor = np.where(nf > self.correction, nf - self.correction, f32(0))
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.sqrt(m2 / divisor)


# ---------------------------------------------------------------------------------------------------------------------
# the kernel, for all outputs at once: planes[o, :] is output o's contiguous reduction range; shifts[o] = elements by which
# its first element is past a vec-aligned address
# ---------------------------------------------------------------------------------------------------------------------
def _thread_reduce(planes, shifts, cfg: ReduceConfig, ops):
    O, n = planes.shape
    bw, bh, vec = cfg.block_width, cfg.block_height, cfg.vec
    rows = bh if cfg.split_warps else 1
    T = bw * rows                                      # threads cooperating on one output; linear id t = tx + ty * bw
    stride = cfg.step_input
    assert stride == T
    t = np.arange(T)
    tx, ty = t % bw, t // bw
    tail_ok = (ty == 0)                                # should_reduce_tail: threadIdx.y == 0 when the block reduces over y
    if not cfg.vectorize:
        vt0 = vec                                      # vt0 == input_vec_size for both ops
        acc = [ops.ident((O, T)) for _ in range(vt0)]
        idx = t.copy()
        while True:                                    # all threads share the trip count test per thread: mask instead
            full = idx + (vt0 - 1) * stride < n
            if not full.any():
                break
            for i in range(vt0):
                pos = np.minimum(idx + i * stride, n - 1)
                acc[i] = ops.reduce(acc[i], planes[:, pos], np.broadcast_to(full, (O, T)))
            idx = np.where(full, idx + stride * vt0, idx)
        for i in range(vt0):
            ok = idx < n
            pos = np.minimum(idx, n - 1)
            acc[i] = ops.reduce(acc[i], planes[:, pos], np.broadcast_to(ok, (O, T)))
            idx = np.where(ok, idx + stride, idx)
        out = acc[0]
        for i in range(1, vt0):
            out = ops.combine(out, acc[i])
        return out

    # vectorised path; outputs are grouped by their head shift (same control flow within a group)
    result = ops.ident((O, T))
    for sh in np.unique(shifts):
        sel = np.nonzero(shifts == sh)[0]
        P = planes[sel]
        Og = len(sel)
        value = ops.ident((Og, T))
        start, end, shift = 0, n, int(sh)
        if shift > 0:
            # data -= shift; end += shift; threads shift <= tx < vec (and tail_ok) reduce element tx - shift
            m = (tx >= shift) & (tx < vec) & tail_ok
            pos = np.clip(tx - shift, 0, n - 1)
            value = ops.reduce(value, P[:, pos], np.broadcast_to(m, (Og, T)))
            start = vec - shift                        # first aligned element of the plane
            end = n + shift - vec
            shift = vec - shift
        acc = [value] + [ops.ident((Og, T)) for _ in range(vec - 1)]
        body = P[:, start:start + max(end, 0)]
        nvec = max(end, 0) // vec                      # idx * vec + vec - 1 < end
        K = _div_up(nvec, T) if nvec else 0
        for k in range(K):
            idx = t + k * T
            ok = idx < nvec
            pos = np.minimum(idx, max(nvec - 1, 0)) * vec
            for i in range(vec):
                # libtorch_hip.so's main loop (disassembled): the SLP vectoriser paired accumulator 0's `m2 + delta * new_delta` with
                # accumulator 1's `mean + delta / n` into one v_pk_add_f32, so accumulator 0 rounds the product (v_mul_f32) before
                # the add while accumulator 1 keeps the contracted v_fmac_f32; head, tail and the scalar path contract everywhere
                acc[i] = ops.reduce(acc[i], body[:, pos + i], None if ok.all() else np.broadcast_to(ok, (Og, T)), fused=(i != 0))
        if end > 0:
            tail_start = end - end % vec
            tidx = tail_start + tx
            m = tail_ok & (tidx < end)
            if m.any():
                pos = np.minimum(tidx, end - 1)
                acc[0] = ops.reduce(acc[0], body[:, pos], np.broadcast_to(m, (Og, T)))
        out = acc[0]
        for i in range(1, vec):
            out = ops.combine(out, acc[i])
        result[sel] = out
    return result


def _block_reduce(val, cfg: ReduceConfig, ops):
    """val[O, T(, ...)] -> [O(, ...)]: block_x_reduce then block_y_reduce; the value of thread (0, 0)."""
    bw, bh = cfg.block_width, cfg.block_height
    rows = bh if cfg.split_warps else 1
    O = val.shape[0]
    v = val.reshape((O, rows, bw) + val.shape[2:]).copy()
    dim_x = bw
    if dim_x > WARP:
        off = dim_x // 2
        while off >= WARP:
            v[:, :, :off] = ops.combine(v[:, :, :off], v[:, :, off:2 * off])
            off >>= 1
        dim_x = WARP
    off = 1
    while off < dim_x:                                 # USE_ROCM: increasing offsets, shfl_down
        lim = dim_x - off
        v[:, :, :lim] = ops.combine(v[:, :, :lim], v[:, :, off:off + lim])
        off <<= 1
    col = v[:, :, 0].copy()                            # [O, rows, ...]
    if cfg.split_warps:
        off = bh // 2
        while off > 0:
            col[:, :off] = ops.combine(col[:, :off], col[:, off:2 * off])
            off >>= 1
    return col[:, 0]


def reduce_planes(planes, shifts, vec, ops):
    planes = np.ascontiguousarray(planes, dtype=f32)
    O, n = planes.shape
    cfg = ReduceConfig(O, n, vec)
    if not cfg.vectorize:
        shifts = np.zeros(O, dtype=np.int64)
    val = _thread_reduce(planes, np.asarray(shifts), cfg, ops)
    return ops.project(_block_reduce(val, cfg, ops)), cfg


def mean_std(x, base_offset_elems: int = 0, contract: bool = True):
    """x: float32 [b, 3, H, W] (contiguous NCHW, as kornia's rgb_to_lab returns it).  Returns (mean, std) as float32 [b, 3]
    with the bits of `x.mean(dim=[2,3])` / `x.std(dim=[2,3])` evaluated by torch on the MI355X.  `base_offset_elems`: offset
    of x's first element from a 16-byte aligned address, in elements (0 for a fresh tensor)."""
    x = np.ascontiguousarray(x, dtype=f32)
    b, c, H, W = x.shape
    n = H * W
    O = b * c
    planes = x.reshape(O, n)
    start = base_offset_elems + np.arange(O, dtype=np.int64) * n
    factor = f32(O) / f32(np.int64(O) * np.int64(n))            # static_cast<float>(num_output_elements) / numel (int64 -> float)
    mean, _ = reduce_planes(planes, start % 4, 4, MeanOps(factor))
    std, _ = reduce_planes(planes, start % 2, 2, WelfordOps(1.0, contract))
    return mean.reshape(b, c), std.reshape(b, c)
