// vrg_torch_stats.hip -- per-frame Lab mean / std with the BITS torch-ROCm's reductions return on the MI355X.
//
// The reference takes its colour statistics with `lab.mean(dim=[2,3], keepdim=True)` and `lab.std(dim=[2,3], keepdim=True)` on a
// contiguous [b,3,H,W] fp32 tensor, once per batch_size chunk (/root/reference nodes.py:99-100,109-110).  On this GPU those are
// ATen's reduce_kernel<512,1,ReduceOp<float, MeanOps<float,float,float,float>, uint32, float, 4, 4>> and
// <..., WelfordOps<float,float,int,pair<float,float>>, ..., 2, 2> (torch/include/ATen/native/cuda/Reduce.cuh and
// ATen/native/SharedReduceOps.h of the installed torch 2.10.0+rocm7.0).  fp32 sums depend on their order, so "the reference's
// statistics" are a function of the launch geometry torch picks (setReduceConfig, Reduce.cuh:1012-1180) and of how it combines
// per-thread accumulators, lanes and warps -- and, for the Welford update, of which multiply-adds hipcc contracted when it built
// libtorch_hip.so.  The kernels below replay exactly that computation on our interleaved [F][H*W][3] Lab image:
//
//   * geometry (ts_config): per output one row of `bw` lanes, x `bh` rows when the reduction is split across warps; for a 2-dim
//     iterator (H*W contiguous reduced, b*3 kept) ROCm caps max_threads_per_mp at 256, so blocks_per_sm = 256 / 512 = 0 and the
//     input is never split across workgroups: one workgroup per output, b = 1 -> (256,2), b = 2 -> (128,4), b >= 3 -> (64,8) on
//     video-sized frames (confirmed by rocprofv3 grid / workgroup sizes);
//   * thread loop: input_vectorized_thread_reduce_impl (Reduce.cuh:498-556: unaligned head, `vec` accumulators over aligned vectors,
//     tail) from 128 elements up, thread_reduce_impl (:558-624) below;
//   * block_x_reduce (:626-661: LDS tree down to 64 lanes, then shfl_down with INCREASING offsets, the USE_ROCM branch),
//     block_y_reduce (:663-680), project;
//   * contraction, read from the disassembly of libtorch_hip.so's gfx950 code object: every `a + b * c` of WelfordOps is one FMA
//     EXCEPT accumulator 0's `m2 + delta * new_delta` inside the vectorised main loop -- the SLP vectoriser paired that add with
//     accumulator 1's `mean + delta / n` into a v_pk_add_f32, so its product is rounded first (v_mul_f32).  Division and sqrt
//     are IEEE (hipcc's default), as here.
//
// oracle/torch_device_reduce.py restates the same algorithm in numpy; both are held bit-equal to torch on the device
// (tests/golden/torch_reduce_truth.npz collected on the MI355X; tests/test_gpu_parity.py compares with torch directly).
#include "vrg_common.hpp"

namespace vrg {

struct TsCfg { int bw, bh, split, vectorize; };

static int ts_last_pow2(int n) {
    n |= n >> 1; n |= n >> 2; n |= n >> 4; n |= n >> 8; n |= n >> 16;
    const int r = n - (n >> 1);
    return r > 1 ? r : 1;
}
static int64_t ts_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// setReduceConfig for iter.ndim() == 2, reduction over the contiguous fastest dimension, warp 64, 512 threads max.
// false: a geometry with ctas_per_output > 1 (not reachable for >= 2 outputs; kept as a guard).  `num_mp` = the device's CU count
// (256 on the MI355X): it enters only through target_grid_size = num_mp * (max_threads_per_mp / block threads), and ROCm's cap of
// max_threads_per_mp = 256 for 2-dim iterators makes that 0 for the 512-thread blocks of every call with more than one output --
// the geometry of the calls this file replays is therefore the same on any CU count (a compute-partitioned MI355X, another gfx950 SKU).
static bool ts_config(int64_t num_outputs, int64_t n, int vec, TsCfg& c, int num_mp = 256) {
    int64_t dim0 = n;
    c.vectorize = dim0 >= 128;
    if (c.vectorize) dim0 /= vec;
    const int d0 = dim0 < 512 ? ts_last_pow2((int)dim0) : 512;
    const int d1 = num_outputs < 512 ? ts_last_pow2((int)num_outputs) : 512;
    int bw = d0 < 64 ? d0 : 64;
    const int bh = d1 < 512 / bw ? d1 : 512 / bw;
    bw = d0 < 512 / bh ? d0 : 512 / bh;
    int64_t vpt = ts_div_up(n, bw);
    const int thr = bh * 16 < 256 ? bh * 16 : 256;
    c.split = vpt >= thr;
    c.bw = bw; c.bh = bh;
    const int64_t step_in = (int64_t)bw * (c.split ? bh : 1), step_out = c.split ? 1 : bh;
    const int64_t grid_x = ts_div_up(num_outputs, step_out);
    const int max_tpm = grid_x == 1 ? 2048 : 256;          // `grid.x == grid.y == grid.z == 1` as C evaluates it
    const int64_t target = (int64_t)num_mp * (int64_t)(max_tpm / (bw * bh));
    vpt = ts_div_up(n, step_in);
    if (c.split && vpt >= 256 && grid_x <= target) {
        const int64_t c1 = ts_div_up(target, grid_x), c2 = ts_div_up(vpt, 16), c3 = ts_div_up(vpt, 256);
        int64_t ctas = (c1 < c2 ? c1 : c2) > c3 ? (c1 < c2 ? c1 : c2) : c3;
        if (ctas > 256) ctas = 256; else if (ctas > 128) ctas = 128; else if (ctas < 16) ctas = 1;
        if (ctas != 1) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// ops
// ---------------------------------------------------------------------------------------------------------------------
struct Welf { float mean, m2; int n; float nf; };

struct MeanOp {
    typedef float acc_t;
    static constexpr int VEC = 4;
    static __device__ __forceinline__ acc_t ident() { return 0.0f; }
    template <bool FUSED> static __device__ __forceinline__ acc_t reduce(acc_t a, float x) { return a + x; }
    static __device__ __forceinline__ acc_t combine(acc_t a, acc_t b) { return a + b; }
    static __device__ __forceinline__ acc_t shfl_down(acc_t a, int off) { return __shfl_down(a, off, 64); }
};

struct WelfOp {
    typedef Welf acc_t;
    static constexpr int VEC = 2;
    static __device__ __forceinline__ acc_t ident() { return Welf{0.0f, 0.0f, 0, 0.0f}; }
    // WelfordOps::reduce (SharedReduceOps.h:100-113)
    template <bool FUSED> static __device__ __forceinline__ acc_t reduce(acc_t a, float x) {
        const int n1 = a.n + 1;
        const float nf1 = (float)n1;
        const float delta = x - a.mean;
        const float mean1 = a.mean + delta / nf1;
        const float d2 = x - mean1;
        const float m2 = FUSED ? __builtin_fmaf(delta, d2, a.m2) : a.m2 + delta * d2;
        return Welf{mean1, m2, n1, nf1};
    }
    // WelfordOps::combine (:114-131); both multiply-adds are FMAs in libtorch_hip.so at every call site
    static __device__ __forceinline__ acc_t combine(acc_t a, acc_t b) {
        if (a.nf == 0.0f) return b;
        if (b.nf == 0.0f) return a;
        const float delta = b.mean - a.mean;
        const float cnt = a.nf + b.nf;
        const float nb = b.nf / cnt;
        const float mean = __builtin_fmaf(delta, nb, a.mean);
        const float m2 = __builtin_fmaf((delta * delta) * a.nf, nb, a.m2 + b.m2);
        return Welf{mean, m2, -1, cnt};
    }
    static __device__ __forceinline__ acc_t shfl_down(acc_t a, int off) {
        return Welf{__shfl_down(a.mean, off, 64), __shfl_down(a.m2, off, 64), __shfl_down(a.n, off, 64), __shfl_down(a.nf, off, 64)};
    }
};

// WelfordOps::project with correction 1, take_sqrt
static __device__ __forceinline__ float welf_std(const Welf& a) {
    const float divisor = a.nf > 1.0f ? a.nf - 1.0f : 0.0f;
    return __builtin_sqrtf(a.m2 / divisor);
}

// block_x_reduce, then block_y_reduce when the rows share one output.  Every thread of the workgroup calls it (`live`: the thread is
// one of the bw * rows threads of the geometry); the result is valid in thread 0.  `lds` holds blockDim.x accumulators.
template <class OP>
static __device__ typename OP::acc_t ts_block_reduce(typename OP::acc_t v, int bw, int bh, bool split, int t, bool live,
                                                     typename OP::acc_t* lds) {
    const int tx = t % bw, ty = t / bw;
    int dim_x = bw;
    __syncthreads();                                   // lds may still be read by a previous call
    if (dim_x > 64) {
        if (live) lds[t] = v;
        for (int off = dim_x / 2; off >= 64; off >>= 1) {
            __syncthreads();
            if (live && tx < off && tx + off < bw) {
                v = OP::combine(v, lds[t + off]);
                lds[t] = v;
            }
        }
        dim_x = 64;
    }
    __syncthreads();
    for (int off = 1; off < dim_x; off <<= 1) {
        const typename OP::acc_t other = OP::shfl_down(v, off);
        v = OP::combine(v, other);
    }
    if (split) {
        if (live) lds[t] = v;
        for (int off = bh / 2; off > 0; off >>= 1) {
            __syncthreads();
            if (live && ty < off && ty + off < bh) {
                v = OP::combine(v, lds[t + off * bw]);
                lds[t] = v;
            }
        }
    }
    return v;
}

// One thread's share of one output: `X(p)` = element p of the output's reduction range, `shift` = elements by which that range
// starts past a VEC-aligned address in the reference's planar tensor.
template <class OP, class LOAD>
static __device__ __forceinline__ typename OP::acc_t ts_thread_reduce(LOAD X, int64_t n, int shift, bool vectorize, int T, int t, int tx,
                                                                      bool tail_ok) {
    constexpr int VEC = OP::VEC;
    typename OP::acc_t acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = OP::ident();
    if (vectorize) {
        int64_t start = 0, end = n;
        if (shift > 0) {
            if (tx >= shift && tx < VEC && tail_ok) acc[0] = OP::template reduce<true>(acc[0], X((int64_t)(tx - shift)));
            start = VEC - shift;
            end = n + shift - VEC;
        }
        for (int64_t idx = t; idx * VEC + VEC - 1 < end; idx += T) {
            const int64_t p = start + idx * VEC;
            acc[0] = OP::template reduce<false>(acc[0], X(p));        // accumulator 0: unfused in libtorch_hip.so's main loop
#pragma unroll
            for (int i = 1; i < VEC; ++i) acc[i] = OP::template reduce<true>(acc[i], X(p + i));
        }
        const int64_t tail_start = end - end % VEC;
        if (tail_ok && tail_start + tx < end) acc[0] = OP::template reduce<true>(acc[0], X(start + tail_start + tx));
    } else {
        int64_t idx = t;
        while (idx + (int64_t)(VEC - 1) * T < n) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = OP::template reduce<true>(acc[i], X(idx + (int64_t)i * T));
            idx += (int64_t)T * VEC;
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            if (idx < n) acc[i] = OP::template reduce<true>(acc[i], X(idx));
            idx += T;
        }
    }
#pragma unroll
    for (int i = 1; i < VEC; ++i) acc[0] = OP::combine(acc[0], acc[i]);
    return acc[0];
}

// General form: one workgroup per output plane (frame, channel); any geometry, any size, any alignment.  `lab` = first frame of
// the group; plane = plane0 + blockIdx.x; frames are `chunk_frames` per reference call (the plane's position in ITS call fixes
// the alignment of its first element in the reference's tensor).  out[frame][channel][WHICH].
template <class OP, int WHICH>
__global__ void __launch_bounds__(512) k_tstats_plane(const float* __restrict__ lab, int64_t n, int64_t plane0, int chunk_frames, TsCfg cfg,
                                                      float factor, float eps, float* __restrict__ out) {
    __shared__ typename OP::acc_t lds[512];
    const int64_t plane = plane0 + blockIdx.x;
    const int64_t f = plane / 3;
    const int c = (int)(plane % 3);
    const int64_t o_in_call = (f % chunk_frames) * 3 + c;
    const int shift = cfg.vectorize ? (int)((o_in_call * n) % OP::VEC) : 0;
    const int rows = cfg.split ? cfg.bh : 1;
    const int T = cfg.bw * rows;
    const int t = threadIdx.x;
    const bool live = t < T;
    const float* base = lab + (size_t)f * (size_t)n * 3 + c;
    auto X = [base](int64_t p) { return base[(size_t)p * 3]; };
    typename OP::acc_t v = OP::ident();
    if (live) v = ts_thread_reduce<OP>(X, n, shift, cfg.vectorize != 0, T, t, t % cfg.bw, cfg.split ? (t / cfg.bw == 0) : true);
    v = ts_block_reduce<OP>(v, cfg.bw, cfg.bh, cfg.split != 0, t, live, lds);
    if (t == 0) {
        if constexpr (WHICH == 0) out[(size_t)f * 6 + c * 2] = v * factor;
        else out[(size_t)f * 6 + c * 2 + 1] = welf_std(v) + eps;
    }
}

// Video-sized frames (H*W % 4 == 0, 512 cooperating threads for both reductions): the frame is walked ONCE by a workgroup that
// feeds the mean accumulators (vectors of 4 pixels, 512 vectors apart) and the Welford accumulators (vectors of 2 pixels) of torch's
// thread of the same index for all three channels -- 12 B/px of HBM traffic for both statistics.  A "round" = 2048 pixels: one mean
// vector and two Welford vectors per thread; TS_DEPTH rounds of loads are kept in flight in registers (the update chains are
// sequential per thread, so nothing else hides the memory latency at 2 waves per SIMD).
//   PART -1: everything in one workgroup per frame (large batches: HBM bound, 4.6 TB/s measured);
//   PART 0..2 / 3: the Welford reduction of one channel / the three means -- four workgroups per frame for small batches, where
//   one workgroup's chain latency (16,200 dependent Welford updates per accumulator at 4K) would be all there is.  The four
//   workgroups of a frame are placed on one XCD (workgroup id % 8) so that three of them read from L2 what the first one fetched.

template <int PART>
struct TsRound {
    static constexpr bool MEAN = PART == -1 || PART == 3;
    static constexpr bool WELF = PART != 3;
    static constexpr int NC = PART == -1 ? 3 : 1;                       // Welford channels held
    f32x4 m[MEAN ? 3 : 1];
    float w[WELF ? 2 * 2 * NC : 1];                                     // [step][pixel][channel]
    __device__ __forceinline__ void load(const float* __restrict__ base, int64_t r, int t) {
        if constexpr (MEAN) {
            const f32x4* q = reinterpret_cast<const f32x4*>(base + (size_t)(r * 512 + t) * 12);
            m[0] = q[0]; m[1] = q[1]; m[2] = q[2];
        }
        if constexpr (WELF) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float* q = base + (size_t)(r * 1024 + s * 512 + t) * 6;
                if constexpr (PART == -1) {
                    const float2* q2 = reinterpret_cast<const float2*>(q);
                    const float2 b0 = q2[0], b1 = q2[1], b2 = q2[2];
                    w[s * 6 + 0] = b0.x; w[s * 6 + 1] = b0.y; w[s * 6 + 2] = b1.x; w[s * 6 + 3] = b1.y; w[s * 6 + 4] = b2.x; w[s * 6 + 5] = b2.y;
                } else {
                    w[s * 2 + 0] = q[PART];
                    w[s * 2 + 1] = q[3 + PART];
                }
            }
        }
    }
};

// ROWS = false: a 512-thread workgroup is torch's whole block and finishes the reduction (block_x_reduce, block_y_reduce, project).
// ROWS = true: a 256-thread workgroup is HALF of torch's block -- threads [t0, t0 + 256), i.e. whole rows of the (bw, bh) block, since
// bw divides 256 -- and stops after block_x_reduce: the per-row results go to `rows_out` ([3 channels][8 rows] Welf records, then
// [3][8] floats for the means) and k_tstats_rows_finish runs block_y_reduce + project over them.  One wave per SIMD instead of two:
// the Welford update chains of a wave are issue bound next to a second wave's (170 cycles per update with two waves per SIMD).
struct TsRows { Welf w[3][8]; float m[3][8]; };

// The Welford update's division by the running count, delta / n, is what makes its chain long: the backend's IEEE sequence is nine
// dependent instructions (v_div_scale, v_rcp_f32, five FMAs, v_div_fmas, v_div_fixup) of the thirteen per update, ~190 cycles per
// update measured.  n is the same for every accumulator of a thread and known before the data arrives, so rn = 1.0f / n (IEEE, correctly
// rounded) is formed OFF the chain and the quotient by Markstein's sequence  q = delta * rn;  e = fma(-n, q, delta);  q' = fma(e, rn, q)
// -- three dependent operations -- which returns the correctly rounded quotient RN(delta / n) whenever rn is the correctly rounded
// reciprocal and nothing under- or overflows (Markstein 1990; Cornea / Harrison / Tang: exceptions only for divisors with an all-ones
// significand, which a count below 2^24 - 1 never is).  Nothing can under- or overflow for 2^-100 <= |delta| <= 2^100 or delta == 0; a
// thread that ever sees another delta (or a NaN) raises `bad`, and a workgroup with a raised flag throws its accumulators away and
// repeats the frame with the IEEE division (ts_accumulate<..., false>): the result is the IEEE one for every input, the common case
// pays three instructions next to -- not on -- the chain.  tests: every statistics test compares with torch's own kernels; the
// fallback is forced by test_device_statistics_markstein_fallback (subnormal-range frames).
struct TsAcc {
    float ma[3][4];
    Welf wa[3][2];
    int cnt;
};

static __device__ __forceinline__ bool ts_delta_safe(float d) {
    const float a = __builtin_fabsf(d);
    return ((a >= 0x1p-100f) & (a <= 0x1p+100f)) | (d == 0.0f);
}

template <bool FUSED, bool FAST>
static __device__ __forceinline__ void ts_welf_update(Welf& a, float x, float nf1, float rn, bool& bad) {
    const float delta = x - a.mean;
    float q;
    if (FAST) {
        bad = bad | !ts_delta_safe(delta);
        const float q0 = delta * rn;
        const float e = __builtin_fmaf(-nf1, q0, delta);
        q = __builtin_fmaf(e, rn, q0);
    } else {
        q = delta / nf1;
    }
    const float mean1 = a.mean + q;
    const float d2 = x - mean1;
    a.m2 = FUSED ? __builtin_fmaf(delta, d2, a.m2) : a.m2 + delta * d2;
    a.mean = mean1;
}

// The thread loop of one frame part: every full round, then the last partial one.  Returns the `bad` flag of the FAST form.
template <int PART, int TS_DEPTH, bool FAST>
static __device__ __forceinline__ bool ts_accumulate(const float* __restrict__ base, int64_t n, int t, TsAcc& A) {
    typedef TsRound<PART> R;
    constexpr int NC = R::NC, C0 = PART == -1 ? 0 : (PART == 3 ? 0 : PART);
    const int64_t nvm = n / 4, nvw = n / 2;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) A.ma[c][i] = 0.0f;
        A.wa[c][0] = WelfOp::ident(); A.wa[c][1] = WelfOp::ident();
    }
    A.cnt = 0;

    auto mean_vec = [&](const f32x4& a0, const f32x4& a1, const f32x4& a2) {
        const float e[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) A.ma[c][i] = A.ma[c][i] + e[i * 3 + c];
    };
    auto welf_vec = [&](const float* e) {              // e[pixel][channel], NC channels: one update of accumulator 0 and of accumulator 1
        const int n1 = A.cnt + 1;
        const float nf1 = (float)n1;
        const float rn = FAST ? 1.0f / nf1 : 0.0f;     // IEEE; depends on the count only, not on the data
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            ts_welf_update<false, FAST>(A.wa[c][0], e[c], nf1, rn, bad);          // accumulator 0: unfused in libtorch_hip.so's main loop
            ts_welf_update<true, FAST>(A.wa[c][1], e[NC + c], nf1, rn, bad);
        }
        A.cnt = n1;
    };
    // full rounds (every thread has one mean vector and two Welford vectors), TS_DEPTH rounds of loads in flight
    const int64_t rounds = nvm / 512;
    R buf[TS_DEPTH];
#pragma unroll
    for (int i = 0; i < TS_DEPTH; ++i)
        if (i < rounds) buf[i].load(base, i, t);
    for (int64_t r0 = 0; r0 < rounds; r0 += TS_DEPTH) {
#pragma unroll
        for (int i = 0; i < TS_DEPTH; ++i) {
            const int64_t r = r0 + i;
            if (r < rounds) {
                if constexpr (R::MEAN) mean_vec(buf[i].m[0], buf[i].m[1], buf[i].m[2]);
                if constexpr (R::WELF) { welf_vec(buf[i].w); welf_vec(buf[i].w + 2 * NC); }
                if (r + TS_DEPTH < rounds) buf[i].load(base, r + TS_DEPTH, t);
            }
        }
    }
    // the last partial round
    if constexpr (R::MEAN)
        for (int64_t idx = rounds * 512 + t; idx < nvm; idx += 512) {
            const f32x4* q = reinterpret_cast<const f32x4*>(base + (size_t)idx * 12);
            mean_vec(q[0], q[1], q[2]);
        }
    if constexpr (R::WELF)
        for (int64_t idx = rounds * 1024 + t; idx < nvw; idx += 512) {
            const float* q = base + (size_t)idx * 6;
            float e[2 * NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) { e[c] = q[C0 + c]; e[NC + c] = q[3 + C0 + c]; }
            welf_vec(e);
        }
    // the counts as WelfordOps keeps them (int n and float nf; every accumulator of the thread took the same number of updates)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        A.wa[c][0].n = A.cnt; A.wa[c][0].nf = (float)A.cnt;
        A.wa[c][1].n = A.cnt; A.wa[c][1].nf = (float)A.cnt;
    }
    return bad;
}

#ifndef VRG_TS_MARKSTEIN
#define VRG_TS_MARKSTEIN 1
#endif

template <int PART, int TS_DEPTH, bool ROWS = false>
static __device__ void ts_frame_part(const float* __restrict__ base, int64_t n, int bw, int bh, float factor, float eps, float* __restrict__ o6,
                                     Welf* lds_w, int t0 = 0, TsRows* rows_out = nullptr) {
    typedef TsRound<PART> R;
    constexpr int NC = R::NC, C0 = PART == -1 ? 0 : (PART == 3 ? 0 : PART);
    float* lds_m = reinterpret_cast<float*>(lds_w);
    const int t = t0 + (int)threadIdx.x;
    TsAcc A;
    if (VRG_TS_MARKSTEIN && R::WELF) {
        const bool bad = ts_accumulate<PART, TS_DEPTH, true>(base, n, t, A);
        if (__syncthreads_or(bad ? 1 : 0)) ts_accumulate<PART, TS_DEPTH, false>(base, n, t, A);      // a delta outside the proven range: the IEEE division
    } else {
        ts_accumulate<PART, TS_DEPTH, false>(base, n, t, A);
    }
    float (&ma)[3][4] = A.ma;
    Welf (&wa)[3][2] = A.wa;
    const int tl = (int)threadIdx.x;                   // index within this workgroup (== t unless ROWS)
    if constexpr (R::MEAN) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float m = ma[c][0];
            m = m + ma[c][1];
            m = m + ma[c][2];
            m = m + ma[c][3];
            if constexpr (ROWS) {
                m = ts_block_reduce<MeanOp>(m, bw, bh, false, tl, true, lds_m);          // block_x_reduce only
                if (tl % bw == 0) rows_out->m[c][t / bw] = m;
            } else {
                m = ts_block_reduce<MeanOp>(m, bw, bh, true, t, true, lds_m);
                if (t == 0) o6[c * 2] = m * factor;
            }
        }
    }
    if constexpr (R::WELF) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            Welf w = WelfOp::combine(wa[c][0], wa[c][1]);
            if constexpr (ROWS) {
                w = ts_block_reduce<WelfOp>(w, bw, bh, false, tl, true, lds_w);
                if (tl % bw == 0) rows_out->w[C0 + c][t / bw] = w;
            } else {
                w = ts_block_reduce<WelfOp>(w, bw, bh, true, t, true, lds_w);
                if (t == 0) o6[(C0 + c) * 2 + 1] = welf_std(w) + eps;
            }
        }
    }
}

// Small batches: eight 256-thread workgroups per frame -- (Welford of channel 0 / 1 / 2, the three means) x (lower / upper half of
// torch's 512 threads) -- all on one XCD (workgroup id % 8), one wave per SIMD.
template <int DEPTH>
__global__ void __launch_bounds__(256) k_tstats_rows(const float* __restrict__ lab, int64_t n, int64_t frames, int bw, int bh, TsRows* __restrict__ rows) {
    __shared__ Welf lds_w[256];
    const int64_t w = blockIdx.x;
    const int64_t f = (w & 7) + 8 * (w >> 6);
    const int sub = (int)((w >> 3) & 7), part = sub >> 1, t0 = (sub & 1) * 256;
    if (f >= frames) return;
    const float* base = lab + (size_t)f * (size_t)n * 3;
    TsRows* ro = rows + f;
    switch (part) {
        case 0: ts_frame_part<0, DEPTH, true>(base, n, bw, bh, 0.0f, 0.0f, nullptr, lds_w, t0, ro); break;
        case 1: ts_frame_part<1, DEPTH, true>(base, n, bw, bh, 0.0f, 0.0f, nullptr, lds_w, t0, ro); break;
        case 2: ts_frame_part<2, DEPTH, true>(base, n, bw, bh, 0.0f, 0.0f, nullptr, lds_w, t0, ro); break;
        default: ts_frame_part<3, DEPTH, true>(base, n, bw, bh, 0.0f, 0.0f, nullptr, lds_w, t0, ro); break;
    }
}

// block_y_reduce (the tree over the bh rows, in the order ts_block_reduce walks it) + project, one thread per output
__global__ void __launch_bounds__(64) k_tstats_rows_finish(const TsRows* __restrict__ rows, int64_t frames, int bh, float factor, float eps,
                                                          float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= frames * 6) return;
    const int64_t f = i / 6;
    const int c = (int)(i % 6) >> 1, which = (int)(i & 1);
    if (which == 0) {
        float v[8];
        for (int r = 0; r < bh; ++r) v[r] = rows[f].m[c][r];
        for (int off = bh / 2; off > 0; off >>= 1)
            for (int ty = 0; ty < off; ++ty) v[ty] = MeanOp::combine(v[ty], v[ty + off]);
        out[(size_t)f * 6 + c * 2] = v[0] * factor;
    } else {
        Welf v[8];
        for (int r = 0; r < bh; ++r) v[r] = rows[f].w[c][r];
        for (int off = bh / 2; off > 0; off >>= 1)
            for (int ty = 0; ty < off; ++ty) v[ty] = WelfOp::combine(v[ty], v[ty + off]);
        out[(size_t)f * 6 + c * 2 + 1] = welf_std(v[0]) + eps;
    }
}

template <bool SPLIT, int DEPTH>
__global__ void __launch_bounds__(512) k_tstats_frame(const float* __restrict__ lab, int64_t n, int64_t frames, int bw, int bh, float factor,
                                                      float eps, float* __restrict__ out) {
    __shared__ Welf lds_w[512];
    if constexpr (!SPLIT) {
        const int64_t f = blockIdx.x;
        ts_frame_part<-1, DEPTH>(lab + (size_t)f * (size_t)n * 3, n, bw, bh, factor, eps, out + (size_t)f * 6, lds_w);
    } else {
        // workgroup w -> XCD w % 8: frame f = (w % 8) + 8 * (w / 32), part = (w / 8) % 4, so a frame's four parts share an L2
        const int64_t w = blockIdx.x;
        const int64_t f = (w & 7) + 8 * (w >> 5);
        const int part = (int)((w >> 3) & 3);
        if (f >= frames) return;
        const float* base = lab + (size_t)f * (size_t)n * 3;
        float* o6 = out + (size_t)f * 6;
        switch (part) {
            case 0: ts_frame_part<0, DEPTH>(base, n, bw, bh, factor, eps, o6, lds_w); break;
            case 1: ts_frame_part<1, DEPTH>(base, n, bw, bh, factor, eps, o6, lds_w); break;
            case 2: ts_frame_part<2, DEPTH>(base, n, bw, bh, factor, eps, o6, lds_w); break;
            default: ts_frame_part<3, DEPTH>(base, n, bw, bh, factor, eps, o6, lds_w); break;
        }
    }
}

// planes [o0, o1) of one reference call: the sub-iterators TensorIterator::with_32bit_indexing would produce (depth first, first half =
// floor(size / 2)) when the call's tensor has more than 2^29 elements; each has its own geometry
// (the mean factor is NOT per sub-iterator: mean_kernel_impl forms float(num_output_elements) / numel once from the whole call and
// gpu_reduce_kernel hands the same `ops` to every sub_iter, Reduce.cuh:1273-1279 -- (float)O_sub / (float)(O_sub * n) differs from it
// by one ulp whenever O * n is not an fp32 number, e.g. 87 x 1079 x 1919 frames)
static int ts_launch_planes(const float* lab_call, int64_t n, int64_t o0, int64_t o1, int chunk_frames, float factor, float eps, float* out_call,
                            int num_mp, hipStream_t st) {
    const int64_t O = o1 - o0;
    if (O * n > ((int64_t)1 << 29) && O > 1) {
        const int64_t half = O / 2;
        int rc = ts_launch_planes(lab_call, n, o0, o0 + half, chunk_frames, factor, eps, out_call, num_mp, st);
        if (rc != VRG_OK) return rc;
        return ts_launch_planes(lab_call, n, o0 + half, o1, chunk_frames, factor, eps, out_call, num_mp, st);
    }
    if (O * n > ((int64_t)1 << 29)) return VRG_ERR_UNSUPPORTED;      // a single plane beyond 32-bit indexing: torch splits the reduction itself
    TsCfg cm, cw;
    if (!ts_config(O, n, 4, cm, num_mp) || !ts_config(O, n, 2, cw, num_mp)) return VRG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((k_tstats_plane<MeanOp, 0>), dim3((unsigned)O), dim3(512), 0, st, lab_call, n, o0, chunk_frames, cm, factor, eps, out_call);
    hipLaunchKernelGGL((k_tstats_plane<WelfOp, 1>), dim3((unsigned)O), dim3(512), 0, st, lab_call, n, o0, chunk_frames, cw, factor, eps, out_call);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

#ifndef VRG_TS_SPLIT_MAX_FRAMES
#define VRG_TS_SPLIT_MAX_FRAMES 64
#endif
constexpr int64_t TS_SPLIT_MAX_FRAMES = VRG_TS_SPLIT_MAX_FRAMES;

// `count` reference calls of `b` frames each, starting at `lab` / `out`
#ifndef VRG_TS_ROWS_MAX_FRAMES
#define VRG_TS_ROWS_MAX_FRAMES 32
#endif
constexpr int64_t TS_ROWS_MAX_FRAMES = VRG_TS_ROWS_MAX_FRAMES;

static int ts_launch_calls(const float* lab, int64_t n, int64_t count, int b, float eps, float* out, int num_mp, void* scratch, int64_t scratch_bytes,
                           hipStream_t st) {
    if (count <= 0 || b <= 0) return VRG_OK;
    const int64_t O = (int64_t)b * 3;
    const float factor = (float)O / (float)(O * n);                  // static_cast<float>(num_output_elements) / numel, of the WHOLE call
    if (O * n > ((int64_t)1 << 29)) {
        for (int64_t k = 0; k < count; ++k) {
            const int rc = ts_launch_planes(lab + (size_t)k * b * n * 3, n, 0, O, b, factor, eps, out + (size_t)k * b * 6, num_mp, st);
            if (rc != VRG_OK) return rc;
        }
        return VRG_OK;
    }
    TsCfg cm, cw;
    if (!ts_config(O, n, 4, cm, num_mp) || !ts_config(O, n, 2, cw, num_mp)) return VRG_ERR_UNSUPPORTED;
    const int64_t frames = count * b;
    const bool whole = (n % 4 == 0) && cm.vectorize && cw.vectorize && cm.split && cw.split && cm.bw * cm.bh == 512 && cw.bw * cw.bh == 512 &&
                       cm.bw == cw.bw;
    if (whole) {
        // measured on the MI355X (4K frames, ms): 1 frame 1.45 split / 3.57 whole; 64 frames 1.98 / 3.76; 128 frames 5.2 / 3.9 (four
        // readers per frame stop sharing their L2 lines); 256 frames 11.5 / 5.5 (the whole-frame form runs at 4.6 TB/s there).
        // Two rounds of loads in flight help the latency-bound split form (1.45 vs 1.72 with four), none the HBM-bound one.
        // up to 32 frames (and a caller-supplied scratch buffer): eight half-block workgroups per frame, one wave per SIMD
        if (frames <= TS_ROWS_MAX_FRAMES && scratch && scratch_bytes >= frames * (int64_t)sizeof(TsRows) && cm.bh == cw.bh && cm.bh <= 8 && 256 % cm.bw == 0 &&
            (reinterpret_cast<uintptr_t>(scratch) & 15) == 0) {
            TsRows* rows = reinterpret_cast<TsRows*>(scratch);
            hipLaunchKernelGGL((k_tstats_rows<2>), dim3((unsigned)(64 * ((frames + 7) / 8))), dim3(256), 0, st, lab, n, frames, cm.bw, cm.bh, rows);
            hipLaunchKernelGGL(k_tstats_rows_finish, dim3((unsigned)((frames * 6 + 63) / 64)), dim3(64), 0, st, rows, frames, cm.bh, factor, eps, out);
            VRG_CHECK_LAUNCH();
            return VRG_OK;
        }
        const bool split = frames <= TS_SPLIT_MAX_FRAMES;
        const int depth = split ? 2 : 1;
        const dim3 grid(split ? (unsigned)(32 * ((frames + 7) / 8)) : (unsigned)frames);
#define TS_LAUNCH(S, D) hipLaunchKernelGGL((k_tstats_frame<S, D>), grid, dim3(512), 0, st, lab, n, frames, cm.bw, cm.bh, factor, eps, out)
        if (depth == 2) TS_LAUNCH(true, 2); else TS_LAUNCH(false, 1);
#undef TS_LAUNCH
    } else {
        // all planes of all calls in one launch: plane -> (frame, channel), position in its call from frame % b
        hipLaunchKernelGGL((k_tstats_plane<MeanOp, 0>), dim3((unsigned)(frames * 3)), dim3(512), 0, st, lab, n, (int64_t)0, b, cm, factor, eps, out);
        hipLaunchKernelGGL((k_tstats_plane<WelfOp, 1>), dim3((unsigned)(frames * 3)), dim3(512), 0, st, lab, n, (int64_t)0, b, cw, factor, eps, out);
    }
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

}  // namespace vrg

extern "C" int64_t vrg_lab_stats_torch_scratch_bytes(int64_t frames) {
    if (frames <= 0) return 0;
    const int64_t f = frames < vrg::TS_ROWS_MAX_FRAMES ? frames : vrg::TS_ROWS_MAX_FRAMES;       // only the small-batch form uses it
    return f * (int64_t)sizeof(vrg::TsRows);
}

extern "C" int vrg_lab_stats_torch_f32(const float* lab, int64_t frames, int32_t height, int32_t width, int32_t chunk_frames,
                                       float* mean_std, float eps, void* stream) {
    return vrg_lab_stats_torch_ws_f32(lab, frames, height, width, chunk_frames, mean_std, eps, nullptr, 0, stream);
}

extern "C" int vrg_lab_stats_torch_ws_f32(const float* lab, int64_t frames, int32_t height, int32_t width, int32_t chunk_frames,
                                          float* mean_std, float eps, void* scratch, int64_t scratch_bytes, void* stream) {
    using namespace vrg;
    if (frames < 0 || height <= 0 || width <= 0 || chunk_frames <= 0) return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    if (!lab || !mean_std) return VRG_ERR_BAD_ARG;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VRG_ERR_NO_DEVICE;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return VRG_ERR_NO_DEVICE;
    // The CU count enters torch's geometry twice: target_grid_size (inert here, see ts_config) and `force_splitting_output`, which
    // setReduceConfig only considers on devices with fewer than 100 CUs (a CPX-partitioned MI355X has 32): not replayed, the host falls
    // back to the fp64 statistics there (ops._cm_stats).
    if (cus <= 0) return VRG_ERR_NO_DEVICE;
    if (cus < 100) return VRG_ERR_UNSUPPORTED;
    const int64_t n = (int64_t)height * width;
    if (frames * 3 > 0x7fffffff / 2) return VRG_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int64_t full = frames / chunk_frames;
    const int tail = (int)(frames % chunk_frames);
    int rc = ts_launch_calls(lab, n, full, chunk_frames, eps, mean_std, cus, scratch, scratch_bytes, st);
    if (rc != VRG_OK) return rc;
    if (tail) rc = ts_launch_calls(lab + (size_t)full * chunk_frames * n * 3, n, 1, tail, eps, mean_std + (size_t)full * chunk_frames * 6, cus, scratch, scratch_bytes, st);
    return rc;
}

// Host-only: the geometry ts_config derives for (num_outputs, reduction length, vector width): cfg4 = {block_width, block_height,
// split across warps, vectorised}; VRG_ERR_UNSUPPORTED where torch would split the reduction across workgroups.  No GPU needed
// (tests/test_torch_reduce_oracle.py compares it with what rocprofv3 recorded for torch's own launches).
extern "C" int vrg_debug_torch_reduce_config(int64_t num_outputs, int64_t reduce_len, int32_t vec, int32_t* cfg4) {
    if (!cfg4 || num_outputs < 1 || reduce_len < 1 || (vec != 2 && vec != 4)) return VRG_ERR_BAD_ARG;
    vrg::TsCfg c;
    if (!vrg::ts_config(num_outputs, reduce_len, vec, c)) return VRG_ERR_UNSUPPORTED;
    cfg4[0] = c.bw; cfg4[1] = c.bh; cfg4[2] = c.split; cfg4[3] = c.vectorize;
    return VRG_OK;
}
