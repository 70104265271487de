// vrg_tstats_body.hpp -- the device side of the torch-order statistics replay (vrg_torch_stats.hip): the reduction ops, torch's block reduce, and the whole-frame thread loops (see vrg_torch_stats.hip for
// the design and the sources they follow).
#pragma once
#include "vrg_common.hpp"

namespace vrg {

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
// -- three dependent operations.  That sequence is the correctly rounded quotient for most but not all divisors (Brisebarre / Muller /
// Raina 2004 characterise the exceptions), so equality with delta / n is established BY ENUMERATION instead: it commutes with the sign
// of delta and with scaling delta by a power of two while nothing leaves the normal range, hence 2^23 significands per count are all
// inputs, and vrg_selftest_welford_division sweeps every count up to TS_MARKSTEIN_MAX_COUNT = 2^20 x every significand: zero mismatches
// (profiles/r03_welford_division_sweep.json; larger frames take the IEEE division).  Nothing leaves the normal range for
// 2^-100 <= |delta| <= 2^100 or delta == 0; a
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
    // The loads are BRANCH-FREE (a round index past the end is clamped to the last round and its data is not used): a load inside a
    // conditional block makes the compiler wait for `vmcnt(0)` where the paths join, i.e. for the loads just issued -- the rounds
    // "in flight" were then never in flight (the ISA of round 2's loop had a vmcnt(0) per round; deeper TS_DEPTH only cost registers).
    const int64_t rounds = nvm / 512;
    R buf[TS_DEPTH];
    auto process = [&](const R& b) {
        if constexpr (R::MEAN) mean_vec(b.m[0], b.m[1], b.m[2]);
        if constexpr (R::WELF) { welf_vec(b.w); welf_vec(b.w + 2 * NC); }
    };
    if constexpr (TS_DEPTH == 1) {
        // one round per thread in flight (the whole-frame form of large batches: two workgroups per CU keep HBM busy by themselves, and
        // this plain form measured 4.4 ms per 256 x 4K frames against 5.5 for the branch-free one below)
        if (rounds > 0) buf[0].load(base, 0, t);
        for (int64_t r = 0; r < rounds; ++r) {
            process(buf[0]);
            if (r + 1 < rounds) buf[0].load(base, r + 1, t);
        }
    } else if (rounds > 0) {
        const int64_t last = rounds - 1;
#pragma unroll
        for (int i = 0; i < TS_DEPTH; ++i) buf[i].load(base, i < last ? i : last, t);
        int64_t r0 = 0;
        for (; r0 + TS_DEPTH <= rounds; r0 += TS_DEPTH) {            // whole groups of TS_DEPTH rounds: no conditional inside
#pragma unroll
            for (int i = 0; i < TS_DEPTH; ++i) {
                process(buf[i]);
                const int64_t nx = r0 + i + TS_DEPTH;
                buf[i].load(base, nx < last ? nx : last, t);
            }
        }
#pragma unroll
        for (int i = 0; i < TS_DEPTH; ++i)                            // the last, partial group (its rounds were requested by the last whole one)
            if (r0 + i < rounds) process(buf[i]);
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

#define VRG_TS_MARKSTEIN 1
// Counts up to which q' == delta / n is established by enumeration (vrg_selftest_welford_division over every count x every fp32
// significand: tests/test_gpu_parity.py sweeps a sample of counts, tools/welford_division_sweep.py all of them -- profiles/).
constexpr int64_t TS_MARKSTEIN_MAX_COUNT = 1 << 20;

template <int PART, int TS_DEPTH, bool ROWS = false>
static __device__ void ts_frame_part(const float* __restrict__ base, int64_t n, int bw, int bh, float factor, float eps, float* __restrict__ o6,
                                     Welf* lds_w, int t0 = 0, TsRows* rows_out = nullptr) {
    typedef TsRound<PART> R;
    constexpr int NC = R::NC, C0 = PART == -1 ? 0 : (PART == 3 ? 0 : PART);
    float* lds_m = reinterpret_cast<float*>(lds_w);
    const int t = t0 + (int)threadIdx.x;
    TsAcc A;
    if (VRG_TS_MARKSTEIN && R::WELF && n / 1024 + 2 <= TS_MARKSTEIN_MAX_COUNT) {       // (an accumulator takes at most n / 1024 + 1 updates)
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

// The body of one of the eight half-block workgroups of a frame (k_tstats_rows): `w` = the workgroup's index in that grid.
template <int DEPTH>
__device__ __forceinline__ void tstats_rows_body(int64_t w, const float* __restrict__ lab, int64_t n, int64_t frames, int bw, int bh,
                                                 TsRows* __restrict__ rows, Welf* lds_w) {
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

}  // namespace vrg
