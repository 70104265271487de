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
#include <atomic>
#include "vrg_tstats_body.hpp"
#include "vrg_tstats_config.hpp"

namespace vrg {

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

// Small batches: eight 256-thread workgroups per frame -- (Welford of channel 0 / 1 / 2, the three means) x (lower / upper half of
// torch's 512 threads) -- all on one XCD (workgroup id % 8), one wave per SIMD.
template <int DEPTH>
__global__ void __launch_bounds__(256) k_tstats_rows(const float* __restrict__ lab, int64_t n, int64_t frames, int bw, int bh, TsRows* __restrict__ rows) {
    __shared__ Welf lds_w[256];
    tstats_rows_body<DEPTH>(blockIdx.x, lab, n, frames, bw, bh, rows, lds_w);
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

// ---------------------------------------------------------------------------------------------------------------------
// Round 4 -- small batches (<= TS_ROWS_MAX_FRAMES frames, a scratch buffer): ONE ACCUMULATOR PER LANE.
// torch's reduction of a plane is 512 threads x 2 Welford accumulators (+ 512 x 4 mean accumulators), each a strictly sequential
// chain over n / 1024 (n / 2048) elements; nothing about the ORDER says that the two accumulators of a thread have to live in the
// same lane, or the three channels in the same workgroup.  The half-block form above keeps torch's thread = our lane: 8 workgroups
// of 4 waves per frame, every wave issuing ~80 instructions per round for its two accumulators (IEEE reciprocal of the count,
// two updates, range flags) at the single-wave issue rate -- 1.39 ms per 4K frame, while 97 % of the chip idles -- and every one
// of the 8 workgroups streams the WHOLE frame through its CU's L1 (99.5 MB at ~140 GB/s per CU = 0.7 ms: the floor of that form).
// Here a workgroup owns 64 of torch's 512 threads and runs them as SEVEN waves: wave (c, a), c = 0..2, a = 0..1, carries Welford
// accumulator a of channel c of those 64 threads -- one update per step and lane, ~14 instructions --, the seventh wave their
// mean accumulators.  The six Welford waves walk the same 1.5 KB of the interleaved image per step (each uses 4 of every 24 bytes,
// together all of them), so the CU's L1 fetches every line once and the workgroup streams 1/8 of the frame.  The reciprocal of the
// running count -- the same number for every lane, known before any data -- comes from an LDS table {(float)k, RN(1 / k)} filled once
// per workgroup with the IEEE division (k <= TS_LANES_MAX_STEPS; a table in global memory read with scalar loads measured 1.17 ms instead of 0.78).
// 65 KB of LDS and 250 VGPRs per workgroup: this is the LATENCY form -- it wins when the GPU is otherwise idle (the statistics of the reference frame of a
// SMALL step: 4 / 8 / 16 frames per step 2.0 / 2.8 / 4.5 ms against 2.4 / 3.4 / 5.0) and loses beside a full-size pass 1 (64 frames: 16.8 against 14.4 ms:
// it cannot be placed until pass 1 drains, then delays pass 2), so it is chosen by the CALLER (vrg_lab_stats_torch_lat_f32), never automatically.  Per-thread accumulators go to a scratch record; the finishing
// kernel (512 threads per frame = torch's block) combines accumulators 0 and 1 of each thread and runs block_x_reduce /
// block_y_reduce / project exactly as ts_frame_part does.  Same update arithmetic (ts_welf_update, Markstein division behind the
// range flag, IEEE repeat of the workgroup when the flag is raised), same order: bit-identical.
// ---------------------------------------------------------------------------------------------------------------------
struct TsLaneRec { float mean, m2; int cnt; int pad; };
struct TsLanes { TsLaneRec w[3][2][512]; float m[3][512]; };
constexpr int TS_LANES_MAX_STEPS = 16384;            // table entries (LDS: 8 bytes each) -- frames up to 16.7 M pixels

#define VRG_TS_LANES_DEPTH 8      /* Welford steps (one 4-byte load per lane each) requested ahead.  8 = a 12 KB window shared by the six Welford waves in the CU's 32 KB L1; 16 / 32 / 48 measured SLOWER (0.95 / 0.91 / 2.3 ms per 4K frame against 0.76-0.79: the window falls out of the L1, then out of the registers) -- profiles/r04_bench_stats_lanes_depths.json */
#define VRG_TS_LANES_MEAN_DEPTH 12   /* mean rounds (48 bytes per lane each) requested ahead (16: more than 256 registers, 2.1 ms) */
#define VRG_TS_LANES_MAX_FRAMES 2    /* one 4K frame 0.78 ms against the half-block form's 1.12 (-30 %); 4-16 frames: equal; 32: the half-block form wins (1.25 against 1.6) */

template <bool FUSED, bool FAST, int DEPTH>
static __device__ __forceinline__ bool ts_lane_chain(const float* __restrict__ ub, uint32_t lane_bytes, int steps, bool extra, const float2* __restrict__ tab,
                                                     Welf& acc, int& cnt) {
    // ub (wave-uniform): element of step 0 of thread 0 of this accumulator; lane_bytes: this lane's byte offset from it (loop invariant);
    // a step advances 1024 pixels = 12288 bytes (scalar arithmetic).  `steps` whole steps for every lane, then one more for the lanes
    // with `extra`.  Requests are branch-free and DEPTH steps ahead: a step index past the end is clamped to the last one (scalar min).
    bool bad = false;
    acc = WelfOp::ident();
    float xb[DEPTH];
    const int last = steps > 0 ? steps - 1 : 0;
    auto load = [&](int k) {
        const int kk = k < last ? k : last;                            // scalar
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ub + (size_t)kk * 3072) + lane_bytes);
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) xb[i] = load(i);
    int k0 = 0;
    for (; k0 + DEPTH <= steps; k0 += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const float2 t = tab[k0 + i];                             // {(float)(k + 1), RN(1 / (k + 1))}: uniform address, one LDS broadcast
            ts_welf_update<FUSED, FAST>(acc, xb[i], t.x, t.y, bad);
            xb[i] = load(k0 + i + DEPTH);
        }
    }
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        if (k0 + i < steps) {
            const float2 t = tab[k0 + i];
            ts_welf_update<FUSED, FAST>(acc, xb[i], t.x, t.y, bad);
        }
    cnt = steps;
    if (extra) {
        const float2 t = tab[steps];
        const float x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ub + (size_t)steps * 3072) + lane_bytes);
        ts_welf_update<FUSED, FAST>(acc, x, t.x, t.y, bad);
        cnt = steps + 1;
    }
    return bad;
}

template <int DEPTH>
__global__ void __launch_bounds__(448) k_tstats_lanes(const float* __restrict__ lab, int64_t n, int64_t frames, TsLanes* __restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) float2 ts_rn_tab[];          // [steps + 1]: {(float)(k + 1), RN(1 / (k + 1))}
    // workgroup w -> XCD w % 8: frame f = (w % 8) + 8 * (w / 64), thread group j = (w / 8) % 8: a frame's eight workgroups share an L2
    const int64_t w = blockIdx.x;
    const int64_t f = (w & 7) + 8 * (w >> 6);
    const int j = (int)((w >> 3) & 7);
    if (f >= frames) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int t = 64 * j + lane;                                                 // torch's thread index
    const int64_t nvw = n / 2, nvm = n / 4;
    const int64_t wsteps = nvw / 512;                                            // Welford steps every thread takes
    const bool wextra = (int64_t)t < nvw - wsteps * 512;
    for (int64_t i = threadIdx.x; i <= wsteps; i += 448) {
        const float nf = (float)(i + 1);
        ts_rn_tab[i] = float2{nf, 1.0f / nf};                                    // IEEE, correctly rounded (hipcc's default division)
    }
    __syncthreads();
    const float* base = lab + (size_t)f * (size_t)n * 3;
    TsLanes* R = recs + f;
    bool bad = false;
    Welf acc = WelfOp::ident();
    int cnt = 0;
    const bool fast = VRG_TS_MARKSTEIN && wsteps + 2 <= TS_MARKSTEIN_MAX_COUNT;
    const int c = wave >> 1, a = wave & 1;
    const float* ub = base + (wave < 6 ? a * 3 + c : 0);                          // wave-uniform
    const uint32_t lane_bytes = (uint32_t)t * 24u;                               // element (2 t + a) * 3 + c
    const int ws = (int)wsteps;
    if (wave < 6) {
        if (fast) bad = a ? ts_lane_chain<true, true, DEPTH>(ub, lane_bytes, ws, wextra, ts_rn_tab, acc, cnt) : ts_lane_chain<false, true, DEPTH>(ub, lane_bytes, ws, wextra, ts_rn_tab, acc, cnt);
        else if (a) ts_lane_chain<true, false, DEPTH>(ub, lane_bytes, ws, wextra, ts_rn_tab, acc, cnt);
        else ts_lane_chain<false, false, DEPTH>(ub, lane_bytes, ws, wextra, ts_rn_tab, acc, cnt);
    } else {
        // the mean accumulators of the 64 threads: vectors of four pixels, 512 vectors apart (ts_accumulate's mean_vec, thread t)
        float ma[3][4];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
#pragma unroll
            for (int i = 0; i < 4; ++i) ma[cc][i] = 0.0f;
        auto mean_vec = [&](const f32x4& a0, const f32x4& a1, const f32x4& a2) {
            const float e[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) ma[cc][i] = ma[cc][i] + e[i * 3 + cc];
        };
        constexpr int MD = VRG_TS_LANES_MEAN_DEPTH;
        const int64_t msteps = nvm / 512;
        const bool mextra = (int64_t)t < nvm - msteps * 512;
        const int64_t last = msteps > 0 ? msteps - 1 : 0;
        f32x4 mb[MD][3];
        auto mload = [&](int i, int64_t r) {
            const f32x4* q = reinterpret_cast<const f32x4*>(base + (size_t)(r * 512 + t) * 12);
            mb[i][0] = q[0]; mb[i][1] = q[1]; mb[i][2] = q[2];
        };
#pragma unroll
        for (int i = 0; i < MD; ++i) mload(i, i < last ? i : last);
        int64_t r0 = 0;
        for (; r0 + MD <= msteps; r0 += MD) {
#pragma unroll
            for (int i = 0; i < MD; ++i) {
                mean_vec(mb[i][0], mb[i][1], mb[i][2]);
                const int64_t nx = r0 + i + MD;
                mload(i, nx < last ? nx : last);
            }
        }
#pragma unroll
        for (int i = 0; i < MD; ++i)
            if (r0 + i < msteps) mean_vec(mb[i][0], mb[i][1], mb[i][2]);
        if (mextra) {
            const f32x4* q = reinterpret_cast<const f32x4*>(base + (size_t)(msteps * 512 + t) * 12);
            mean_vec(q[0], q[1], q[2]);
        }
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            float m = ma[cc][0];
            m = m + ma[cc][1];
            m = m + ma[cc][2];
            m = m + ma[cc][3];
            R->m[cc][t] = m;
        }
    }
    if (fast && __syncthreads_or(bad ? 1 : 0)) {                                 // a delta outside the proven range somewhere in the workgroup: IEEE division
        if (wave < 6) {
            if (a) ts_lane_chain<true, false, DEPTH>(ub, lane_bytes, ws, wextra, ts_rn_tab, acc, cnt);
            else ts_lane_chain<false, false, DEPTH>(ub, lane_bytes, ws, wextra, ts_rn_tab, acc, cnt);
        }
    }
    if (wave < 6) R->w[c][a][t] = TsLaneRec{acc.mean, acc.m2, cnt, 0};
}

// torch's block (512 threads) per frame: accumulators 0 and 1 of each thread combined, then block_x_reduce / block_y_reduce / project
__global__ void __launch_bounds__(512) k_tstats_lanes_finish(const TsLanes* __restrict__ recs, int bw, int bh, float factor, float eps, float* __restrict__ out) {
    __shared__ Welf lds_w[512];
    float* lds_m = reinterpret_cast<float*>(lds_w);
    const int64_t f = blockIdx.x;
    const int t = threadIdx.x;
    const TsLanes* R = recs + f;
    float* o6 = out + (size_t)f * 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float m = R->m[c][t];
        m = ts_block_reduce<MeanOp>(m, bw, bh, true, t, true, lds_m);
        if (t == 0) o6[c * 2] = m * factor;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const TsLaneRec r0 = R->w[c][0][t], r1 = R->w[c][1][t];
        const Welf w0{r0.mean, r0.m2, r0.cnt, (float)r0.cnt}, w1{r1.mean, r1.m2, r1.cnt, (float)r1.cnt};
        Welf w = WelfOp::combine(w0, w1);
        w = ts_block_reduce<WelfOp>(w, bw, bh, true, t, true, lds_w);
        if (t == 0) o6[c * 2 + 1] = welf_std(w) + eps;
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

#define VRG_TS_SPLIT_MAX_FRAMES 64
constexpr int64_t TS_SPLIT_MAX_FRAMES = VRG_TS_SPLIT_MAX_FRAMES;

// `count` reference calls of `b` frames each, starting at `lab` / `out`
#define VRG_TS_ROWS_DEPTH 4   /* rounds of loads in flight per thread: half-block form / four-workgroup form / whole-frame form */
#define VRG_TS_SPLIT_DEPTH 2
#define VRG_TS_WHOLE_DEPTH 1
#define VRG_TS_ROWS_MAX_FRAMES 32
constexpr int64_t TS_ROWS_MAX_FRAMES = VRG_TS_ROWS_MAX_FRAMES;

static int ts_launch_calls(const float* lab, int64_t n, int64_t count, int b, float eps, float* out, int num_mp, void* scratch, int64_t scratch_bytes,
                           hipStream_t st, bool prefer_lanes = false) {
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
        // Rounds of loads in flight (vrg_tstats_body.hpp, ts_accumulate): since the requests are branch-free (round 3) they really are in
        // flight -- four for the half-block form (one 4K frame 1.39 ms against 1.71 with two, 1.89 with round 2's conditional loads), two
        // for the four-workgroup form (64 frames 1.85 ms against 1.98; four: 2.75), the plain one-round loop for the HBM-bound whole-frame
        // form (256 frames 4.4 ms; branch-free 5.5, two / four rounds 6.3 / 6.7): profiles/r03_stats_prefetch_depth_ab.log.
        // up to 32 frames (and a caller-supplied scratch buffer): eight half-block workgroups per frame, one wave per SIMD
#define VRG_TS_LANES 1
        if (VRG_TS_LANES && prefer_lanes && frames <= VRG_TS_LANES_MAX_FRAMES && scratch && scratch_bytes >= frames * (int64_t)sizeof(TsLanes) && cm.bh == cw.bh &&
            n / 1024 + 2 <= TS_LANES_MAX_STEPS && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0) {
            // one accumulator per lane: eight 7-wave workgroups per frame + torch's block as the finishing kernel
            TsLanes* recs = reinterpret_cast<TsLanes*>(scratch);
            const size_t lds = (size_t)(n / 1024 + 2) * sizeof(float2);
            // the attribute belongs to the function ON A DEVICE: once per device of this process (a bit per device index; a GPU beyond
            // 64 sets it on every call), set by whichever host thread gets there first -- setting it twice is harmless
            static std::atomic<uint64_t> attr_devices{0};
            int device = 0;
            if (hipGetDevice(&device) != hipSuccess) return VRG_ERR_LAUNCH;
            const uint64_t bit = (device >= 0 && device < 64) ? (1ull << device) : 0;
            if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_tstats_lanes<VRG_TS_LANES_DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        TS_LANES_MAX_STEPS * (int)sizeof(float2)) != hipSuccess)
                    return VRG_ERR_LAUNCH;
                attr_devices.fetch_or(bit, std::memory_order_release);
            }
            hipLaunchKernelGGL((k_tstats_lanes<VRG_TS_LANES_DEPTH>), dim3((unsigned)(64 * ((frames + 7) / 8))), dim3(448), lds, st, lab, n, frames, recs);
            hipLaunchKernelGGL(k_tstats_lanes_finish, dim3((unsigned)frames), dim3(512), 0, st, recs, cm.bw, cm.bh, factor, eps, out);
            VRG_CHECK_LAUNCH();
            return VRG_OK;
        }
        if (frames <= TS_ROWS_MAX_FRAMES && scratch && scratch_bytes >= frames * (int64_t)sizeof(TsRows) && cm.bh == cw.bh && cm.bh <= 8 && 256 % cm.bw == 0 &&
            (reinterpret_cast<uintptr_t>(scratch) & 15) == 0) {
            TsRows* rows = reinterpret_cast<TsRows*>(scratch);
            hipLaunchKernelGGL((k_tstats_rows<VRG_TS_ROWS_DEPTH>), dim3((unsigned)(64 * ((frames + 7) / 8))), dim3(256), 0, st, lab, n, frames, cm.bw, cm.bh, rows);
            hipLaunchKernelGGL(k_tstats_rows_finish, dim3((unsigned)((frames * 6 + 63) / 64)), dim3(64), 0, st, rows, frames, cm.bh, factor, eps, out);
            VRG_CHECK_LAUNCH();
            return VRG_OK;
        }
        const bool split = frames <= TS_SPLIT_MAX_FRAMES;
        if (split) {
            hipLaunchKernelGGL((k_tstats_frame<true, VRG_TS_SPLIT_DEPTH>), dim3((unsigned)(32 * ((frames + 7) / 8))), dim3(512), 0, st, lab, n, frames, cm.bw,
                               cm.bh, factor, eps, out);
        } else {
            // One workgroup streams one frame.  Round 6 (profiles/r06_tstats_sizes.json): 256 4K frames in one launch -- one workgroup per CU --
            // read at 5.5 TB/s, 512 frames in one launch (two resident per CU: 512 concurrent 100 MB streams) at 4.0, 384 at 3.6, but 512 as
            // two launches of 256 at 5.5 again; address translation is not involved (TCP_UTCL1 misses: 75 of 9.5e8 requests), the L2's read
            // latency rises 838 -> 1003 cycles with the second resident stream per CU.  So: at most one workgroup per CU per launch.
            const int64_t per = num_mp > 0 ? num_mp : 256;
            for (int64_t f0 = 0; f0 < frames; f0 += per) {
                const int64_t nf = frames - f0 < per ? frames - f0 : per;
                hipLaunchKernelGGL((k_tstats_frame<false, VRG_TS_WHOLE_DEPTH>), dim3((unsigned)nf), dim3(512), 0, st, lab + (size_t)f0 * (size_t)n * 3, n, nf,
                                   cm.bw, cm.bh, factor, eps, out + (size_t)f0 * 6);
            }
        }
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
    if (frames <= 0 || frames > vrg::TS_ROWS_MAX_FRAMES) return 0;       // only the small-batch forms use a scratch buffer
    return frames * (int64_t)(sizeof(vrg::TsLanes) > sizeof(vrg::TsRows) ? sizeof(vrg::TsLanes) : sizeof(vrg::TsRows));       // the larger of the two small-batch forms' records
}

extern "C" int vrg_lab_stats_torch_f32(const float* lab, int64_t frames, int32_t height, int32_t width, int32_t chunk_frames,
                                       float* mean_std, float eps, void* stream) {
    return vrg_lab_stats_torch_ws_f32(lab, frames, height, width, chunk_frames, mean_std, eps, nullptr, 0, stream);
}

static int lab_stats_torch_any(const float* lab, int64_t frames, int32_t height, int32_t width, int32_t chunk_frames, float* mean_std, float eps, void* scratch,
                               int64_t scratch_bytes, void* stream, bool prefer_lanes);

extern "C" int vrg_lab_stats_torch_ws_f32(const float* lab, int64_t frames, int32_t height, int32_t width, int32_t chunk_frames,
                                          float* mean_std, float eps, void* scratch, int64_t scratch_bytes, void* stream) {
    return lab_stats_torch_any(lab, frames, height, width, chunk_frames, mean_std, eps, scratch, scratch_bytes, stream, false);
}

extern "C" int vrg_lab_stats_torch_lat_f32(const float* lab, int64_t frames, int32_t height, int32_t width, int32_t chunk_frames,
                                           float* mean_std, float eps, void* scratch, int64_t scratch_bytes, void* stream) {
    return lab_stats_torch_any(lab, frames, height, width, chunk_frames, mean_std, eps, scratch, scratch_bytes, stream, true);
}

static int lab_stats_torch_any(const float* lab, int64_t frames, int32_t height, int32_t width, int32_t chunk_frames, float* mean_std, float eps, void* scratch,
                               int64_t scratch_bytes, void* stream, bool prefer_lanes) {
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
    int rc = ts_launch_calls(lab, n, full, chunk_frames, eps, mean_std, cus, scratch, scratch_bytes, st, prefer_lanes);
    if (rc != VRG_OK) return rc;
    if (tail) rc = ts_launch_calls(lab + (size_t)full * chunk_frames * n * 3, n, 1, tail, eps, mean_std + (size_t)full * chunk_frames * 6, cus, scratch, scratch_bytes, st, prefer_lanes);
    return rc;
}
