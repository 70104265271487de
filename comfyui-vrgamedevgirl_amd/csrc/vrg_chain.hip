// vrg_chain.hip -- Lab statistics (colour-match pass 1) and the fused grain -> LUT -> colour match ->
// 3x3 sharpen chain.  gfx950 only.
//
// "pre" stages (grain, LUT, colour match) are pure per-pixel functions of the input pixel and its
// absolute position; chain_pre() evaluates them for one pixel.  The fused kernels call it
//   * once per pixel                      (point-wise chains: no sharpen stage),
//   * once per tile+halo pixel into LDS   (chains ending in a 3x3 stencil),
//   * once per pixel inside the reduction (statistics of the grain->LUT output).
// That makes every fused result bit-identical to running the stand-alone kernels back to back.
#include "vrg_common.hpp"
#include <type_traits>
#include "vrg_chain_stages.hpp"

namespace vrg {

constexpr int STATS_BPF_MAX = 128;        // reduction blocks per frame (function of the frame size only)
constexpr int STATS_PX_PER_BLOCK = 16384;

__host__ __device__ inline int stats_blocks_per_frame(int64_t pixels) {
    int64_t b = (pixels + STATS_PX_PER_BLOCK - 1) / STATS_PX_PER_BLOCK;
    return (int)(b < 1 ? 1 : (b > STATS_BPF_MAX ? STATS_BPF_MAX : b));
}

// ----------------------------------------------------------------------------------------------
// Point-wise chain (no sharpen): one pixel per thread.
// ----------------------------------------------------------------------------------------------
template <int STAGES, class IO>
__global__ __launch_bounds__(256) void k_chain_pointwise(const typename IO::elem* __restrict__ in, typename IO::elem* __restrict__ out,
                                                          int32_t ppf, ChainK D) {
    VRG_CM_MATH(PT, (STAGES & VRG_STAGE_COLORMATCH) != 0, (STAGES & VRG_STAGE_FASTMATH) != 0, D.dm);
    const int32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= ppf) return;
    const int64_t f = blockIdx.y;
    const px3 v = IO::load_stream(in + f * ppf + p);
    const float x[3] = {v.r, v.g, v.b};
    float o[3];
    chain_pre<STAGES>(D, frame_ctx<STAGES>(D, f), p, x, o, PT);
    IO::store_stream(out + f * ppf + p, px3{o[0], o[1], o[2]});
}

// Colour-match chains on fp32 frames (no stencil behind them): FOUR pixels per thread, 1024 consecutive pixels per workgroup.  All
// four requests are issued before the arithmetic tables are staged and land under the staging and the first pixels' ~450
// instructions each; the table staging and its barrier are paid once per 1024 pixels instead of once per 256; the tail is handled
// without an exec-masked block (re-read pixel, store dropped by the buffer descriptor's range check).  Frames below 2 GiB.
template <int STAGES>
__global__ __launch_bounds__(256) void k_chain_pointwise4(const px3* __restrict__ in, px3* __restrict__ out, int32_t ppf, ChainK D) {
    const int64_t f = blockIdx.y;
    const px3* fin = in + f * ppf;
    const int32_t p0 = (int32_t)blockIdx.x * 1024 + (int32_t)threadIdx.x;
    px3 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int32_t p = p0 + 256 * j;
        v[j] = load_px_stream(fin + (p < ppf ? p : 0));
    }
    VRG_CM_MATH(PT, true, (STAGES & VRG_STAGE_FASTMATH) != 0, D.dm);
    typedef unsigned u3 __attribute__((ext_vector_type(3)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(out + f * ppf), 0, ppf * 12, 0x00020000);
    const FrameCtx FC = frame_ctx<STAGES>(D, f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int32_t p = p0 + 256 * j;
        const bool valid = p < ppf;
        const float x[3] = {v[j].r, v[j].g, v[j].b};
        float o[3];
        chain_pre<STAGES>(D, FC, valid ? p : 0, x, o, PT);
        __builtin_amdgcn_raw_buffer_store_b96(u3{__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2])}, rs,
                                              (int)(valid ? (uint32_t)p * 12u : 0x80000000u), 0, 0);
    }
}

// ----------------------------------------------------------------------------------------------
// Chain ending in a 3x3 stencil: 32x64-pixel tile per 256-thread block; the pre-stage result of the
// tile plus a one-pixel halo is staged in LDS as three planes (conflict-free row reads), then every
// thread produces 8 output pixels.  Halo recompute: (34*66)/(32*64) = 1.096x.
// ----------------------------------------------------------------------------------------------
#define VRG_TILE_H 32
constexpr int TILE_H = VRG_TILE_H, TILE_W = 64;
constexpr int TILE_ROWS = TILE_H / 4;      // rows per thread: 4 row groups of 64 columns in a 256-thread block
constexpr int HALO_H = TILE_H + 2, HALO_W = TILE_W + 2;
constexpr int LDS_PITCH = HALO_W + 1;

// XCD-aware work mapping: workgroup b is observed to run on XCD b % 8, each XCD has its own L2.  Work item
// (frame, tile) = (b % 8) * ceil(total/8) + b / 8 gives every XCD one contiguous run of tiles, so that the halo
// rows/columns a tile shares with its neighbours are re-read from the same L2 instead of from HBM (PMC: the
// naive mapping fetched 16.9 B/px for 13.2 B/px of tile+halo reads).  Placement only affects speed.
template <int STAGES, class IO>
__global__ __launch_bounds__(256) void k_chain_tile(const typename IO::elem* __restrict__ in, typename IO::elem* __restrict__ out,
                                                     int32_t H, int32_t W, int32_t tiles_x, int32_t tiles_per_frame, uint32_t total_work,
                                                     ChainK D) {
    __shared__ float tile[3][HALO_H][LDS_PITCH];
    VRG_CM_MATH(PT, (STAGES & VRG_STAGE_COLORMATCH) != 0, (STAGES & VRG_STAGE_FASTMATH) != 0, D.dm);
    const uint32_t per_xcd = (total_work + 7u) / 8u;
    const uint32_t work = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || work >= total_work) return;      // uniform per workgroup (after the barrier above)
    const uint32_t tile_id = work % (uint32_t)tiles_per_frame;
    const int32_t ty0 = (int32_t)(tile_id / (uint32_t)tiles_x) * TILE_H;
    const int32_t tx0 = (int32_t)(tile_id % (uint32_t)tiles_x) * TILE_W;
    const int64_t f = work / (uint32_t)tiles_per_frame;
    const int32_t ppf = H * W;
    const typename IO::elem* fin = in + f * ppf;
    const bool zero = D.zero_border != 0;
    const FrameCtx FC = frame_ctx<STAGES>(D, f);

#define VRG_TILE_PRELOAD 1
    constexpr int N_IT = (HALO_H * HALO_W + 255) / 256;
    if (VRG_TILE_PRELOAD && !(STAGES & VRG_STAGE_COLORMATCH) && !((STAGES & VRG_STAGE_GRAIN) && !(STAGES & VRG_STAGE_LUT))) {      // (grain -> sharpen alone measured 5 % slower with it)
        // light pre stages: all of this thread's tile + halo pixels are requested first, branch-free (coordinates clamped into the frame
        // -- that IS the replicate border -- and an index past the tile re-reads its first pixel), then processed: the requests of
        // the conditional loop below are issued and waited for one at a time, N_IT = 9 exposed latencies per thread.  Sharpen-only tile
        // 0.70 -> 0.53 ms per 16 x 4K frames, uint8 unsharp 245,000 -> 327,000 Mpixels/s, LUT (+ grain) + sharpen tiles 1.13x
        // (profiles/r03_tile_preload_ab.log)
        px3 v[N_IT];
        int32_t pp[N_IT];
#pragma unroll
        for (int k = 0; k < N_IT; ++k) {
            int i = (int)threadIdx.x + 256 * k;
            i = i < HALO_H * HALO_W ? i : 0;
            const int hy = i / HALO_W, hx = i - hy * HALO_W;
            int y = ty0 + hy - 1, x = tx0 + hx - 1;
            y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
            x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
            pp[k] = y * W + x;
            v[k] = IO::load(fin + pp[k]);
        }
#pragma unroll
        for (int k = 0; k < N_IT; ++k) {
            const int i = (int)threadIdx.x + 256 * k;
            if (i < HALO_H * HALO_W) {
                const int hy = i / HALO_W, hx = i - hy * HALO_W;
                const int y = ty0 + hy - 1, x = tx0 + hx - 1;
                const bool inside = (y >= 0) && (y < H) && (x >= 0) && (x < W);
                float o[3] = {0.0f, 0.0f, 0.0f};
                if (inside || !zero) {
                    const float xi[3] = {v[k].r, v[k].g, v[k].b};
                    chain_pre<STAGES>(D, FC, pp[k], xi, o, PT);
                }
                tile[0][hy][hx] = o[0];
                tile[1][hy][hx] = o[1];
                tile[2][hy][hx] = o[2];
            }
        }
    } else
    for (int i = threadIdx.x; i < HALO_H * HALO_W; i += 256) {
        const int hy = i / HALO_W, hx = i - hy * HALO_W;
        int y = ty0 + hy - 1, x = tx0 + hx - 1;
        const bool inside = (y >= 0) && (y < H) && (x >= 0) && (x < W);
        float o[3] = {0.0f, 0.0f, 0.0f};
        // pixels right/below the frame that only pad the last tiles are never read by a valid output
        // except through the border rule, which is coordinate clamping (replicate) or zero.
        if (inside || !zero) {
            y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
            x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
            const int32_t p = y * W + x;
            const px3 v = IO::load(fin + p);   // plain load: halo pixels are re-read by the neighbouring tiles (L2 hits)
            const float xi[3] = {v.r, v.g, v.b};
            chain_pre<STAGES>(D, FC, p, xi, o, PT);
        }
        tile[0][hy][hx] = o[0];
        tile[1][hy][hx] = o[1];
        tile[2][hy][hx] = o[2];
    }
    __syncthreads();
    // Each thread owns one column of the tile and 8 consecutive rows: adjacent lanes are adjacent pixels (coalesced
    // streaming stores, conflict-free LDS rows), the 3x3 window slides down in registers (30 LDS reads per channel for
    // 8 pixels instead of 72) and the index arithmetic is paid once per thread.
    typename IO::elem* fout = out + f * ppf;
    const int lx = threadIdx.x & (TILE_W - 1), ly0 = (threadIdx.x / TILE_W) * TILE_ROWS;
    const int x = tx0 + lx;
    if (x >= W) return;
    float res[TILE_ROWS][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float p[3][3];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) p[dy + 1][dx] = tile[c][ly0 + dy][lx + dx];
#pragma unroll
        for (int k = 0; k < TILE_ROWS; ++k) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                p[0][dx] = p[1][dx];
                p[1][dx] = p[2][dx];
                p[2][dx] = tile[c][ly0 + k + 2][lx + dx];
            }
            res[k][c] = stencil_value(D.stencil_op, p, D.strength, D.zero_border);
        }
    }
#pragma unroll
    for (int k = 0; k < TILE_ROWS; ++k) {
        const int y = ty0 + ly0 + k;
        if (y < H) IO::store_stream(fout + (y * W + x), px3{res[k][0], res[k][1], res[k][2]});
    }
}

// ----------------------------------------------------------------------------------------------
// Lab statistics.  Per (frame, channel): n, mean, M2 in fp64.  Stage 1: each block reduces a
// contiguous slice of the frame's pixels to six fp64 sums of (lab - pivot), (lab - pivot)^2, where
// pivot = Lab of the frame's first pixel (kills the cancellation of near-constant frames);
// wave64 DPP-free shuffle tree -> LDS -> one partial per block.  Stage 2: one wave per frame adds
// the partials in a fixed order.  No atomics: the result is deterministic and independent of how
// many frames a call covers (the block count depends on the frame size only).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

template <int STAGES, bool STATS = true>
__global__ __launch_bounds__(256) void k_lab_partials(const px3* __restrict__ in, int32_t ppf, int32_t bpf, ChainK D,
                                                       double* __restrict__ partials, px3* __restrict__ lab_out) {
    __shared__ double red[STATS ? 4 : 1][6];
    VRG_CM_MATH(PT, true, (STAGES & VRG_STAGE_FASTMATH) != 0, D.dm);
    const int64_t f = blockIdx.y;
    const px3* fin = in + f * ppf;
    const FrameCtx FC = frame_ctx<STAGES>(D, f);
    float pivot[3];
    {
        const px3 v0 = fin[0];
        const float x0[3] = {v0.r, v0.g, v0.b};
        float pre[3];
        chain_pre<STAGES>(D, FC, 0, x0, pre, PT);
        rgb_to_lab(pre, pivot, PT);
    }
    const int32_t per = (ppf + bpf - 1) / bpf;
    const int32_t lo = blockIdx.x * per;
    const int32_t hi = lo + per < ppf ? lo + per : ppf;
    double s1[3] = {0.0, 0.0, 0.0}, s2[3] = {0.0, 0.0, 0.0};
    for (int32_t p = lo + threadIdx.x; p < hi; p += 256) {
        const px3 v = load_px_stream(fin + p);
        const float x[3] = {v.r, v.g, v.b};
        float pre[3], lab[3];
        chain_pre<STAGES>(D, FC, p, x, pre, PT);
        rgb_to_lab(pre, lab, PT);
        if (lab_out) store_px_stream(lab_out + f * ppf + p, px3{lab[0], lab[1], lab[2]});
#pragma unroll
        for (int c = 0; c < (STATS ? 3 : 0); ++c) {
            const double d = (double)lab[c] - (double)pivot[c];
            s1[c] += d;
            s2[c] += d * d;
        }
    }
    if (!STATS) return;                 // Lab image only (statistics: vrg_torch_stats.hip)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double a = wave_sum(s1[c]);
        const double b = wave_sum(s2[c]);
        if (lane == 0) { red[wave][c] = a; red[wave][3 + c] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const double t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        partials[(f * bpf + blockIdx.x) * 6 + threadIdx.x] = t;
    }
}

template <int STAGES>
__global__ __launch_bounds__(64) void k_lab_merge(const px3* __restrict__ in, int32_t ppf, int32_t bpf, ChainK D,
                                                   const double* __restrict__ partials, double* __restrict__ stats) {
    VRG_CM_MATH(PT, true, (STAGES & VRG_STAGE_FASTMATH) != 0, D.dm);
    const int64_t f = blockIdx.x;
    float pivot[3];
    {
        const px3 v0 = in[f * ppf];
        const float x0[3] = {v0.r, v0.g, v0.b};
        float pre[3];
        chain_pre<STAGES>(D, frame_ctx<STAGES>(D, f), 0, x0, pre, PT);
        rgb_to_lab(pre, pivot, PT);
    }
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        double s1 = 0.0, s2 = 0.0;
        for (int b = 0; b < bpf; ++b) {
            s1 += partials[(f * bpf + b) * 6 + c];
            s2 += partials[(f * bpf + b) * 6 + 3 + c];
        }
        const double n = (double)ppf;
        const double dm = s1 / n;
        double m2 = s2 - s1 * dm;
        if (m2 < 0.0) m2 = 0.0;
        stats[(f * 3 + c) * 3 + 0] = n;
        stats[(f * 3 + c) * 3 + 1] = (double)pivot[c] + dm;
        stats[(f * 3 + c) * 3 + 2] = m2;
    }
}

// {n, mean, M2} fp64 -> {mean, std_unbiased + 1e-5} fp32 (nodes.py:99-100: .std() is unbiased; n == 1 -> NaN like torch)
__global__ void k_stats_finalize(const double* __restrict__ stats, float* __restrict__ ms, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const double n = stats[i * 3], mean = stats[i * 3 + 1], m2 = stats[i * 3 + 2];
    const float sd = (float)__builtin_sqrt(m2 / (n - 1.0));
    ms[i * 2] = (float)mean;
    ms[i * 2 + 1] = sd + 1e-5f;
}

template <int STAGES>
static int launch_stats(const float* in, int64_t frames, int32_t H, int32_t W, const ChainK& D, double* stats, void* scratch,
                        hipStream_t st, float* lab_out = nullptr) {
    const int64_t ppf = (int64_t)H * W;
    const int bpf = stats_blocks_per_frame(ppf);
    double* partials = reinterpret_cast<double*>(scratch);
    for (int64_t f0 = 0; f0 < frames; f0 += 32768) {
        const int64_t nf = frames - f0 < 32768 ? frames - f0 : 32768;
        ChainK d = D;
        if (STAGES & VRG_STAGE_GRAIN) {
            if (f0 % D.noise.chunk_frames) return VRG_ERR_UNSUPPORTED;
            d.noise.chunk0 += f0 / D.noise.chunk_frames;
        }
        const px3* src = reinterpret_cast<const px3*>(in) + f0 * ppf;
        if (!stats) {                   // Lab image only
            hipLaunchKernelGGL((k_lab_partials<STAGES, false>), dim3((uint32_t)bpf, (uint32_t)nf), dim3(256), 0, st, src, (int32_t)ppf, bpf, d,
                               (double*)nullptr, reinterpret_cast<px3*>(lab_out) + f0 * ppf);
            if (hipGetLastError() != hipSuccess) return VRG_ERR_LAUNCH;
            continue;
        }
        hipLaunchKernelGGL((k_lab_partials<STAGES>), dim3((uint32_t)bpf, (uint32_t)nf), dim3(256), 0, st, src, (int32_t)ppf, bpf, d,
                           partials + f0 * bpf * 6, lab_out ? reinterpret_cast<px3*>(lab_out) + f0 * ppf : nullptr);
        hipLaunchKernelGGL(k_lab_merge<STAGES>, dim3((uint32_t)nf), dim3(64), 0, st, src, (int32_t)ppf, bpf, d,
                           partials + f0 * bpf * 6, stats + f0 * 9);
        if (hipGetLastError() != hipSuccess) return VRG_ERR_LAUNCH;
    }
    return VRG_OK;
}

// (a function template of its own so that k_chain_pointwise4 is only instantiated for fp32 frames and colour-match chains)
template <int STAGES, class IO>
static typename std::enable_if<std::is_same<IO, IoF32>::value && (STAGES & VRG_STAGE_COLORMATCH) && !(STAGES & VRG_STAGE_GRAIN)>::type
launch_pointwise4(const px3* src, px3* dst, int64_t ppf, int64_t nf, const ChainK& d, hipStream_t st) {
    hipLaunchKernelGGL((k_chain_pointwise4<STAGES>), dim3((uint32_t)((ppf + 1023) / 1024), (uint32_t)nf), dim3(256), 0, st, src, dst, (int32_t)ppf, d);
}
template <int STAGES, class IO>
static typename std::enable_if<!(std::is_same<IO, IoF32>::value && (STAGES & VRG_STAGE_COLORMATCH) && !(STAGES & VRG_STAGE_GRAIN))>::type
launch_pointwise4(const typename IO::elem*, typename IO::elem*, int64_t, int64_t, const ChainK&, hipStream_t) {}

template <int STAGES, class IO = IoF32>
static int launch_chain(const void* in, void* out, int64_t frames, int32_t H, int32_t W, const ChainK& D, bool sharpen,
                        hipStream_t st) {
    const int64_t ppf = (int64_t)H * W;
    const int tx = (W + TILE_W - 1) / TILE_W, ty = (H + TILE_H - 1) / TILE_H;
    const int64_t tpf = (int64_t)tx * ty;
    // frames per launch: 2-D grid y limit for the point-wise kernel, 32-bit work-item count for the tile kernel
    int64_t step = 32768;
    if (sharpen) {
        step = (1ll << 23) / tpf;
        if (step < 1) return VRG_ERR_UNSUPPORTED;
    }
    if (STAGES & VRG_STAGE_GRAIN) {
        if (step < D.noise.chunk_frames) return VRG_ERR_UNSUPPORTED;
        step -= step % D.noise.chunk_frames;
    }
    if ((STAGES & VRG_STAGE_COLORMATCH) && D.cm.ref_frames != 1) {
        if (step < D.cm.ref_frames) return VRG_ERR_UNSUPPORTED;
        step -= step % D.cm.ref_frames;     // (both alignments together are only needed when ref_frames divides chunk_frames)
    }
    for (int64_t f0 = 0; f0 < frames; f0 += step) {
        const int64_t nf = frames - f0 < step ? frames - f0 : step;
        ChainK d = D;
        if (STAGES & VRG_STAGE_GRAIN) {
            if (f0 % D.noise.chunk_frames) return VRG_ERR_UNSUPPORTED;
            d.noise.chunk0 += f0 / D.noise.chunk_frames;
        }
        if (STAGES & VRG_STAGE_COLORMATCH) {
            if (D.cm.ref_frames != 1 && (f0 % D.cm.ref_frames)) return VRG_ERR_UNSUPPORTED;
            d.cm.img_ms += f0 * 6;
        }
        const typename IO::elem* src = reinterpret_cast<const typename IO::elem*>(in) + f0 * ppf;
        typename IO::elem* dst = reinterpret_cast<typename IO::elem*>(out) + f0 * ppf;
        if (sharpen) {
            const uint32_t total = (uint32_t)(tpf * nf);
            const uint32_t blocks = ((total + 7u) / 8u) * 8u;
            hipLaunchKernelGGL((k_chain_tile<STAGES, IO>), dim3(blocks), dim3(256), 0, st, src, dst, H, W, tx, (int32_t)tpf, total, d);
        } else if ((STAGES & VRG_STAGE_COLORMATCH) && !(STAGES & VRG_STAGE_GRAIN) && std::is_same<IO, IoF32>::value && ppf * 12 < ((int64_t)1 << 31)) {
            launch_pointwise4<STAGES, IO>(src, dst, ppf, nf, d, st);
        } else {
            hipLaunchKernelGGL((k_chain_pointwise<STAGES, IO>), dim3((uint32_t)((ppf + 255) / 256), (uint32_t)nf), dim3(256), 0, st,
                               src, dst, (int32_t)ppf, d);
        }
        if (hipGetLastError() != hipSuccess) return VRG_ERR_LAUNCH;
    }
    return VRG_OK;
}

int fill_chain(const vrg_chain_desc* d, int32_t H, int32_t W, ChainK& D) {
    D = ChainK{};
    D.stages = d->stages;
    D.dm = host_dev_math();
    if (d->stages & VRG_STAGE_GRAIN) {
        if (d->noise.chunk_frames < 1 || d->noise.grid_threads == 0 || (d->noise.grid_threads % 256u)) return VRG_ERR_BAD_ARG;
        if ((int64_t)d->noise.chunk_frames * H * W * 3 > 0x7fffffffll) return VRG_ERR_UNSUPPORTED;
        D.I = d->intensity; D.S = d->sat; D.T = d->one_minus_sat;
        D.noise = make_noise(&d->noise, (int64_t)H * W * 3);
    }
    if (d->stages & VRG_STAGE_LUT) {
        if (!d->lut || d->lut_size < 2 || (d->blend_mode != 1 && d->blend_mode != 2)) return VRG_ERR_BAD_ARG;
        D.lut = make_lut(d->lut, d->lut_size, d->domain_min, d->domain_max, d->blend_mode, d->blend, d->one_minus_blend);
    }
    if (d->stages & VRG_STAGE_COLORMATCH) {
        D.cm = CmK{d->img_ms, d->ref_ms, d->ref_frames, d->k, d->one_minus_k};
    }
    if (d->stages & VRG_STAGE_SHARPEN) {
        if (d->stencil_op < 0 || d->stencil_op > 2 || d->border < 0 || d->border > 1) return VRG_ERR_BAD_ARG;
        D.stencil_op = d->stencil_op; D.zero_border = d->border == VRG_BORDER_ZERO; D.strength = d->strength;
    }
    return VRG_OK;
}

int launch_march(const float* in, float* out, int64_t frames, int32_t H, int32_t W, const ChainK& D0, int stages, hipStream_t st);
bool apply_march_applicable(int stages, int32_t H, int32_t W);                                   // vrg_apply_march.hip
int launch_apply_march(const float* in, float* out, int64_t frames, int32_t H, int32_t W, const ChainK& D, int stages, bool fast, hipStream_t st);
bool produce_applicable(int stages, int64_t frame_elems);
int launch_grain_u8(const uint8_t* in, uint8_t* out, int64_t frames, int32_t height, int32_t width, float intensity, float sat,
                    float one_minus_sat, const vrg_noise_desc* nd, void* stream);      // vrg_pointwise.hip
bool lut_lds_applicable(int lut_size, int64_t pixels);
int launch_lut_lds(const void* in, void* out, int64_t pixels, const LutParams& P, bool u8, hipStream_t st);
int64_t produce_scratch_bytes(const ChainK& D, int64_t frames, int64_t fe);
int launch_produce(const float* in, float* lab_out, int64_t frames, int32_t H, int32_t W, const ChainK& D, int stages, double* stats,
                   void* scratch, hipStream_t st);

}  // namespace vrg

using namespace vrg;

extern "C" {

int64_t vrg_lab_stats_scratch_bytes(int64_t frames) { return frames < 0 ? 0 : frames * STATS_BPF_MAX * 6 * (int64_t)sizeof(double); }

int64_t vrg_chain_stats_scratch_bytes(int64_t frames, int32_t height, int32_t width, const vrg_chain_desc* desc) {
    if (frames < 0 || height <= 0 || width <= 0 || !desc) return 0;
    int64_t need = vrg_lab_stats_scratch_bytes(frames);
    const int64_t fe = (int64_t)height * width * 3;
    vrg_chain_desc pre = *desc;
    pre.stages &= (VRG_STAGE_GRAIN | VRG_STAGE_LUT);
    ChainK D;
    if (produce_applicable(pre.stages, fe) && fill_chain(&pre, height, width, D) == VRG_OK && frames % D.noise.chunk_frames == 0) {
        const int64_t p = produce_scratch_bytes(D, frames, fe);
        if (p > need) need = p;
    }
    return need;
}

int vrg_lab_stats_f32(const float* in, int64_t frames, int32_t height, int32_t width, double* stats, void* scratch, int32_t cm_math,
                      void* stream) {
    if (!in || !stats || !scratch || frames < 0 || height <= 0 || width <= 0 || (cm_math != VRG_CM_MATH_DEVICE && cm_math != VRG_CM_MATH_FAST))
        return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    if ((int64_t)height * width > 0x7fffffff) return VRG_ERR_UNSUPPORTED;
    ChainK D{};
    D.dm = host_dev_math();
    if (cm_math == VRG_CM_MATH_FAST) return launch_stats<VRG_STAGE_FASTMATH>(in, frames, height, width, D, stats, scratch, (hipStream_t)stream);
    return launch_stats<0>(in, frames, height, width, D, stats, scratch, (hipStream_t)stream);
}

int vrg_lab_stats_finalize(const double* stats, float* mean_std, int64_t frames, void* stream) {
    if (!stats || !mean_std || frames < 0) return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    const int64_t count = frames * 3;
    hipLaunchKernelGGL(k_stats_finalize, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats, mean_std,
                       count);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

int vrg_chain_stats_f32(const float* in, int64_t frames, int32_t height, int32_t width, const vrg_chain_desc* desc, double* stats,
                        void* scratch, void* stream) {
    return vrg_chain_stats_lab_f32(in, nullptr, frames, height, width, desc, stats, scratch, stream);
}

int vrg_chain_stats_lab_f32(const float* in, float* lab_out, int64_t frames, int32_t height, int32_t width,
                            const vrg_chain_desc* desc, double* stats, void* scratch, void* stream) {
    // stats == NULL (with lab_out): the Lab image only -- no statistics, no scratch
    if (!in || !desc || (!stats && !lab_out) || (stats && !scratch) || frames < 0 || height <= 0 || width <= 0) return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    if ((int64_t)height * width > 0x7fffffff / 3) return VRG_ERR_UNSUPPORTED;
    vrg_chain_desc pre = *desc;
    pre.stages &= (VRG_STAGE_GRAIN | VRG_STAGE_LUT);   // statistics are taken on the colour-match *input*
    ChainK D;
    const int rc = fill_chain(&pre, height, width, D);
    if (rc) return rc;
    if (desc->cm_math != VRG_CM_MATH_DEVICE && desc->cm_math != VRG_CM_MATH_FAST) return VRG_ERR_BAD_ARG;
    const bool fast = desc->cm_math == VRG_CM_MATH_FAST;
    // chains that start with grain: shared-Philox pass (vrg_produce.hip) unless the A/B knob 0x200 asks for the
    // general kernel; tiny frames always take the general kernel
    if (!(desc->variant & 0x200) && produce_applicable(pre.stages, (int64_t)height * width * 3) && frames % D.noise.chunk_frames == 0)
        return launch_produce(in, lab_out, frames, height, width, D, pre.stages | (fast ? VRG_STAGE_FASTMATH : 0), stats, scratch,
                              (hipStream_t)stream);
#define CALL(S) launch_stats<S>(in, frames, height, width, D, stats, scratch, (hipStream_t)stream, lab_out)
    switch ((pre.stages & 3) | (fast ? 4 : 0)) {
        case 0: return CALL(0);
        case 1: return CALL(1);
        case 2: return CALL(2);
        case 3: return CALL(3);
        case 4: return CALL(0 | VRG_STAGE_FASTMATH);
        case 5: return CALL(1 | VRG_STAGE_FASTMATH);
        case 6: return CALL(2 | VRG_STAGE_FASTMATH);
        default: return CALL(3 | VRG_STAGE_FASTMATH);
    }
#undef CALL
}

int vrg_fused_chain_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width, const vrg_chain_desc* desc,
                        void* stream) {
    if (!in || !out || !desc || frames < 0 || height <= 0 || width <= 0) return VRG_ERR_BAD_ARG;
    if (desc->stages == 0 || (desc->stages & ~31)) return VRG_ERR_BAD_ARG;
    if ((desc->stages & VRG_STAGE_FROM_LAB) && (desc->stages & 7) != VRG_STAGE_COLORMATCH) return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    if ((int64_t)height * width > 0x7fffffff / 3) return VRG_ERR_UNSUPPORTED;
    if ((desc->stages & VRG_STAGE_COLORMATCH) && (!desc->img_ms || !desc->ref_ms || desc->ref_frames < 1)) return VRG_ERR_BAD_ARG;
    if ((desc->variant & 0xff) > 2 || (desc->variant & ~0x2ff)) return VRG_ERR_UNSUPPORTED;
    if (desc->cm_math != VRG_CM_MATH_DEVICE && desc->cm_math != VRG_CM_MATH_FAST) return VRG_ERR_BAD_ARG;
    ChainK D;
    const int rc = fill_chain(desc, height, width, D);
    if (rc) return rc;
    const bool sharpen = (desc->stages & VRG_STAGE_SHARPEN) != 0;
    // variant 2: register-resident wave march (vrg_march.hip); variant 1: LDS tile / point-wise kernels below.
    // variant 0 picks by measurement (profiles/r01_diag_kernels.json): grain -> (LUT) -> sharpen chains are ALU
    // bound and the march wins (Philox shared across four strips, no LDS); point-wise chains, chains without
    // grain and the colour-match apply pass run faster on the higher-occupancy tile / point-wise kernels.
    int variant = desc->variant & 0xff;
    if (variant == 0 && desc->stages == VRG_STAGE_GRAIN)      // grain alone: the shared-Philox grain kernel (one Philox call per four elements)
        return vrg_grain_f32(in, out, frames, height, width, desc->intensity, desc->sat, desc->one_minus_sat, &desc->noise, stream);
    if (variant == 0) {
        variant = ((desc->stages & VRG_STAGE_GRAIN) && (desc->stages & VRG_STAGE_SHARPEN) && !(desc->stages & VRG_STAGE_COLORMATCH)) ? 2 : 1;
        // a cube of at most 21^3 lives in LDS inside the march kernel: no gather path, so it also wins for LUT -> sharpen
        // (126 vs 66 Gpix/s with a 17^3 cube) and grain -> LUT (118 vs 70); LUT-only chains take k_lut3d_lds below
        if ((desc->stages & VRG_STAGE_LUT) && !(desc->stages & VRG_STAGE_COLORMATCH) && desc->stages != VRG_STAGE_LUT &&
            lut_lds_applicable(desc->lut_size, frames * (int64_t)height * width) && frames * (int64_t)height * width >= 24000000ll)   // enough strips for 12-wave workgroups on every CU
            variant = 2;
        // grain -> LUT over a global table (no stencil): since the march's pixel rows are non-temporal accesses (round 5) its steady-row body
        // beats the point-wise kernel -- one Philox call per four elements instead of one per element, quad-cooperative gathers -- once the
        // launch holds a few rounds of waves: 128 x 1080p 78 / 95 against 76 / 89 Gpix/s (uniform / video-like), 64 x 4K 83 / 100 against 76 /
        // 89, with a 25^3 cube 82 / 111 against 77 / 91; below ~10,000 strip jobs (3 waves per slot of the chip) it loses: 96 x 720p 59
        // against 75, 8 x 1080p 48 against 67 (profiles/r05_bench_flat_march.json, r05_bench_flat_march_sizes.json)
        if (variant == 1 && desc->stages == (VRG_STAGE_GRAIN | VRG_STAGE_LUT) && desc->noise.chunk_frames > 0 && desc->noise.grid_threads > 0 &&
            frames % desc->noise.chunk_frames == 0) {
            const int64_t chunk_elems = (int64_t)desc->noise.chunk_frames * height * width * 3;
            const int64_t four_g = 4ll * desc->noise.grid_threads;
            const int64_t jobs = (frames / desc->noise.chunk_frames) * ((chunk_elems + four_g - 1) / four_g) * ((width + 62) / 63);
            if (jobs >= 10000) variant = 2;
        }
    }
    if (variant == 2) {
        // the march kernel has no colour-match stage and addresses a chunk with 32-bit element offsets (<= 0x60000000
        // elements): what it cannot take goes to the tile / point-wise kernels below (any variant: same results)
        const int rc = launch_march(in, out, frames, height, width, D, desc->stages, (hipStream_t)stream);
        if (rc != VRG_ERR_UNSUPPORTED) return rc;
    }
    // (LUT ->) colour match -> stencil without a grain stage -- pass 2 of the headline chain: the apply march (variant 0 and 2; 1 forces
    // the LDS-tile kernel for A/B and cross-checks).  Frames whose statistics groups it cannot keep together fall through.
    if ((desc->variant & 0xff) != 1 && apply_march_applicable(desc->stages, height, width)) {
        const int rc = launch_apply_march(in, out, frames, height, width, D, desc->stages, desc->cm_math == VRG_CM_MATH_FAST, (hipStream_t)stream);
        if (rc != VRG_ERR_UNSUPPORTED) return rc;
    }
    if ((desc->variant & 0xff) == 0 && desc->stages == VRG_STAGE_LUT && lut_lds_applicable(desc->lut_size, frames * height * width))
        return launch_lut_lds(in, out, frames * (int64_t)height * width, D.lut, false, (hipStream_t)stream);   // small cube: table in LDS
#define CALL(S) launch_chain<S>(in, out, frames, height, width, D, sharpen, (hipStream_t)stream)
    if (!(desc->stages & VRG_STAGE_COLORMATCH)) {
        switch (desc->stages & 3) {
            case 0: return CALL(0);
            case 1: return CALL(1);
            case 2: return CALL(2);
            default: return CALL(3);
        }
    }
    constexpr int CM = VRG_STAGE_COLORMATCH, FM = VRG_STAGE_FASTMATH;
    if (desc->cm_math == VRG_CM_MATH_FAST) {
        if (desc->stages & VRG_STAGE_FROM_LAB) return CALL(CM | VRG_STAGE_FROM_LAB | FM);
        switch (desc->stages & 3) {
            case 0: return CALL(CM | FM);
            case 1: return CALL(CM | 1 | FM);
            case 2: return CALL(CM | 2 | FM);
            default: return CALL(CM | 3 | FM);
        }
    }
    if (desc->stages & VRG_STAGE_FROM_LAB) return CALL(CM | VRG_STAGE_FROM_LAB);
    switch (desc->stages & 3) {
        case 0: return CALL(CM);
        case 1: return CALL(CM | 1);
        case 2: return CALL(CM | 2);
        default: return CALL(CM | 3);
    }
#undef CALL
}

// uint8 BGR frames in and out (video routes): grain / LUT / 3x3 sharpen in any combination, no colour match (the
// routes have none; its statistics pass would need the fp32 image anyway).  LDS-tile and point-wise kernels only.
int vrg_fused_chain_u8(const uint8_t* in, uint8_t* out, int64_t frames, int32_t height, int32_t width, const vrg_chain_desc* desc,
                       void* stream) {
    if (!in || !out || !desc || frames < 0 || height <= 0 || width <= 0) return VRG_ERR_BAD_ARG;
    if (desc->stages == 0 || (desc->stages & ~(VRG_STAGE_GRAIN | VRG_STAGE_LUT | VRG_STAGE_SHARPEN))) return VRG_ERR_UNSUPPORTED;
    if (frames == 0) return VRG_OK;
    if ((int64_t)height * width > 0x7fffffff / 3) return VRG_ERR_UNSUPPORTED;
    ChainK D;
    const int rc = fill_chain(desc, height, width, D);
    if (rc) return rc;
    const bool sharpen = (desc->stages & VRG_STAGE_SHARPEN) != 0;
    hipStream_t st = (hipStream_t)stream;
    if (desc->stages == VRG_STAGE_GRAIN && (desc->variant & 0xff) == 0)      // grain alone: the shared-Philox grain kernel on uint8 frames
        return launch_grain_u8(in, out, frames, height, width, desc->intensity, desc->sat, desc->one_minus_sat, &desc->noise, stream);
    if (desc->stages == VRG_STAGE_LUT && lut_lds_applicable(desc->lut_size, frames * height * width))
        return launch_lut_lds(in, out, frames * (int64_t)height * width, D.lut, true, st);                      // small cube: table in LDS
    switch (desc->stages & 3) {
        case 0: return launch_chain<0, IoU8>(in, out, frames, height, width, D, sharpen, st);
        case 1: return launch_chain<1, IoU8>(in, out, frames, height, width, D, sharpen, st);
        case 2: return launch_chain<2, IoU8>(in, out, frames, height, width, D, sharpen, st);
        default: return launch_chain<3, IoU8>(in, out, frames, height, width, D, sharpen, st);
    }
}

}  // extern "C"
