// vrg_produce_body.hpp -- the workgroup body of colour-match pass 1 for chains that start with grain (see vrg_produce.hip for the
// design), as a device function (kept separate from the launcher so that other kernels can run it as a workgroup role).
#pragma once
#include "vrg_chain_stages.hpp"

namespace vrg {

constexpr int PR_SUB = 768;      // subsequences per sub-range = 256 pixels per sibling
#define VRG_PR_SUBS 8
constexpr int PR_SUBS = VRG_PR_SUBS;       // sub-ranges per block
constexpr int PR_RUN = PR_SUB * PR_SUBS;

struct ProduceK {
    int32_t numel;        // chunk elements
    int32_t fe;           // frame elements (H*W*3)
    int32_t chunk_frames;
    uint32_t G, K, NB;    // subsequences, call indices per chunk, blocks per (chunk, k)
    uint32_t chunks;
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// does sibling m of block (k, ib) own pixels of two different frames?  (element indices of a chunk fit 31 bits: numel is
// an int32 -- 32-bit divisions, a dozen of them per workgroup)
__device__ __forceinline__ bool run_crosses_frame(const ProduceK& P, int64_t A) {
    if (A >= P.numel) return false;
    int64_t last = A + PR_RUN - 1;
    if (last > P.numel - 1) last = P.numel - 1;
    return ((uint32_t)A / (uint32_t)P.fe) != ((uint32_t)last / (uint32_t)P.fe);
}

#define VRG_PRODUCE_WAVES 4   /* device-policy kernel: 130 -> 127 VGPRs, 4 waves per SIMD, statistics pass -7 % (A/B) */
#define VRG_PRODUCE_WAVES_LABONLY 4
// STATS = false: the Lab image only (the statistics are then torch's own reductions over that image, vrg_torch_stats.hip): no
// accumulators, no records, and a run that crosses a frame boundary needs no second instantiation.
inline void produce_geometry(const ChainK& D, int64_t frames, int64_t fe, ProduceK& P) {
    P.fe = (int32_t)fe;
    P.chunk_frames = D.noise.chunk_frames;
    P.numel = (int32_t)(fe * D.noise.chunk_frames);
    P.G = D.noise.G;
    P.K = (uint32_t)((P.numel + 4ll * P.G - 1) / (4ll * P.G));
    P.NB = (P.G + PR_RUN - 1) / PR_RUN;
    P.chunks = (uint32_t)(frames / D.noise.chunk_frames);
}

// The workgroup body of k_produce_lab, callable from other kernels: `bx` = the workgroup's
// index in the produce grid, `sn` / `red` = its LDS (4 x (PR_SUB + 4) floats of staged normals; STATS: 4 x 12 doubles), `PT` = the
// colour-match arithmetic object (its tables already staged in LDS by the caller, all threads synchronised).
template <int STAGES, bool TWO_PART, bool STATS, class MATH>
__device__ __forceinline__ void produce_lab_body(uint32_t bx, const float* __restrict__ in, float* __restrict__ lab_out, const ProduceK& P, const ChainK& D,
                                                 const float* __restrict__ pivots, double* __restrict__ rec, int32_t* __restrict__ rec_frame,
                                                 float (*sn)[PR_SUB + 4], double (*red)[12], const MATH& PT) {
    const uint32_t per_chunk = P.K * P.NB;
    const uint32_t G = P.G;
    uint32_t chunk, k, ib;
    if (!TWO_PART) {
        // one workgroup per (chunk, k, run); runs that cross a frame boundary are left to the TWO_PART launch
        chunk = bx / per_chunk;
        const uint32_t rem = bx - chunk * per_chunk;
        k = rem / P.NB;
        ib = rem - k * P.NB;
    } else {
        // one workgroup per (chunk, frame boundary b, sibling m): the run whose sibling m contains element b*fe;
        // a run crossed by several siblings is taken by the lowest such sibling only
        const uint32_t per = (uint32_t)(P.chunk_frames - 1) * 4u;
        chunk = bx / per;
        const uint32_t j = bx - chunk * per;
        const int64_t edge = (int64_t)(j / 4u + 1u) * P.fe;           // first element of frame b
        const uint32_t m_here = j & 3u;
        const int64_t q = (edge - 1) / G;                            // quarter of the last element of frame b-1
        if ((uint32_t)(q & 3) != m_here) return;
        k = (uint32_t)(q >> 2);
        ib = (uint32_t)(((edge - 1) - q * (int64_t)G) / PR_RUN);
    }
    const int64_t q0 = (int64_t)4 * G * k;
    const uint32_t run0 = ib * PR_RUN;                       // first subsequence of the block
    bool crosses = false;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const bool c = run_crosses_frame(P, q0 + (int64_t)G * m + run0);
        if (TWO_PART && c && !crosses && (uint32_t)m != (bx & 3u)) return;    // a lower sibling owns this run
        crosses = crosses || c;
    }
    if (STATS && crosses != TWO_PART) return;
    const uint32_t block_lin = chunk * per_chunk + k * P.NB + ib;    // record slot of the run

    const float* cin = in + (int64_t)chunk * P.numel;
    float* clab = lab_out ? lab_out + (int64_t)chunk * P.numel : nullptr;
    const uint64_t seed = chunk_seed(D.noise, chunk);
    const uint64_t off = chunk_offset(D.noise, chunk);
    const uint64_t ctr = (off >> 2) + k;
    const int tid = threadIdx.x;

    // frames of the four sibling runs and their pivots
    int fr[4];
    int64_t fb[4];                 // first element of the next frame (TWO_PART split point)
    float pv[4][2][3];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int64_t A = q0 + (int64_t)G * m + run0;
        fr[m] = A < P.numel ? (int)((uint32_t)A / (uint32_t)P.fe) : -1;
        fb[m] = ((int64_t)fr[m] + 1) * P.fe;
#pragma unroll
        for (int part = 0; part < (STATS ? (TWO_PART ? 2 : 1) : 0); ++part) {
            int f = fr[m] + part;
            f = f < 0 ? 0 : (f > P.chunk_frames - 1 ? P.chunk_frames - 1 : f);
            const float* pp = pivots + ((int64_t)chunk * P.chunk_frames + f) * 3;
            pv[m][part][0] = pp[0]; pv[m][part][1] = pp[1]; pv[m][part][2] = pp[2];
        }
    }
    const FrameCtx FC0{};            // no colour-match stage in front of the Lab transform, and the normals come from the shared draws
    double s1[4][TWO_PART ? 2 : 1][3], s2[4][TWO_PART ? 2 : 1][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int part = 0; part < (TWO_PART ? 2 : 1); ++part)
#pragma unroll
            for (int c = 0; c < 3; ++c) { s1[m][part][c] = 0.0; s2[m][part][c] = 0.0; }

#define VRG_PR_PIPE 1     /* Lab-only form: 1 = pixels requested before the noise synthesis, 2 = and the LUT gathers pipelined, 3 = two siblings before, two after the barrier */
    static_assert((STAGES & VRG_STAGE_GRAIN) != 0, "pass 1 of chains that START WITH GRAIN: its Lab transform relies on grain's clamp (rgb_to_lab_unit)");
    constexpr bool PIPE = VRG_PR_PIPE && !STATS && (STAGES & VRG_STAGE_GRAIN) && !(STAGES & VRG_STAGE_COLORMATCH);
    constexpr bool HAS_LUT = (STAGES & VRG_STAGE_LUT) != 0;
    for (int sub = 0; sub < PR_SUBS; ++sub) {
        const uint32_t I = run0 + (uint32_t)sub * PR_SUB;            // first subsequence of this sub-range
        if (I >= G) break;                                           // uniform
        const uint32_t valid_n = (G - I) < (uint32_t)PR_SUB ? (G - I) : (uint32_t)PR_SUB;
        // PIPE: the four siblings' pixels are requested BEFORE the noise synthesis (branch-free: a lane without a pixel re-reads the
        // chunk's first one), so that they arrive under its ~550 instructions instead of being waited for after the barrier
        px3 pre_px[4];
        auto place = [&](int m, int& p0, int64_t& e0) {              // this thread's pixel of sibling m: position in the sub-range, element, has one?
            const int64_t a = q0 + (int64_t)G * m + I;               // first element of the sibling's sub-range (uniform)
            const int shift = (int)((3u - (uint32_t)a % 3u) % 3u);   // re-alignment to the pixel grid (chunk elements fit 31 bits)
            p0 = shift + 3 * tid;
            e0 = a + p0;
            return p0 < (int)valid_n && e0 + 2 < P.numel;
        };
        auto request = [&](int m) {
            int p0;
            int64_t e0;
            const bool ok = place(m, p0, e0);
            pre_px[m] = load_px_stream(reinterpret_cast<const px3*>(cin + (ok ? e0 : 0)));
        };
        if (PIPE) {
#pragma unroll
            for (int m = 0; m < (VRG_PR_PIPE == 3 ? 2 : 4); ++m) request(m);
        }
        // ---- noise of the sub-range into LDS: 3 Philox calls per thread, 12 normals
        {
            float nz[3][4];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32x4 r = philox_for(seed, I + 3u * tid + j, ctr);
                const f32x2 a = box_muller(r.x, r.y);
                const f32x2 b = box_muller(r.z, r.w);
                nz[j][0] = a.x; nz[j][1] = a.y; nz[j][2] = b.x; nz[j][3] = b.y;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (3u * tid + j < valid_n) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) sn[m][3 * tid + j] = nz[j][m];
                }
            if (tid < 8) {                                           // the two elements past the sub-range, per sibling
                const int m = tid >> 1, h = tid & 1;
                const int64_t li = q0 + (int64_t)G * m + I + valid_n + h;
                sn[m][valid_n + h] = (li < P.numel) ? torch_randn_element(seed, off, G, (uint64_t)li) : 0.0f;
            }
        }
        __syncthreads();
        if (PIPE) {
            // ---- one pixel per thread and sibling, software-pipelined: the LUT gathers of sibling m + 1 are issued BEFORE the Lab
            // arithmetic of sibling m (six powers, ~400 instructions) and land under it; one LutFetch is live at a time (its registers
            // are free again once sibling m is interpolated), so the pipeline costs no occupancy.  Branch-free up to the store: lanes
            // without a pixel work on the re-read one and skip only the store.
            LutFetch F;
            float gr[3];
            auto issue = [&](int m) {
                const float x[3] = {pre_px[m].r, pre_px[m].g, pre_px[m].b};
                int p0;
                int64_t e0;
                if (!place(m, p0, e0)) p0 = 0;
                const float n[3] = {sn[m][p0], sn[m][p0 + 1], sn[m][p0 + 2]};
                grain_pixel_nan_branch(x, n, D.I, D.S, D.T, gr);
                if (HAS_LUT) lut_fetch_issue(D.lut, gr, F);
            };
            if (VRG_PR_PIPE == 3) { request(2); request(3); }         // land under the Lab arithmetic of siblings 0 and 1
            if (VRG_PR_PIPE == 2) issue(0);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float y[3], pre[3], lab[3];
                if (VRG_PR_PIPE != 2) issue(m);
                if (HAS_LUT) {
                    lut_fetch_finish(F, y);
                    lut_blend(D.lut, gr, y, pre);
                } else {
                    pre[0] = gr[0]; pre[1] = gr[1]; pre[2] = gr[2];
                }
                if (VRG_PR_PIPE == 2 && m < 3) issue(m + 1);
                rgb_to_lab_unit(pre, lab, PT);      // pre left grain's / the cube's clamp
                int p0;
                int64_t e0;
                if (place(m, p0, e0) && clab) store_px_stream(reinterpret_cast<px3*>(clab + e0), px3{lab[0], lab[1], lab[2]});
            }
        } else
        // ---- one pixel per thread and sibling
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int64_t a = q0 + (int64_t)G * m + I;               // first element of the sibling's sub-range
            const int shift = (int)((3 - a % 3) % 3);                // re-alignment to the pixel grid
            const int p0 = shift + 3 * tid;                          // position of the pixel's channel 0 in the sub-range
            const int64_t e0 = a + p0;
            if (p0 < (int)valid_n && e0 + 2 < P.numel) {
                const px3 v = load_px_stream(reinterpret_cast<const px3*>(cin + e0));
                const float x[3] = {v.r, v.g, v.b};
                const float n[3] = {sn[m][p0], sn[m][p0 + 1], sn[m][p0 + 2]};
                float pre[3], lab[3];
                chain_apply_stages<STAGES>(D, FC0, x, n, pre, PT);
                rgb_to_lab_unit(pre, lab, PT);      // pre left grain's / the cube's clamp
                if (clab) store_px_stream(reinterpret_cast<px3*>(clab + e0), px3{lab[0], lab[1], lab[2]});
                const int part = TWO_PART ? (e0 >= fb[m] ? 1 : 0) : 0;
#pragma unroll
                for (int c = 0; c < (STATS ? 3 : 0); ++c) {
                    if (TWO_PART) {
                        const double d0 = (double)lab[c] - (double)pv[m][0][c];
                        const double d1 = (double)lab[c] - (double)pv[m][1][c];
                        s1[m][0][c] += part == 0 ? d0 : 0.0;  s2[m][0][c] += part == 0 ? d0 * d0 : 0.0;
                        s1[m][1][c] += part == 1 ? d1 : 0.0;  s2[m][1][c] += part == 1 ? d1 * d1 : 0.0;
                    } else {
                        const double d = (double)lab[c] - (double)pv[m][0][c];
                        s1[m][0][c] += d;
                        s2[m][0][c] += d * d;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (!STATS) return;
    // ---- one record per (block, sibling, part)
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int part = 0; part < (TWO_PART ? 2 : 1); ++part) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double a = wave_sum_f64(s1[m][part][c]);
                const double b = wave_sum_f64(s2[m][part][c]);
                if (lane == 0) { red[wave][part * 6 + c] = a; red[wave][part * 6 + 3 + c] = b; }
            }
        }
        __syncthreads();
        const int64_t base = ((int64_t)block_lin * 4 + m) * 2;
        if (tid < (TWO_PART ? 12 : 6)) {
            const double t = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
            rec[(base + tid / 6) * 6 + tid % 6] = t;
        }
        if (tid == 0) {
            const int f0 = fr[m];
            rec_frame[base] = f0 >= 0 ? (int32_t)(chunk * P.chunk_frames + f0) : -1;
            rec_frame[base + 1] = (TWO_PART && f0 >= 0 && f0 + 1 < P.chunk_frames) ? (int32_t)(chunk * P.chunk_frames + f0 + 1) : -1;
        }
        __syncthreads();
    }
}


}  // namespace vrg
