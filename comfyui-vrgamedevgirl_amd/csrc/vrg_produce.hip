// vrg_produce.hip -- colour-match pass 1 for chains that start with grain: grain -> (LUT) -> Lab, with the Lab
// image stored and the per-frame statistics reduced, and with the Philox work shared.  gfx950 only.
//
// The general statistics kernel (k_lab_partials, vrg_chain.hip) draws each pixel's three normals with three
// Philox calls (one per element, one of four outputs used).  Here a block owns a run of Philox subsequences
// of one call index k -- SUBS sub-ranges of 768 subsequences -- and therefore the four element ranges
// {4Gk + G*m + idx}, m = 0..3, G apart: per sub-range every thread makes 3 Philox calls (12 normals) and
// processes 4 pixels, one per sibling range (768 elements = 256 pixels per range).  G is not a multiple of 3,
// so a sibling's pixels are shifted by 0..2 elements against the thread's subsequences: the normals go through
// LDS, and the two elements past the end of a sub-range come from the general per-element routine.
// A pixel belongs to the block that owns its channel-0 element, so every pixel is processed exactly once.
//
// Statistics: fp64 sums of (lab - pivot), (lab - pivot)^2 per sibling stay in registers across the SUBS
// sub-ranges and are reduced once per block (wave shuffles -> LDS -> one record per (block, sibling)); a merge
// kernel adds the records of each frame in a fixed order (deterministic, no atomics).  A sibling run that
// crosses a frame boundary needs two accumulator sets; those few blocks are handled by the TWO_PART
// instantiation in a second launch so that the common case keeps its registers.
#include "vrg_produce_body.hpp"

namespace vrg {

#define VRG_PRODUCE_MAX_WAVES_LABONLY 8   /* cap on the workgroups per CU (= waves per SIMD) of the Lab-only form; 8 = none.  Measured with LDS padding, 64 frames: 4 / 5 / 6 per CU = 8.02 / 7.64 / 7.54 ms (profiles/r04_ab_pass1_occupancy.json): the six that 80 VGPRs allow are the fastest */
template <int STAGES, bool TWO_PART, bool STATS = true>
__global__ __launch_bounds__(256, TWO_PART ? 1 : (STATS ? VRG_PRODUCE_WAVES : VRG_PRODUCE_WAVES_LABONLY)) void k_produce_lab(const float* __restrict__ in, float* __restrict__ lab_out, ProduceK P, ChainK D,
                                                      const float* __restrict__ pivots, double* __restrict__ rec,
                                                      int32_t* __restrict__ rec_frame) {
    // optional occupancy cap of the Lab-only form (a launch bound only sets a minimum): unused LDS makes the (N + 1)-th workgroup not fit
    constexpr int CAP = (!TWO_PART && !STATS) ? VRG_PRODUCE_MAX_WAVES_LABONLY : 8;
    constexpr int PAD = CAP >= 8 ? 16 : (160 * 1024 / (CAP + 1) + 512 - 14400 > 16 ? 160 * 1024 / (CAP + 1) + 512 - 14400 : 16);
    __shared__ char occupancy_pad[PAD];
    if (P.numel < 0) occupancy_pad[threadIdx.x] = (char)blockIdx.x;      // never true: keeps the array
    __shared__ float sn[4][PR_SUB + 4];
    __shared__ double red[STATS ? 4 : 1][12];
    VRG_CM_MATH(PT, true, (STAGES & VRG_STAGE_FASTMATH) != 0, D.dm);
    produce_lab_body<STAGES, TWO_PART, STATS>(blockIdx.x, in, lab_out, P, D, pivots, rec, rec_frame, sn, red, PT);
}

// Lab of every frame's first pixel after the pre stages: the pivot of that frame's shifted sums
template <int STAGES>
__global__ __launch_bounds__(64) void k_frame_pivots(const px3* __restrict__ in, int32_t ppf, int64_t frames, ChainK D, float* __restrict__ pivots) {
    VRG_CM_MATH(PT, true, (STAGES & VRG_STAGE_FASTMATH) != 0, D.dm);
    const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (f >= frames) return;
    const px3 v0 = in[f * ppf];
    const float x0[3] = {v0.r, v0.g, v0.b};
    float pre[3], lab[3];
    chain_pre<STAGES>(D, frame_ctx<STAGES>(D, f), 0, x0, pre, PT);
    rgb_to_lab(pre, lab, PT);
    pivots[f * 3] = lab[0]; pivots[f * 3 + 1] = lab[1]; pivots[f * 3 + 2] = lab[2];
}

// per frame: add the records tagged with this frame (they all live in the frame's chunk) in a fixed order
__global__ __launch_bounds__(256) void k_produce_merge(const double* __restrict__ rec, const int32_t* __restrict__ rec_frame,
                                                        const float* __restrict__ pivots, uint32_t recs_per_chunk, int32_t chunk_frames,
                                                        int32_t ppf, double* __restrict__ stats) {
    __shared__ double red[4][6];
    const int32_t f = blockIdx.x;
    const int64_t first = (int64_t)(f / chunk_frames) * recs_per_chunk;
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (uint32_t r = threadIdx.x; r < recs_per_chunk; r += 256) {
        if (rec_frame[first + r] == f) {
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] += rec[(first + r) * 6 + i];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double a = wave_sum_f64(acc[i]);
        if (lane == 0) red[wave][i] = a;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        const double s1 = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
        const double s2 = ((red[0][3 + c] + red[1][3 + c]) + red[2][3 + c]) + red[3][3 + c];
        const double n = (double)ppf;
        const double dm = s1 / n;
        double m2 = s2 - s1 * dm;
        if (m2 < 0.0) m2 = 0.0;
        stats[((int64_t)f * 3 + c) * 3 + 0] = n;
        stats[((int64_t)f * 3 + c) * 3 + 1] = (double)pivots[(int64_t)f * 3 + c] + dm;
        stats[((int64_t)f * 3 + c) * 3 + 2] = m2;
    }
}

// Can the shared-Philox pass 1 be used?  (grain stage present; a block's run crosses at most one frame boundary)
bool produce_applicable(int stages, int64_t frame_elems) {
    return (stages & VRG_STAGE_GRAIN) && !(stages & VRG_STAGE_COLORMATCH) && frame_elems >= PR_RUN + 3;
}

int64_t produce_scratch_bytes(const ChainK& D, int64_t frames, int64_t fe) {
    ProduceK P;
    produce_geometry(D, frames, fe, P);
    const int64_t blocks = (int64_t)P.chunks * P.K * P.NB;
    return blocks * 8 * (6 * 8 + 4) + frames * 3 * 4 + 256;
}

template <int STAGES>
static int launch_produce_t(const float* in, float* lab_out, int64_t frames, int32_t H, int32_t W, const ChainK& D, double* stats,
                            void* scratch, hipStream_t st) {
    const int64_t fe = (int64_t)H * W * 3;
    ProduceK P;
    produce_geometry(D, frames, fe, P);
    const int64_t blocks = (int64_t)P.chunks * P.K * P.NB;
    if (blocks >= (1ll << 24) || frames % D.noise.chunk_frames) return VRG_ERR_UNSUPPORTED;
    if (!stats) {                    // Lab image only
        hipLaunchKernelGGL((k_produce_lab<STAGES, false, false>), dim3((uint32_t)blocks), dim3(256), 0, st, in, lab_out, P, D, (const float*)nullptr,
                           (double*)nullptr, (int32_t*)nullptr);
        return hipGetLastError() == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH;
    }
    char* base = reinterpret_cast<char*>(scratch);
    double* rec = reinterpret_cast<double*>(base);
    int32_t* rec_frame = reinterpret_cast<int32_t*>(base + blocks * 8 * 6 * 8);
    float* pivots = reinterpret_cast<float*>(base + blocks * 8 * (6 * 8 + 4));
    const px3* src = reinterpret_cast<const px3*>(in);
    hipLaunchKernelGGL(k_frame_pivots<STAGES>, dim3((uint32_t)((frames + 63) / 64)), dim3(64), 0, st, src, (int32_t)(H * W), frames, D, pivots);
    hipLaunchKernelGGL((k_produce_lab<STAGES, false>), dim3((uint32_t)blocks), dim3(256), 0, st, in, lab_out, P, D, pivots, rec, rec_frame);
    if (P.chunk_frames > 1)
        hipLaunchKernelGGL((k_produce_lab<STAGES, true>), dim3(P.chunks * (uint32_t)(P.chunk_frames - 1) * 4u), dim3(256), 0, st, in, lab_out, P,
                           D, pivots, rec, rec_frame);
    hipLaunchKernelGGL(k_produce_merge, dim3((uint32_t)frames), dim3(256), 0, st, rec, rec_frame, pivots, P.K * P.NB * 8u,
                       P.chunk_frames, (int32_t)(H * W), stats);
    return hipGetLastError() == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH;
}

int launch_produce(const float* in, float* lab_out, int64_t frames, int32_t H, int32_t W, const ChainK& D, int stages, double* stats,
                   void* scratch, hipStream_t st) {
    if (stages & VRG_STAGE_FASTMATH) {
        if ((stages & 3) == 3) return launch_produce_t<3 | VRG_STAGE_FASTMATH>(in, lab_out, frames, H, W, D, stats, scratch, st);
        return launch_produce_t<1 | VRG_STAGE_FASTMATH>(in, lab_out, frames, H, W, D, stats, scratch, st);
    }
    if ((stages & 3) == 3) return launch_produce_t<3>(in, lab_out, frames, H, W, D, stats, scratch, st);
    return launch_produce_t<1>(in, lab_out, frames, H, W, D, stats, scratch, st);
}

}  // namespace vrg
