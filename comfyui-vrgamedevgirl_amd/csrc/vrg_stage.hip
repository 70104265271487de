// vrg_stage.hip -- one launch = one stage of the headline chain's software pipeline over frame ranges.  gfx950 only.
//
// grain -> LUT -> colour match -> sharpen runs as pass 1 (grain -> LUT -> Lab, k_produce_lab), the torch-order statistics of the Lab
// image (k_tstats_*), and pass 2 (match -> Lab->RGB -> stencil, k_apply_march).  Pass 1 is bound by the LUT's gather address path AND
// the vector ALUs (64 % busy), pass 2 by the vector ALUs alone, the statistics by the latency of their dependent update chains: three
// different bottlenecks of a CU.  Two kernels on two streams do not share a CU on this machine (full-size grids are dispatched one
// after the other; profiles/r03_pipelined_pieces_sweep.log, r03_stats_overlap_sweep.log, r03_persistent_pass2_overlap_sweep.log), but
// workgroups of ONE launch do: foreign vector-ALU work placed inside pass 1's launch costs a third to a half of what it costs as a
// kernel of its own (profiles/r03_pass1_valu_ballast_experiment.txt).  So a stage kernel gives every workgroup a ROLE:
//     statistics (half-block rows form) of frame range s-1   -- first in the grid, so that their long chains start at once
//     pass 1 of frame range s   /   pass 2 of frame range s-2  -- interleaved evenly (Bresenham) over the rest of the grid
// and the host walks s = 0 .. ranges + 1 (ops.fused_chain).  Dependencies are kernel boundaries: the statistics of a range need its
// Lab image (previous stage), pass 2 needs the statistics (previous stage + the finishing kernel).  Every role runs the SAME device
// function as its stand-alone kernel (vrg_produce_body.hpp, vrg_tstats_body.hpp, vrg_apply_body.hpp): results are bit-identical.
#include "vrg_apply_body.hpp"
#include "vrg_produce_body.hpp"
#include "vrg_tstats_body.hpp"

namespace vrg {

int fill_chain(const vrg_chain_desc* d, int32_t H, int32_t W, ChainK& D);                      // vrg_chain.hip
bool produce_applicable(int stages, int64_t frame_elems);                                       // vrg_produce.hip
bool apply_march_applicable(int stages, int32_t H, int32_t W);                                  // vrg_apply_march.hip
int ts_rows_geometry(int64_t n, int b, int num_mp, int& bw, int& bh, float& factor);            // vrg_torch_stats.hip
int ts_rows_finish(const void* rows, int64_t frames, int bh, float factor, float eps, float* out, hipStream_t st);

struct StageK {
    // role T: statistics rows of t_frames frames
    const float* t_lab; int64_t t_n, t_frames; int32_t t_bw, t_bh; TsRows* t_rows; uint32_t t_blocks;
    // role P1
    const float* p1_in; float* p1_lab; ProduceK p1; uint32_t p1_blocks;
    // role P2
    const px3* p2_in; px3* p2_out; int32_t H, W, strips_x, segs_y; uint32_t p2_total_waves, p2_blocks;
};

template <int PRE>                       // PRE = the pre stages of pass 1: VRG_STAGE_GRAIN, optionally | VRG_STAGE_LUT
__global__ __launch_bounds__(256, 5) void k_stage(StageK S, ChainK D1, ChainK D2) {      // 5 waves per SIMD: at most 96 VGPRs
    __shared__ __attribute__((aligned(16))) float sn[4][PR_SUB + 4];                            // pass 1's staged normals; the statistics role's tree buffer
    VRG_CM_MATH(PT, true, false, D1.dm);                                                        // device policy: dev_pow_ziv's table
    static_assert(sizeof(float) * 4 * (PR_SUB + 4) >= sizeof(Welf) * 256, "the statistics role reuses pass 1's LDS");
    const uint32_t b = blockIdx.x;
    if (b < S.t_blocks) {
        tstats_rows_body<2>(b, S.t_lab, S.t_n, S.t_frames, S.t_bw, S.t_bh, S.t_rows, reinterpret_cast<Welf*>(&sn[0][0]));
        return;
    }
    const uint32_t j = b - S.t_blocks, n12 = S.p1_blocks + S.p2_blocks;
    const uint32_t before = (uint32_t)(((uint64_t)j * S.p2_blocks) / n12);                      // pass-2 workgroups among the first j
    const uint32_t upto = (uint32_t)(((uint64_t)(j + 1) * S.p2_blocks) / n12);
    if (upto > before)
        apply_march_body<VRG_STAGE_COLORMATCH | VRG_STAGE_FROM_LAB>(before, S.p2_blocks, S.p2_in, S.p2_out, S.H, S.W, S.strips_x, S.segs_y,
                                                                    S.p2_total_waves, D2, PT);
    else
        produce_lab_body<PRE, false, false>(j - before, S.p1_in, S.p1_lab, S.p1, D1, nullptr, nullptr, nullptr, sn, nullptr, PT);
}

}  // namespace vrg

using namespace vrg;

extern "C" int64_t vrg_chain_stage_scratch_bytes(int64_t stats_frames) { return stats_frames <= 0 ? 0 : stats_frames * (int64_t)sizeof(TsRows); }

extern "C" int vrg_chain_stage_f32(const vrg_stage_desc* s, void* stream) {
    if (!s || s->height <= 0 || s->width <= 0 || s->p1_frames < 0 || s->stats_frames < 0 || s->p2_frames < 0) return VRG_ERR_BAD_ARG;
    if (s->p1_frames == 0 && s->stats_frames == 0 && s->p2_frames == 0) return VRG_OK;
    const int32_t H = s->height, W = s->width;
    if ((int64_t)H * W > 0x7fffffff / 3) return VRG_ERR_UNSUPPORTED;
    const int64_t fe = (int64_t)H * W * 3;
    hipStream_t st = (hipStream_t)stream;
    StageK S{};
    ChainK D1{}, D2{};
    D1.dm = host_dev_math(); D2.dm = host_dev_math();
    int pre = VRG_STAGE_GRAIN;
    if (s->p1_frames > 0) {
        if (!s->p1_in || !s->p1_lab || !s->p1_desc) return VRG_ERR_BAD_ARG;
        vrg_chain_desc d = *s->p1_desc;
        if (d.cm_math != VRG_CM_MATH_DEVICE) return VRG_ERR_UNSUPPORTED;
        d.stages &= (VRG_STAGE_GRAIN | VRG_STAGE_LUT);
        pre = d.stages;
        const int rc = fill_chain(&d, H, W, D1);
        if (rc) return rc;
        if (!produce_applicable(pre, fe) || s->p1_frames % D1.noise.chunk_frames) return VRG_ERR_UNSUPPORTED;
        produce_geometry(D1, s->p1_frames, fe, S.p1);
        const int64_t blocks = (int64_t)S.p1.chunks * S.p1.K * S.p1.NB;
        if (blocks >= (1ll << 24)) return VRG_ERR_UNSUPPORTED;
        S.p1_in = s->p1_in; S.p1_lab = s->p1_lab; S.p1_blocks = (uint32_t)blocks;
    }
    float factor = 0.0f;
    if (s->stats_frames > 0) {
        if (!s->stats_lab || !s->stats_mean_std || !s->stats_scratch || s->stats_chunk_frames <= 0) return VRG_ERR_BAD_ARG;
        if (s->stats_frames % s->stats_chunk_frames) return VRG_ERR_UNSUPPORTED;                 // whole calls of equal size only
        if (s->stats_scratch_bytes < vrg_chain_stage_scratch_bytes(s->stats_frames) || (reinterpret_cast<uintptr_t>(s->stats_scratch) & 15)) return VRG_ERR_BAD_ARG;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return VRG_ERR_NO_DEVICE;
        int bw = 0, bh = 0;
        const int rc = ts_rows_geometry((int64_t)H * W, s->stats_chunk_frames, cus, bw, bh, factor);
        if (rc) return rc;
        S.t_lab = s->stats_lab; S.t_n = (int64_t)H * W; S.t_frames = s->stats_frames; S.t_bw = bw; S.t_bh = bh;
        S.t_rows = reinterpret_cast<TsRows*>(s->stats_scratch);
        S.t_blocks = (uint32_t)(64 * ((s->stats_frames + 7) / 8));
    }
    if (s->p2_frames > 0) {
        if (!s->p2_lab || !s->p2_out || !s->p2_desc) return VRG_ERR_BAD_ARG;
        const vrg_chain_desc* d = s->p2_desc;
        if (d->cm_math != VRG_CM_MATH_DEVICE || d->stages != (VRG_STAGE_COLORMATCH | VRG_STAGE_FROM_LAB | VRG_STAGE_SHARPEN)) return VRG_ERR_UNSUPPORTED;
        if (!d->img_ms || !d->ref_ms || d->ref_frames < 1) return VRG_ERR_BAD_ARG;
        const int rc = fill_chain(d, H, W, D2);
        if (rc) return rc;
        if (!apply_march_applicable(d->stages, H, W)) return VRG_ERR_UNSUPPORTED;
        if (d->ref_frames != 1 && s->p2_frames % d->ref_frames) return VRG_ERR_UNSUPPORTED;
        S.H = H; S.W = W;
        S.strips_x = (W + APPLY_COLS - 1) / APPLY_COLS; S.segs_y = (H + APPLY_ROWS - 1) / APPLY_ROWS;
        const int64_t total = (int64_t)S.strips_x * S.segs_y * s->p2_frames;
        if (total >= (1ll << 30)) return VRG_ERR_UNSUPPORTED;
        S.p2_total_waves = (uint32_t)total;
        S.p2_blocks = (((uint32_t)total + 3u) / 4u + 7u) / 8u * 8u;
        S.p2_in = reinterpret_cast<const px3*>(s->p2_lab); S.p2_out = reinterpret_cast<px3*>(s->p2_out);
    }
    const uint64_t grid = (uint64_t)S.t_blocks + S.p1_blocks + S.p2_blocks;
    if (grid >= (1ull << 31)) return VRG_ERR_UNSUPPORTED;
    if (S.p1_blocks + S.p2_blocks == 0) {
        // statistics alone (no frames to produce or apply in this stage): the stand-alone kernels' job
        return VRG_ERR_UNSUPPORTED;
    }
    if (pre == (VRG_STAGE_GRAIN | VRG_STAGE_LUT)) hipLaunchKernelGGL((k_stage<VRG_STAGE_GRAIN | VRG_STAGE_LUT>), dim3((uint32_t)grid), dim3(256), 0, st, S, D1, D2);
    else hipLaunchKernelGGL((k_stage<VRG_STAGE_GRAIN>), dim3((uint32_t)grid), dim3(256), 0, st, S, D1, D2);
    if (hipGetLastError() != hipSuccess) return VRG_ERR_LAUNCH;
    if (s->stats_frames > 0) return ts_rows_finish(s->stats_scratch, s->stats_frames, S.t_bh, factor, s->stats_eps, s->stats_mean_std, st);
    return VRG_OK;
}
