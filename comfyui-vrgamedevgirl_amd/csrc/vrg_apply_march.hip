// vrg_apply_march.hip -- chains WITHOUT a grain stage that end in a 3x3 stencil, as a register-resident wave march:
// (LUT) -> (colour match | colour match from a stored Lab image) -> sharpen.  gfx950 only.
//
// This is pass 2 of the headline chain (match -> Lab->RGB -> unsharp on the Lab image of pass 1).  The LDS-tile kernel
// (k_chain_tile) evaluates the pre stages for a 34 x 66 halo'd tile per 32 x 64 outputs (9.6 % recomputed), writes them to LDS,
// synchronises, and reads 30 LDS values per 8 outputs and channel.  Here one wave64 owns a strip of 62 output columns and marches
// down APPLY_ROWS rows: lane l evaluates the pre stages of column x0 - 1 + l ONCE per row (lanes 0 and 63 are the strip's halo
// columns: 2 of 64 = 3.2 % recomputed, plus 2 priming rows per segment), keeps the last three processed rows in registers, takes
// the left / right taps from the neighbouring lanes with DPP wave shifts (once per row, kept with the row), and stores 62 pixels
// per row.  No LDS tile, no barrier after the table fill.  Frame borders are coordinate clamping (replicate) or a zero mask, per
// lane, so strips and segments need no special cases; the last strip of a row and the last segment of a frame START EARLIER
// instead of ending short (they overlap their neighbours and store the same values twice).  Same per-pixel functions as every
// other kernel (chain_apply_stages, stencil_value): bit-identical results.
#include "vrg_apply_body.hpp"

namespace vrg {

template <int STAGES, bool GENERAL>
__global__ __launch_bounds__(256) void k_apply_march(const px3* __restrict__ in, px3* __restrict__ out, int32_t H, int32_t W, int32_t strips_x,
                                                      int32_t segs_y, uint32_t total_waves, ChainK D) {
    VRG_CM_MATH(PT, (STAGES & VRG_STAGE_COLORMATCH) != 0, (STAGES & VRG_STAGE_FASTMATH) != 0, D.dm);
    apply_march_body<STAGES, decltype(PT), GENERAL>(blockIdx.x, gridDim.x, in, out, H, W, strips_x, segs_y, total_waves, D, PT);
}

template <int STAGES>
static int launch_apply_march_t(const float* in, float* out, int64_t frames, int32_t H, int32_t W, const ChainK& D, hipStream_t st) {
    const int32_t strips_x = (W + APPLY_COLS - 1) / APPLY_COLS, segs_y = (H + APPLY_ROWS - 1) / APPLY_ROWS;
    const int64_t per_frame = (int64_t)strips_x * segs_y;
    int64_t step = ((int64_t)1 << 30) / per_frame;
    if (step < 1) return VRG_ERR_UNSUPPORTED;
    if ((STAGES & VRG_STAGE_COLORMATCH) && D.cm.ref_frames != 1) {
        if (step < D.cm.ref_frames) return VRG_ERR_UNSUPPORTED;
        step -= step % D.cm.ref_frames;
    }
    const int64_t ppf = (int64_t)H * W;
    for (int64_t f0 = 0; f0 < frames; f0 += step) {
        const int64_t nf = frames - f0 < step ? frames - f0 : step;
        ChainK d = D;
        if (STAGES & VRG_STAGE_COLORMATCH) d.cm.img_ms += f0 * 6;
        const uint32_t total = (uint32_t)(per_frame * nf);
        const uint32_t groups = (total + 3u) / 4u;
        const uint32_t blocks = ((groups + 7u) / 8u) * 8u;
        // frames of at least one strip x one segment and below 2 GiB: the form without conditional blocks (vrg_apply_body.hpp)
        if (W >= APPLY_COLS && H >= APPLY_ROWS && ppf * 12 < ((int64_t)1 << 31))
            hipLaunchKernelGGL((k_apply_march<STAGES, false>), dim3(blocks), dim3(256), 0, st, reinterpret_cast<const px3*>(in) + f0 * ppf,
                               reinterpret_cast<px3*>(out) + f0 * ppf, H, W, strips_x, segs_y, total, d);
        else
            hipLaunchKernelGGL((k_apply_march<STAGES, true>), dim3(blocks), dim3(256), 0, st, reinterpret_cast<const px3*>(in) + f0 * ppf,
                               reinterpret_cast<px3*>(out) + f0 * ppf, H, W, strips_x, segs_y, total, d);
        if (hipGetLastError() != hipSuccess) return VRG_ERR_LAUNCH;
    }
    return VRG_OK;
}

// chains the apply march takes: a colour-match stage (from RGB or from a stored Lab image), optionally behind a LUT, in front of a
// stencil; frames at least 3 rows high (shorter ones go to the tile kernel)
bool apply_march_applicable(int stages, int32_t H, int32_t W) {
    return (stages & VRG_STAGE_SHARPEN) && (stages & VRG_STAGE_COLORMATCH) && !(stages & VRG_STAGE_GRAIN) && H >= 3 && W >= 3;
}

int launch_apply_march(const float* in, float* out, int64_t frames, int32_t H, int32_t W, const ChainK& D, int stages, bool fast, hipStream_t st) {
    constexpr int CM = VRG_STAGE_COLORMATCH, FL = VRG_STAGE_FROM_LAB, FM = VRG_STAGE_FASTMATH, LU = VRG_STAGE_LUT;
    const int key = (stages & (LU | FL)) | (fast ? FM : 0);
    switch (key) {
        case 0: return launch_apply_march_t<CM>(in, out, frames, H, W, D, st);
        case LU: return launch_apply_march_t<CM | LU>(in, out, frames, H, W, D, st);
        case FL: return launch_apply_march_t<CM | FL>(in, out, frames, H, W, D, st);
        case FM: return launch_apply_march_t<CM | FM>(in, out, frames, H, W, D, st);
        case LU | FM: return launch_apply_march_t<CM | LU | FM>(in, out, frames, H, W, D, st);
        case FL | FM: return launch_apply_march_t<CM | FL | FM>(in, out, frames, H, W, D, st);
        default: return VRG_ERR_UNSUPPORTED;
    }
}

}  // namespace vrg
