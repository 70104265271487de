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
#include "vrg_chain_stages.hpp"

namespace vrg {

#ifndef VRG_APPLY_ROWS
#define VRG_APPLY_ROWS 60     /* rows per strip segment, a multiple of 3 (the row registers rotate by name) */
#endif
constexpr int APPLY_ROWS = VRG_APPLY_ROWS;
constexpr int APPLY_COLS = 62;                    // output columns per wave
static_assert(APPLY_ROWS % 3 == 0, "APPLY_ROWS must be a multiple of 3");

__device__ __forceinline__ float am_prev(float v) {   // value held by lane-1 (lane 0: unused)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float am_next(float v) {   // value held by lane+1 (lane 63: unused)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

struct AmRow { float l[3], c[3], r[3]; };               // one processed row: left tap, own value, right tap per channel

template <int STAGES>
__global__ __launch_bounds__(256) void k_apply_march(const px3* __restrict__ in, px3* __restrict__ out, int32_t H, int32_t W, int32_t strips_x,
                                                      int32_t segs_y, uint32_t total_waves, ChainK D) {
    VRG_CM_MATH(PT, (STAGES & VRG_STAGE_COLORMATCH) != 0, (STAGES & VRG_STAGE_FASTMATH) != 0, D.dm);
    // XCD-aware placement as in k_chain_tile: workgroup b runs on XCD b % 8; give every XCD one contiguous run of work
    // (a launch with fewer workgroups than wave groups walks them with a stride of gridDim.x / 8 per XCD: the persistent form, used
    // by an experiment that ran pass 2 beside pass 1 -- slower, ops.default_stats_pieces -- and kept because it costs nothing)
    const uint32_t groups = (total_waves + 3u) / 4u;
    const uint32_t per_xcd = (groups + 7u) / 8u;
    const int lane = threadIdx.x & 63;
    const bool zero = D.zero_border != 0;
    const int64_t ppf = (int64_t)H * W;
    const float n0[3] = {0.0f, 0.0f, 0.0f};
    for (uint32_t slot = blockIdx.x >> 3; slot < per_xcd; slot += (gridDim.x >> 3)) {
    const uint32_t grp = (blockIdx.x & 7u) * per_xcd + slot;
    if (grp >= groups) break;
    const uint32_t wv = grp * 4u + (threadIdx.x >> 6);
    if (wv >= total_waves) break;
    const uint32_t strip = wv % (uint32_t)strips_x;
    const uint32_t rest = wv / (uint32_t)strips_x;
    const uint32_t seg = rest % (uint32_t)segs_y;
    const int64_t f = rest / (uint32_t)segs_y;
    int32_t x0 = (int32_t)strip * APPLY_COLS, y0 = (int32_t)seg * APPLY_ROWS;
    if (W >= APPLY_COLS) x0 = x0 < W - APPLY_COLS ? x0 : W - APPLY_COLS;          // last strip: overlap instead of a ragged end
    if (H >= APPLY_ROWS) y0 = y0 < H - APPLY_ROWS ? y0 : H - APPLY_ROWS;          // last segment: likewise
    const int32_t rows = H >= APPLY_ROWS ? APPLY_ROWS : H;
    const int32_t x = x0 - 1 + lane;                                             // the column this lane evaluates
    const bool x_in = x >= 0 && x < W;
    const int32_t xc = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    const bool stores = lane >= 1 && lane <= APPLY_COLS && x_in;                 // (x_in: frames narrower than a strip)
    const px3* fin = in + f * ppf;
    px3* fout = out + f * ppf;
    const FrameCtx FC = frame_ctx<STAGES>(D, f);

    auto load = [&](int32_t y) {                                                  // raw pixel of row y (clamped), this lane's column
        const int32_t yc = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        return fin[(int64_t)yc * W + xc];
    };
    auto process = [&](const px3& v, int32_t y) {                                 // pre stages + the row's left / right taps
        AmRow r;
        const float xi[3] = {v.r, v.g, v.b};
        float o[3];
        chain_apply_stages<STAGES>(D, FC, xi, n0, o, PT);
        const bool inside = x_in && y >= 0 && y < H;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r.c[c] = (zero && !inside) ? 0.0f : o[c];                             // replicate border = the clamped coordinate's own value
            r.l[c] = am_prev(r.c[c]);
            r.r[c] = am_next(r.c[c]);
        }
        return r;
    };
    auto emit = [&](int32_t y, const AmRow& a, const AmRow& b, const AmRow& c) {
        float res[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float p[3][3] = {{a.l[ch], a.c[ch], a.r[ch]}, {b.l[ch], b.c[ch], b.r[ch]}, {c.l[ch], c.c[ch], c.r[ch]}};
            res[ch] = stencil_value(D.stencil_op, p, D.strength, D.zero_border);
        }
        if (stores && y < y0 + rows) fout[(int64_t)y * W + x] = px3{res[0], res[1], res[2]};
    };

    AmRow r0 = process(load(y0 - 1), y0 - 1);
    AmRow r1 = process(load(y0), y0);
    AmRow r2;
    const int32_t y1 = y0 + rows;
    px3 q = load(y0 + 1);
    for (int32_t y = y0; y < y1; y += 3) {                                        // three steps per trip: r0, r1, r2 rotate by name
        r2 = process(q, y + 1);
        q = load(y + 2);
        emit(y, r0, r1, r2);
        r0 = process(q, y + 2);
        q = load(y + 3);
        emit(y + 1, r1, r2, r0);
        r1 = process(q, y + 3);
        q = load(y + 4);
        emit(y + 2, r2, r0, r1);
    }
    }       // persistent walk
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
        hipLaunchKernelGGL((k_apply_march<STAGES>), dim3(blocks), dim3(256), 0, st, reinterpret_cast<const px3*>(in) + f0 * ppf,
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
