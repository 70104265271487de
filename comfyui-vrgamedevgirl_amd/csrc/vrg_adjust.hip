// vrg_adjust.hip -- the 13-slider "Adjust" of the video routes:
// _apply_adjust_tensor, VRGDG_LUTVideoTools.py:307-391 of the reference.  gfx950 only.
//
//   point stage : clamp, white-balance shift, exposure, contrast, saturation, highlight / shadow / white / black
//                 masks on the post-saturation luma                                       (pure per pixel)
//   clarity     : x + (x - box9_reflect(x)) * clarity * 1.55 * (0.35 + midtone(x) * 0.65)  (k x k box, k = min(9, odd H, odd W))
//   sharpen     : x + (x - box3_replicate(x)) * sharpen * 5
//   tail        : fade, vignette (torch.linspace grid), clamp
//
// Bit-exactness needs the reference's summation order: avg_pool2d adds the k*k taps in raster order, one
// rounding per add, then divides by k*k.  The stencil kernels therefore keep one sequential chain per output
// (no separable / sliding sums); each thread produces 4 horizontally adjacent pixels so that one row of 12
// (resp. 6) LDS values feeds 4 chains.  The point stage is recomputed for the halo instead of being stored.
// With clarity AND sharpen the second box needs the first one's result in a 1-pixel ring: two passes through a
// caller-supplied temporary (48 B/px); every other combination is a single 24 B/px pass.
#include "vrg_common.hpp"
#include "vrg_adjust_math.hpp"

namespace vrg {

// ---------------------------------------------------------------------------------------------------------
// no stencil: point stage + tail, one pixel per thread.  enabled == 0: clamp only.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adjust_point(const px3* __restrict__ in, px3* __restrict__ out, int32_t H, int32_t W, AdjustK A) {
    const int32_t ppf = H * W;
    const int32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= ppf) return;
    const int64_t at = (int64_t)blockIdx.y * ppf + p;
    const px3 s = load_px_stream(in + at);
    const float x[3] = {s.r, s.g, s.b};
    float v[3];
    if (!A.enabled) {
        v[0] = clamp01(x[0]); v[1] = clamp01(x[1]); v[2] = clamp01(x[2]);
    } else {
        adjust_point(A, x, v);
        adjust_tail(A, p / W, p % W, H, W, v);
    }
    store_px_stream(out + at, px3{v[0], v[1], v[2]});
}

// ---------------------------------------------------------------------------------------------------------
// box-detail stencils.  Tile 32 x 64 output pixels, 256 threads, 4 adjacent pixels per thread, two rows of
// work per thread.  RADIUS 4 = clarity (reflect border, k x k box with k <= 9), RADIUS 1 = sharpen (replicate).
// PRE: the tile's input is the frame itself and the point stage is applied while filling LDS; otherwise the
// input already is the (clarity) intermediate.  TAIL: apply fade / vignette / clamp before storing.
// ---------------------------------------------------------------------------------------------------------
constexpr int AT_H = 32, AT_W = 64;

template <int RADIUS, bool PRE, bool TAIL>
__global__ __launch_bounds__(256) void k_adjust_box(const px3* __restrict__ in, px3* __restrict__ out, int32_t H, int32_t W,
                                                     int32_t tiles_x, AdjustK A) {
    constexpr int LH = AT_H + 2 * RADIUS, LW = AT_W + 2 * RADIUS, PITCH = LW + 4 - (LW % 4 ? LW % 4 : 4) + 4;
    __shared__ __attribute__((aligned(16))) float tile[3][LH][PITCH];
    const int32_t ty0 = (blockIdx.x / tiles_x) * AT_H;
    const int32_t tx0 = (blockIdx.x % tiles_x) * AT_W;
    const int64_t fbase = (int64_t)blockIdx.y * H * W;
    const px3* fin = in + fbase;
    const int rad = RADIUS == 1 ? 1 : A.box / 2;          // actual halo used by the box (<= RADIUS)

    for (int i = threadIdx.x; i < LH * LW; i += 256) {
        const int hy = i / LW, hx = i - hy * LW;
        int y = ty0 + hy - RADIUS, x = tx0 + hx - RADIUS;
        if (RADIUS == 1) {                                // F.pad(mode="replicate")
            y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
            x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
        } else {                                          // F.pad(mode="reflect"), valid for |offset| <= rad < dim
            if (y < 0) y = -y;
            if (y > H - 1) y = 2 * (H - 1) - y;
            if (x < 0) x = -x;
            if (x > W - 1) x = 2 * (W - 1) - x;
            y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);      // positions further out than `rad` are never read
            x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
        }
        const px3 s = fin[y * W + x];
        float v[3] = {s.r, s.g, s.b};
        if (PRE) {
            float o[3];
            adjust_point(A, v, o);
            v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
        }
        tile[0][hy][hx] = v[0];
        tile[1][hy][hx] = v[1];
        tile[2][hy][hx] = v[2];
    }
    __syncthreads();

    const int k = 2 * rad + 1;
    const float kk = (float)(k * k);
    px3* fout = out + fbase;
    for (int it = 0; it < 2; ++it) {
        const int q = threadIdx.x + 256 * it;             // 512 groups of 4 pixels
        const int ly = q / (AT_W / 4), lx = (q % (AT_W / 4)) * 4;
        const int y = ty0 + ly, x0 = tx0 + lx;
        if (y >= H) continue;
        float res[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            // raster order over the k x k window: for each row, taps left to right (avg_pool2d's running sum)
            for (int dy = -rad; dy <= rad; ++dy) {
                const float* row = &tile[c][ly + RADIUS + dy][lx + RADIUS - rad];
                float r[4 + 2 * RADIUS];
#pragma unroll
                for (int j = 0; j < 4 + 2 * RADIUS; ++j) r[j] = (j < 4 + 2 * rad) ? row[j] : 0.0f;
#pragma unroll
                for (int dx = 0; dx < 2 * RADIUS + 1; ++dx) {
                    if (dx < k) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) acc[o] = acc[o] + r[o + dx];
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) res[o][c] = RADIUS == 1 ? div9(acc[o]) : acc[o] / kk;
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int x = x0 + o;
            if (x >= W) continue;
            const float ctr[3] = {tile[0][ly + RADIUS][lx + RADIUS + o], tile[1][ly + RADIUS][lx + RADIUS + o],
                                  tile[2][ly + RADIUS][lx + RADIUS + o]};
            float v[3];
            if (RADIUS == 1) adjust_sharpen_mix(A, ctr, res[o], v);
            else adjust_clarity_mix(A, ctr, res[o], v);
            if (TAIL) adjust_tail(A, y, x, H, W, v);
            store_px_stream(fout + (y * W + x), px3{v[0], v[1], v[2]});
        }
    }
}

}  // namespace vrg

using namespace vrg;

extern "C" int vrg_adjust_f32(const float* in, float* out, float* tmp, int64_t frames, int32_t height, int32_t width,
                              const vrg_adjust_desc* d, void* stream) {
    if (!in || !out || !d || frames < 0 || height <= 0 || width <= 0) return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    const int64_t ppf = (int64_t)height * width;
    if (ppf > 0x7fffffff / 3) return VRG_ERR_UNSUPPORTED;
    AdjustK A{};
    A.enabled = d->enabled;
    for (int c = 0; c < 3; ++c) A.shift[c] = d->shift[c];
    A.exposure = d->exposure; A.contrast = d->contrast; A.saturation = d->saturation;
    A.highlights = d->highlights; A.shadows = d->shadows; A.whites = d->whites; A.blacks = d->blacks;
    A.clarity = d->clarity; A.sharpen = d->sharpen;
    A.fade_mul = d->fade_mul; A.fade_add = d->fade_add; A.vignette = d->vignette;
    A.has_fade = d->has_fade; A.has_vignette = d->has_vignette;
    const int box = adjust_box_size(height, width);
    A.box = box;
    A.has_clarity = d->enabled && d->has_clarity && box >= 3;      // kernel < 3: blur == source, detail == 0, x + 0*... == x
    A.has_sharpen = d->enabled && d->has_sharpen;
    hipStream_t st = (hipStream_t)stream;
    const px3* src = reinterpret_cast<const px3*>(in);
    px3* dst = reinterpret_cast<px3*>(out);
    const int tx = (width + AT_W - 1) / AT_W, ty = (height + AT_H - 1) / AT_H;
    for (int64_t f0 = 0; f0 < frames; f0 += 32768) {
        const uint32_t nf = (uint32_t)(frames - f0 < 32768 ? frames - f0 : 32768);
        const px3* s = src + f0 * ppf;
        px3* o = dst + f0 * ppf;
        const dim3 tg((uint32_t)(tx * ty), nf);
        if (!A.has_clarity && !A.has_sharpen) {
            hipLaunchKernelGGL(k_adjust_point, dim3((uint32_t)((ppf + 255) / 256), nf), dim3(256), 0, st, s, o, height, width, A);
        } else if (A.has_clarity && !A.has_sharpen) {
            hipLaunchKernelGGL((k_adjust_box<4, true, true>), tg, dim3(256), 0, st, s, o, height, width, tx, A);
        } else if (!A.has_clarity) {
            hipLaunchKernelGGL((k_adjust_box<1, true, true>), tg, dim3(256), 0, st, s, o, height, width, tx, A);
        } else {
            if (!tmp) return VRG_ERR_BAD_ARG;
            px3* t = reinterpret_cast<px3*>(tmp) + f0 * ppf;
            hipLaunchKernelGGL((k_adjust_box<4, true, false>), tg, dim3(256), 0, st, s, t, height, width, tx, A);
            hipLaunchKernelGGL((k_adjust_box<1, false, true>), tg, dim3(256), 0, st, t, o, height, width, tx, A);
        }
        if (hipGetLastError() != hipSuccess) return VRG_ERR_LAUNCH;
    }
    return VRG_OK;
}
