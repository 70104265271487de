// vrg_lut_tetra.hip -- the per-pixel step of the opening colour match as ffmpeg performs it (reference filter graph
// "lut3d=file=...; blend=all_expr='A*(1-(w))+B*(w)'", VRGDG_WorkflowRunnerNodes.py:4407-4412 of the reference), on decoded
// 8-bit frames.  gfx950 only.
//
// ffmpeg is a third-party dependency of the reference and is absent here: this is a RESTATEMENT of its published filter
// arithmetic (libavfilter/vf_lut3d.c: interp_tetrahedral, the 8-bit packed-RGB slice function and parse_cube's scale;
// libavfilter/vf_blend.c: the expression path), PARITY UNPINNED -- no ffmpeg binary or vector to check it against, and the
// real graph also converts yuv420p <-> RGB around these filters (swscale), which is not part of the arithmetic here.
//   rgb     = byte * (1.0f / 255)                                    scale_f
//   s       = clip(rgb * (lut_scale * (N-1)), 0, N-1)                av_clipf; lut_scale = clip(1 / (max - min), 0, 1)
//   prev    = (int)s, next = min(prev + 1, N-1), d = s - prev
//   c       = tetrahedral interpolation of lut[r][g][b] on the ordering of (d.r, d.g, d.b): six cases, each
//             ((w0 * c000 + w1 * cA) + w2 * cB) + w3 * c111 with the weights of vf_lut3d.c, fp32, no contraction
//   matched = av_clip_uint8((int)(c * 255.0f))                       truncation, then clip to 0..255
//   out     = (uint8_t)(A * (1 - w) + matched * w)                   blend expression in double, A = source byte,
//                                                                    w = per-frame weight; truncation
// One pixel per thread; the cube of the opening colour match is 17^3 (59 KB: L1 / L2 resident), and the callers are
// codec-bound, so there is nothing to tune here.
#include "vrg_common.hpp"

namespace vrg {

struct TetraK {
    const float* table;      // [N][N][N][3], index [blue][green][red] (the .cube file order, red fastest)
    int32_t n;
    float scale[3];          // lut_scale * (N-1) per channel (R, G, B)
};

__device__ __forceinline__ float tetra_clipf(float a, float lo, float hi) { return a < lo ? lo : (a > hi ? hi : a); }
__device__ __forceinline__ uint8_t tetra_clip_u8(int a) { return (a & ~0xFF) ? (uint8_t)((~a) >> 31) : (uint8_t)a; }

__global__ __launch_bounds__(256) void k_lut3d_tetra_u8(const bgr8* __restrict__ in, bgr8* __restrict__ out, int64_t pixels, int64_t ppf, TetraK T,
                                                         const double* __restrict__ weights) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels) return;
    const bgr8 src = in[i];
    const int n = T.n;
    const float top = (float)(n - 1);
    const float scale_f = 1.0f / 255.0f;
    const uint8_t byte[3] = {src.r, src.g, src.b};
    float s[3], d[3];
    int prev[3], next[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (float)byte[c] * scale_f;
        s[c] = tetra_clipf(v * T.scale[c], 0.0f, top);
        prev[c] = (int)s[c];
        next[c] = prev[c] + 1 < n - 1 ? prev[c] + 1 : n - 1;
        d[c] = s[c] - (float)prev[c];
    }
    auto node = [&](int r, int g, int b) -> const float* { return T.table + ((size_t)((b * n + g) * n + r)) * 3; };
    const float* c000 = node(prev[0], prev[1], prev[2]);
    const float* c111 = node(next[0], next[1], next[2]);
    const float* ca;
    const float* cb;
    float w0, w1, w2, w3;
    const float dr = d[0], dg = d[1], db = d[2];
    if (dr > dg) {
        if (dg > db) {
            ca = node(next[0], prev[1], prev[2]); cb = node(next[0], next[1], prev[2]);       // c100, c110
            w0 = 1.0f - dr; w1 = dr - dg; w2 = dg - db; w3 = db;
        } else if (dr > db) {
            ca = node(next[0], prev[1], prev[2]); cb = node(next[0], prev[1], next[2]);       // c100, c101
            w0 = 1.0f - dr; w1 = dr - db; w2 = db - dg; w3 = dg;
        } else {
            ca = node(prev[0], prev[1], next[2]); cb = node(next[0], prev[1], next[2]);       // c001, c101
            w0 = 1.0f - db; w1 = db - dr; w2 = dr - dg; w3 = dg;
        }
    } else {
        if (db > dg) {
            ca = node(prev[0], prev[1], next[2]); cb = node(prev[0], next[1], next[2]);       // c001, c011
            w0 = 1.0f - db; w1 = db - dg; w2 = dg - dr; w3 = dr;
        } else if (db > dr) {
            ca = node(prev[0], next[1], prev[2]); cb = node(prev[0], next[1], next[2]);       // c010, c011
            w0 = 1.0f - dg; w1 = dg - db; w2 = db - dr; w3 = dr;
        } else {
            ca = node(prev[0], next[1], prev[2]); cb = node(next[0], next[1], prev[2]);       // c010, c110
            w0 = 1.0f - dg; w1 = dg - dr; w2 = dr - db; w3 = db;
        }
    }
    uint8_t matched[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = ((w0 * c000[c] + w1 * ca[c]) + w2 * cb[c]) + w3 * c111[c];
        matched[c] = tetra_clip_u8((int)(v * 255.0f));
    }
    if (weights) {
        const double w = weights[i / ppf];
        const double u = 1.0 - w;
#pragma unroll
        for (int c = 0; c < 3; ++c) matched[c] = (uint8_t)((double)byte[c] * u + (double)matched[c] * w);
    }
    bgr8 o;
    o.r = matched[0]; o.g = matched[1]; o.b = matched[2];
    out[i] = o;
}

}  // namespace vrg

using namespace vrg;

extern "C" int vrg_lut3d_tetra_u8(const uint8_t* in, uint8_t* out, int64_t frames, int64_t pixels_per_frame, const float* table,
                                  int32_t lut_size, const float* domain_min, const float* domain_max, const double* weights,
                                  void* stream) {
    if (!in || !out || !table || !domain_min || !domain_max || frames < 0 || pixels_per_frame < 0 || lut_size < 2 || lut_size > 256)
        return VRG_ERR_BAD_ARG;
    const int64_t pixels = frames * pixels_per_frame;
    if (pixels == 0) return VRG_OK;
    const uint64_t blocks = (uint64_t)(pixels + 255) / 256;
    if (blocks > 0x7fffffffull) return VRG_ERR_UNSUPPORTED;
    TetraK T;
    T.table = table;
    T.n = lut_size;
    for (int c = 0; c < 3; ++c) {
        // parse_cube: lut3d->scale = av_clipf(1. / (max - min), 0.f, 1.f) -- the quotient in double, rounded to float by the call
        const float span = domain_max[c] - domain_min[c];                 // float min[3], max[3] there
        const float sc = (float)(1.0 / (double)span);
        const float cl = sc < 0.0f ? 0.0f : (sc > 1.0f ? 1.0f : sc);
        T.scale[c] = cl * (float)(lut_size - 1);
    }
    hipLaunchKernelGGL(k_lut3d_tetra_u8, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const bgr8*>(in),
                       reinterpret_cast<bgr8*>(out), pixels, pixels_per_frame, T, weights);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}
