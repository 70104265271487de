// vrg_stencil.hip -- stand-alone 3x3 stencils (unsharp / laplacian / sobel), any channel count,
// replicate or zero border.  One output element per thread, consecutive lanes = consecutive elements
// of a frame row, so all nine tap loads of a wave are contiguous 256-byte segments; the eight
// neighbour loads hit the vector L1 / L2 (each input byte leaves HBM once).
#include "vrg_common.hpp"

namespace vrg {

__global__ __launch_bounds__(256) void k_stencil3x3(const float* __restrict__ in, float* __restrict__ out, int32_t H, int32_t W,
                                                     int32_t C, int32_t op, int32_t zero_border, float strength) {
    const int32_t row_elems = W * C;
    const int32_t frame_elems = H * row_elems;
    const int32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= frame_elems) return;
    const int64_t fbase = (int64_t)blockIdx.y * frame_elems;
    const int32_t y = e / row_elems;
    const int32_t xc = e - y * row_elems;
    const int32_t x = xc / C;
    const float* src = in + fbase;
    float p[3][3];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int yy = y + dy, xx = x + dx;
            const bool inside = (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            const float v = src[yy * row_elems + (xc + (xx - x) * C)];
            p[dy + 1][dx + 1] = (zero_border && !inside) ? 0.0f : v;
        }
    }
    out[fbase + e] = stencil_value(op, p, strength, zero_border);
}

}  // namespace vrg

using namespace vrg;

extern "C" int vrg_stencil3x3_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width, int32_t channels,
                                  int32_t op, int32_t border, float strength, void* stream) {
    if (!in || !out || frames < 0 || height <= 0 || width <= 0 || channels <= 0 || op < 0 || op > 2 || border < 0 || border > 1)
        return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    if (channels == 3 && (int64_t)height * width <= 0x7fffffff / 3) {
        // RGB frames: the LDS-tiled kernel of the fused chain with only the stencil stage enabled
        // (each input byte leaves HBM once and is re-read from LDS, not from L1/L2)
        vrg_chain_desc d{};
        d.stages = VRG_STAGE_SHARPEN;
        d.stencil_op = op; d.border = border; d.strength = strength;
        return vrg_fused_chain_f32(in, out, frames, height, width, &d, stream);
    }
    const int64_t fe = (int64_t)height * width * channels;
    if (fe > 0x7fffffff) return VRG_ERR_UNSUPPORTED;
    const uint32_t bx = (uint32_t)((fe + 255) / 256);
    for (int64_t f0 = 0; f0 < frames; f0 += 32768) {
        const int64_t nf = frames - f0 < 32768 ? frames - f0 : 32768;
        hipLaunchKernelGGL(k_stencil3x3, dim3(bx, (uint32_t)nf), dim3(256), 0, (hipStream_t)stream, in + f0 * fe, out + f0 * fe,
                           height, width, channels, op, border, strength);
        VRG_CHECK_LAUNCH();
    }
    return VRG_OK;
}
