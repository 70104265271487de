// vrg_chain_stages.hpp -- the "pre" stages of the fused chain (grain, LUT, colour match) as pure
// per-pixel device functions shared by the tile / point-wise kernels (vrg_chain.hip), the statistics
// reduction and the wave-march kernel (vrg_march.hip).
#pragma once
#include "vrg_common.hpp"

namespace vrg {

// What the pre stages need per FRAME: the colour-match statistics rows and the Philox stream of the frame's noise chunk.
// Every kernel's workgroups stay inside one frame, so this is wave-uniform; it is resolved once per workgroup (the frame
// index is 64-bit, and `f % ref_frames` / `f / chunk_frames` written per pixel cost ~130 scalar instructions each that the
// compiler does not hoist out of a divergent pixel loop).
struct FrameCtx {
    float ims[6], rms[6];        // [3][2] (mean, std) of the frame / of the reference frame it is matched to
    SigmaRecip sr;               // the frame's refined reciprocals of std (vrg_pixel_math.hpp: the unscaled form of (lab - mean) / std)
    uint64_t seed, off;          // generator seed / offset of the frame's noise chunk
    uint64_t elem0;              // element index of the frame's first element inside its chunk
};

template <int STAGES>
__device__ __forceinline__ FrameCtx frame_ctx(const ChainK& D, int64_t f) {
    FrameCtx C;
    if (STAGES & VRG_STAGE_COLORMATCH) {
        const float* ims = D.cm.img_ms + f * 6;
        const float* rms = D.cm.ref_ms + (D.cm.ref_frames == 1 ? 0 : (f % D.cm.ref_frames)) * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) { C.ims[i] = ims[i]; C.rms[i] = rms[i]; }
        C.sr = sigma_recip(C.ims);
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) { C.ims[i] = 0.0f; C.rms[i] = 0.0f; }
        C.sr.usable = false;
        C.sr.y1[0] = C.sr.y1[1] = C.sr.y1[2] = 0.0f;
    }
    if (STAGES & VRG_STAGE_GRAIN) {
        const int64_t chunk = f / D.noise.chunk_frames;
        const int64_t fl = f - chunk * D.noise.chunk_frames;
        C.elem0 = (uint64_t)(fl * D.noise.frame_elems);
        C.seed = chunk_seed(D.noise, chunk);
        C.off = chunk_offset(D.noise, chunk);
    } else {
        C.elem0 = 0; C.seed = 0; C.off = 0;
    }
    return C;
}

// grain (with the pixel's three raw normals n) -> LUT -> colour match for a pixel of the frame described by C
template <int STAGES, class MATH>
__device__ __forceinline__ void chain_apply_stages(const ChainK& D, const FrameCtx& C, const float xin[3], const float n[3], float o[3],
                                                   const MATH& PT, const f32x4* lut_nodes = nullptr) {
    float v[3] = {xin[0], xin[1], xin[2]};
    if (STAGES & VRG_STAGE_GRAIN) {
        float g[3];
        grain_pixel(v, n, D.I, D.S, D.T, g);
        v[0] = g[0]; v[1] = g[1]; v[2] = g[2];
    }
    if (STAGES & VRG_STAGE_LUT) {
        float g[3];
        if (lut_nodes) lut_pixel_nodes(D.lut, lut_nodes, v, g);      // small cube staged in LDS by the kernel (uniform choice)
        else lut_pixel(D.lut, v, g);
        v[0] = g[0]; v[1] = g[1]; v[2] = g[2];
    }
    if (STAGES & VRG_STAGE_COLORMATCH) {
        float g[3];
        if (STAGES & VRG_STAGE_FROM_LAB)
            colormatch_from_lab(v, C.ims, C.rms, D.cm.K, D.cm.T, g, PT, &C.sr);     // the input pixel is already Lab
        else
            colormatch_pixel(v, C.ims, C.rms, D.cm.K, D.cm.T, g, PT, &C.sr);
        v[0] = g[0]; v[1] = g[1]; v[2] = g[2];
    }
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
}

// Same, drawing the pixel's normals with the general per-element routine (one Philox call per element):
// for pixel p of the frame described by C.
template <int STAGES, class MATH>
__device__ __forceinline__ void chain_pre(const ChainK& D, const FrameCtx& C, int32_t p, const float xin[3], float o[3],
                                          const MATH& PT) {
    float n[3] = {0.0f, 0.0f, 0.0f};
    if (STAGES & VRG_STAGE_GRAIN) {
        const uint64_t li = C.elem0 + (uint64_t)p * 3u;
#pragma unroll
        for (int c = 0; c < 3; ++c) n[c] = torch_randn_element(C.seed, C.off, D.noise.G, li + c);
    }
    chain_apply_stages<STAGES>(D, C, xin, n, o, PT);
}

}  // namespace vrg
