// vrg_common.hpp -- kernel-side parameter blocks shared by the translation units of libvrgdg_hip.so
#pragma once

// The tuning constants and the A/B or ablation switches that rounds 1-4 passed with -D are FIXED in the product sources (plain #defines
// next to their measurements; the march's and the wrong-pixel ablations are gone altogether).  Nothing in this library is selected by a
// command-line macro: a -D of one of the old names is a build error, so that no flag can turn the product into another -- or a
// wrong-answer -- library.  Variants for A/B runs are separate source files under tools/ab/ (tools/build_variant.py --source, which
// defines VRG_LAB_VARIANT_SOURCE for that one file: the frozen round-4 march still takes its -DVRG_MARCH_... switches that way).
#if !defined(VRG_LAB_VARIANT_SOURCE) && (defined(VRG_APPLY_EARLY_LOAD) || defined(VRG_APPLY_ROWS) || defined(VRG_FLAT_ROWS) || defined(VRG_GRAIN_NT) || \
    defined(VRG_PRODUCE_MAX_WAVES_LABONLY) || defined(VRG_PRODUCE_WAVES) || defined(VRG_PRODUCE_WAVES_LABONLY) || defined(VRG_PR_PIPE) || \
    defined(VRG_PR_SUBS) || defined(VRG_SG_PIPE) || defined(VRG_TILE_H) || defined(VRG_TILE_PRELOAD) || \
    defined(VRG_TS_LANES) || defined(VRG_TS_LANES_DEPTH) || defined(VRG_TS_LANES_MAX_FRAMES) || defined(VRG_TS_LANES_MEAN_DEPTH) || \
    defined(VRG_TS_MARKSTEIN) || defined(VRG_TS_ROWS_DEPTH) || defined(VRG_TS_ROWS_MAX_FRAMES) || defined(VRG_TS_SPLIT_DEPTH) || \
    defined(VRG_TS_SPLIT_MAX_FRAMES) || defined(VRG_TS_WHOLE_DEPTH) || defined(VRG_ZIV_REL) || defined(VRG_MARCH_FAST) || \
    defined(VRG_MARCH_MIN_WAVES) || defined(VRG_MARCH_FINITE) || defined(VRG_MARCH_ABLATE) || defined(VRG_MARCH_QUAD) || \
    defined(VRG_MARCH_ENDIO) || defined(VRG_MARCH_ROTATE) || defined(VRG_MARCH_TAPS_UNFOLD) || defined(VRG_MARCH_FAST_FLAT) || \
    defined(VRG_MARCH_EDGE_SHARED) || defined(VRG_MARCH_FAST_LDS) || defined(VRG_NO_POINTWISE4) || defined(VRG_NO_APPLY_MARCH) || \
    defined(VRG_NO_FLAT_STENCIL) || defined(VRG_APPLY_FORCE_GENERAL) || defined(VRG_ABLATE_GATHER) || defined(VRG_NO_DIVT_FASTPATH))
#error "comfyui-vrgamedevgirl_amd: tuning / ablation macros are not build options of the product sources (see the note above this line)"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vrgdg_hip.h"
#include "../../include/vrgdg_hip_debug.h"
#include "vrg_pixel_math.hpp"

namespace vrg {

// one RGB pixel; 4-byte aligned so that a load/store is a single global_*_dwordx3
struct __attribute__((packed, aligned(4))) px3 { float r, g, b; };

// Frame data is streamed once: point-wise kernels load it, and tile / point-wise kernels store it, with the
// non-temporal hint so that it does not displace the LUT records from the per-XCD L2 (measured: 3x3 stencil
// 4.3 -> 4.76 TB/s).  Kernels whose loads are re-read by neighbouring workgroups (tile halos) and the wave-march
// kernel (61-lane partial-line stores) use plain accesses.  clang merges the three scalars into one
// global_load/store_dwordx3 ... nt.
__device__ __forceinline__ px3 load_px_stream(const px3* p) {
    const float* f = reinterpret_cast<const float*>(p);
    px3 v;
    v.r = __builtin_nontemporal_load(f);
    v.g = __builtin_nontemporal_load(f + 1);
    v.b = __builtin_nontemporal_load(f + 2);
    return v;
}
__device__ __forceinline__ void store_px_stream(px3* p, const px3& v) {
    float* f = reinterpret_cast<float*>(p);
    __builtin_nontemporal_store(v.r, f);
    __builtin_nontemporal_store(v.g, f + 1);
    __builtin_nontemporal_store(v.b, f + 2);
}

// Frame element types at the kernel boundary.  IoF32: the reference's fp32 RGB tensors (12 B/px).  IoU8: decoded video
// frames as cv2 hands them over, uint8 BGR (3 B/px) -- the /255, the *255-clip-truncate and the channel swap of
// _frames_to_tensor / _tensor_to_frames happen in the load / store of the kernel that does the work, so a route batch
// moves 3 + 3 B/px through HBM (and over PCIe) instead of 12 + 12.
struct __attribute__((packed)) bgr8 { uint8_t b, g, r; };

struct IoF32 {
    typedef px3 elem;
    static __device__ __forceinline__ px3 load_stream(const elem* p) { return load_px_stream(p); }
    static __device__ __forceinline__ px3 load(const elem* p) { return *p; }
    static __device__ __forceinline__ void store_stream(elem* p, const px3& v) { store_px_stream(p, v); }
    static __device__ __forceinline__ void store(elem* p, const px3& v) { *p = v; }
};
struct IoU8 {
    typedef bgr8 elem;
    static __device__ __forceinline__ px3 load(const elem* p) {
        const uint8_t* q = reinterpret_cast<const uint8_t*>(p);
        return px3{unit_from_u8(q[2]), unit_from_u8(q[1]), unit_from_u8(q[0])};
    }
    static __device__ __forceinline__ px3 load_stream(const elem* p) { return load(p); }
    static __device__ __forceinline__ void store(elem* p, const px3& v) {
        uint8_t* q = reinterpret_cast<uint8_t*>(p);
        q[0] = u8_from_unit(v.b);
        q[1] = u8_from_unit(v.g);
        q[2] = u8_from_unit(v.r);
    }
    static __device__ __forceinline__ void store_stream(elem* p, const px3& v) { store(p, v); }
};

// Stage the node table of a small cube into LDS (one float4 per node) from the record table of vrg_lut_prepare_f32.
// Every thread of the workgroup must call it; the caller synchronises afterwards.
__device__ __forceinline__ void lut_nodes_to_lds(const LutParams& P, f32x4* nodes, int tid, int nthreads) {
    const int n = P.n, nc = n - 1, total = n * n * n;
    for (int i = tid; i < total; i += nthreads) {
        const int r = i % n, g = (i / n) % n, b = i / (n * n);
        const int b0 = b < nc ? b : nc - 1, g0 = g < nc ? g : nc - 1;
        const int k = (g - g0) * 2 + (b - b0);
        const float* rec = P.cells + (size_t)((b0 * nc + g0) * n + r) * LUT_REC_FLOATS;
        f32x4 v;
        v.x = rec[k]; v.y = rec[4 + k]; v.z = rec[8 + k]; v.w = 0.0f;
        nodes[i] = v;
    }
}

// Device copy of vrg_noise_desc plus the per-call geometry the noise mapping needs.
struct NoiseK {
    uint64_t seed0, seed_stride, off0, off_stride;
    int64_t chunk0;
    int64_t frame_elems;   // H*W*3
    int32_t chunk_frames;
    uint32_t G;
};

VRG_HD uint64_t chunk_seed(const NoiseK& n, int64_t chunk_rel) { return n.seed0 + (uint64_t)(n.chunk0 + chunk_rel) * n.seed_stride; }
VRG_HD uint64_t chunk_offset(const NoiseK& n, int64_t chunk_rel) { return n.off0 + (uint64_t)(n.chunk0 + chunk_rel) * n.off_stride; }

struct CmK {
    const float* img_ms;   // [frames][3][2]
    const float* ref_ms;   // [ref_frames][3][2]
    int32_t ref_frames;
    float K, T;
};

struct ChainK {
    int32_t stages;
    float I, S, T;
    NoiseK noise;
    LutParams lut;
    CmK cm;
    DevMath dm;             // pow exponents of the "device" colour-match arithmetic (runtime values, see vrg_pixel_math.hpp)
    int32_t stencil_op, zero_border;
    float strength;
};

// Internal template bit next to the public VRG_STAGE_* bits: Lab / colour-match arithmetic with the fast policy
// (PowTables) instead of the default device policy (DevMath).  Never part of vrg_chain_desc::stages; set from cm_math.
constexpr int VRG_STAGE_FASTMATH = 32;

inline DevMath host_dev_math() { return DevMath{(float)2.4, (float)(1.0 / 2.4), (float)(1.0 / 3.0), nullptr}; }

template <bool FAST> struct CmMathSel { typedef DevMath type; };
template <> struct CmMathSel<true> { typedef PowTables type; };
__device__ __forceinline__ PowTables cm_make_math(const float* lds, const DevMath&, const PowTables*) { return PowTables{lds, lds + 512}; }
__device__ __forceinline__ DevMath cm_make_math(const float* lds, const DevMath& dm, const DevMath*) {
    DevMath m = dm;
    m.logt = lds;            // dev_pow_ziv's log table, staged by VRG_CM_MATH
    return m;
}

inline NoiseK make_noise(const vrg_noise_desc* d, int64_t frame_elems) {
    NoiseK n;
    n.seed0 = d->seed0; n.seed_stride = d->seed_stride; n.off0 = d->offset0; n.off_stride = d->offset_stride;
    n.chunk0 = d->chunk0; n.frame_elems = frame_elems; n.chunk_frames = d->chunk_frames; n.G = d->grid_threads;
    return n;
}

inline LutParams make_lut(const float* cells, int n, const float dmin[3], const float dmax[3], int blend_mode,
                          float blend, float one_minus_blend) {
    LutParams P;
    P.cells = cells; P.n = n; P.top = (float)(n - 1);
    P.unit_domain = 1;
    for (int c = 0; c < 3; ++c) {
        P.dmin[c] = dmin[c];
        const float span = dmax[c] - dmin[c];
        P.span[c] = span < 1e-6f ? 1e-6f : span;   // torch.clamp(domain_max - domain_min, min=1e-6)
        if (!(P.dmin[c] == 0.0f && P.span[c] == 1.0f)) P.unit_domain = 0;
    }
    P.blend_mode = blend_mode; P.blend = blend; P.one_minus_blend = one_minus_blend;
    return P;
}

// The colour-match arithmetic object of a kernel: NEED = does the kernel evaluate Lab at all, FAST = policy.  The fast
// policy stages its pow tables in LDS (2.5 KB), the device policy the log table of dev_pow_ziv (2 KB).  Every thread of the
// block must execute it (barrier).
#define VRG_CM_MATH(PT, NEED, FAST, DM)                                                                       \
    __shared__ __attribute__((aligned(16))) float vrg_pow_lds_[(NEED) ? ((FAST) ? ::vrg::POW_TABLE_WORDS : ::vrg::ZIV_TABLE_WORDS) : 4]; \
    if (NEED) {                                                                                                \
        if (FAST) ::vrg::pow_tables_fill(vrg_pow_lds_, (int)threadIdx.x, (int)blockDim.x);                    \
        else ::vrg::ziv_table_fill(vrg_pow_lds_, (int)threadIdx.x, (int)blockDim.x);                          \
        __syncthreads();                                                                                       \
    }                                                                                                          \
    typedef typename ::vrg::CmMathSel<(FAST)>::type PT##_t;                                                    \
    const PT##_t PT = ::vrg::cm_make_math(vrg_pow_lds_, (DM), (const PT##_t*)nullptr)

#define VRG_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e_ = hipGetLastError();                   \
        if (e_ != hipSuccess) return VRG_ERR_LAUNCH;         \
    } while (0)

}  // namespace vrg
