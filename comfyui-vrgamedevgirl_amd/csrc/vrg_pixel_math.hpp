// vrg_pixel_math.hpp -- per-pixel arithmetic of the post-processing hot path, written so that
// every fp32 operation of the reference is one correctly rounded fp32 operation in the reference's
// order.  Build with -ffp-contract=off: nothing here may be contracted into
// an FMA except where __builtin_fmaf is written explicitly (the Box-Muller transform, where the
// reference *is* torch's device code and that code uses FMAs).
//
// Reference lines (relative to the reference checkout):
//   grain        nodes.py:51-60, VRGDG_LUTVideoTools.py:262-277
//   LUT          VRGDG_IV_Adjustments.py:288-361
//   stencils     nodes.py:156-384
//   colour match nodes.py:91-124 + kornia.color.{rgb_to_lab,lab_to_rgb} (external, restated)
//   noise        ATen/native/hip/DistributionTemplates.h:52-99, rocrand_philox4x32_10.h:150-310,
//                rocrand_normal.h:52-68
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define VRG_HD __host__ __device__ __forceinline__
#define VRG_D __device__ __forceinline__
#else
#define VRG_HD inline
#define VRG_D inline
#endif

// The three hardware transcendentals of the torch/rocRAND Box-Muller.  The arithmetic-order
// checker in tests/host_math/ (built without a GPU) overrides them with libm stand-ins; that
// build is test scaffolding only and is never loaded by the package.
#ifndef VRG_HW_LOG2
#define VRG_HW_LOG2(x) __builtin_amdgcn_logf(x)     /* v_log_f32  */
#define VRG_HW_SIN_REV(x) __builtin_amdgcn_sinf(x)  /* v_sin_f32, argument in revolutions */
#define VRG_HW_COS_REV(x) __builtin_amdgcn_cosf(x)  /* v_cos_f32 */
#define VRG_HW_EXP2(x) __builtin_amdgcn_exp2f(x)    /* v_exp_f32  */
#define VRG_HW_RCP(x) __builtin_amdgcn_rcpf(x)      /* v_rcp_f32  */
#endif

namespace vrg {

VRG_HD float f32_from_bits(uint32_t b) {
    union { uint32_t u; float f; } c;
    c.u = b;
    return c.f;
}
VRG_HD uint32_t f32_bits(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    return c.u;
}

// clamp(v, 0, 1) with torch.clamp / np.clip NaN behaviour (NaN stays NaN).
// Device: one v_med3_f32 plus a NaN pass-through (v_cmp_u + v_cndmask) -- 4 issue units instead of the 6 of two
// compare/select pairs (compare/select cost 1.5x, profiles/r01_valu_issue_rate.json); same value for every input.
VRG_HD float clamp01(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float m = __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f);
    return (v != v) ? v : m;
#else
    return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
#endif
}
// clamp01 of a pixel's three values.  The NaN pass-through costs a compare and a select per value; a NaN in any of the three makes their
// SUM a NaN (so does +Inf next to -Inf), so one comparison of the sum and a wave-uniform branch decide for the pixel: v_med3_f32 alone
// when no lane of the wave holds a NaN (every frame of a video), clamp01 otherwise.  Same values for every input.
VRG_HD void clamp01_3(const float v[3], float o[3]) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float s = (v[0] + v[1]) + v[2];
    const float a = v[0], b = v[1], c = v[2];
    float x = __builtin_amdgcn_fmed3f(a, 0.0f, 1.0f), y = __builtin_amdgcn_fmed3f(b, 0.0f, 1.0f), z = __builtin_amdgcn_fmed3f(c, 0.0f, 1.0f);
    if (__builtin_amdgcn_ballot_w64(s != s) != 0) {      // the rare side patches the three results in place: no copies where the sides meet
        x = (a != a) ? a : x;
        y = (b != b) ? b : y;
        z = (c != c) ? c : z;
    }
    o[0] = x; o[1] = y; o[2] = z;
#else
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = clamp01(v[c]);
#endif
}
// clamp(v, min=lo)
VRG_HD float clamp_min(float v, float lo) { return v < lo ? lo : v; }
// The base of a power whose value is only USED for v above the threshold (sRGB <-> linear, the Lab cube root: the reference evaluates
// pow on every element and selects afterwards): max(v, lo) in one v_max_f32 -- a NaN becomes lo here, and the select that follows takes
// the other branch for it (NaN > threshold is false), which is NaN: the same element the reference produces.  (clamp_min keeps the NaN
// and costs a compare, a select and their VCC wait states.)
VRG_HD float pow_base_min(float v, float lo) { return __builtin_fmaxf(v, lo); }
// clamp(v, 0, 1) in one v_med3_f32 where NaN cannot occur or the reference leaves NaN undefined (LUT index)
VRG_HD float clamp01_finite(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f);
#else
    return __builtin_fminf(__builtin_fmaxf(v, 0.0f), 1.0f);
#endif
}

// ------------------------------------------------------------------------------------------
// Division by a compile-time constant as multiply + two FMAs (Markstein): q = x*rc, r = fma(-c,q,x),
// q' = fma(r,rc,q) with rc = RN(1/c).  For the constants used here (1.055, 12.92, 0.95047, 1.08883,
// 116, 500, 200, 7.787, 9) this equals the IEEE quotient x/c bit for bit for every fp32 x with
// 1e-30 <= |x| <= 1e30 (checked exhaustively over all 2^32 inputs, tests/test_host_math.py and the
// -m gpu suite); for c = 9 it is exact for every finite x.  3 issue slots instead of ~14.
// ------------------------------------------------------------------------------------------
VRG_HD float div_const(float x, float c, float rc) {
    const float q = x * rc;
    const float r = __builtin_fmaf(-c, q, x);
    return __builtin_fmaf(r, rc, q);
}
#define VRG_DIVC(x, c) ::vrg::div_const((x), (c), 1.0f / (c))

// x / 9.0f for the unsharp mean; +-Inf handled so that the result is IEEE for every input
VRG_HD float div9(float x) {
    const float q = VRG_DIVC(x, 9.0f);
    return __builtin_isfinite(x) ? q : x;   // Inf/NaN pass through like x/9 (one v_cmp_class + v_cndmask)
}


// ------------------------------------------------------------------------------------------
// uint8 <-> unit-range fp32 at the video I/O edge:
// _frames_to_tensor: astype(float32) / 255.0      (VRGDG_LUTVideoTools.py:736-743, StandaloneVideoEnhancer:311-316)
// _tensor_to_frames: clip(x * 255.0, 0, 255).astype(uint8) -- truncation (LUTVideoTools.py:746-752, Enhancer:319-324)
// The FMA-form quotient equals v / 255.0f for all 256 inputs (tests/test_host_math.py).  NaN quantises to 0
// (what the x86 cast in numpy's astype yields).
// ------------------------------------------------------------------------------------------
VRG_HD float unit_from_u8(uint8_t v) { return VRG_DIVC((float)v, 255.0f); }
VRG_HD uint8_t u8_from_unit(float x) {
    const float y = x * 255.0f;
    const float c = __builtin_fminf(__builtin_fmaxf(y, 0.0f), 255.0f);     // fmax(NaN, 0) = 0
    return (uint8_t)(int)c;
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (rocrand_philox4x32_10.h:270-303)
// ------------------------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

VRG_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// One call of hiprand_normal4 after hiprand_init(seed, subsequence, offset) advanced by `call`
// calls: counter = offset/4 + call (64-bit, low words), subsequence in the high words.
VRG_HD u32x4 philox_for(uint64_t seed, uint64_t subsequence, uint64_t counter) {
    return philox4x32_10((uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)subsequence,
                         (uint32_t)(subsequence >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
}

// rocrand box_muller (rocrand_normal.h:52-68) exactly as hipcc compiles it inside torch's
// normal kernel for gfx950 (disassembly of libtorch_hip.so: u and v are single FMAs, logf is
// v_log_f32 + the two-constant ln2 product, sqrtf is correctly rounded, __sincosf is
// v_sin_f32 / v_cos_f32 of v * 1/(2*pi)).  u >= 2^-32 so logf's denormal pre-scale never fires.
struct f32x2 { float x, y; };

VRG_D float log_of_unit_uniform(float u) {
    const float y = VRG_HW_LOG2(u);
    const float ln2_hi = f32_from_bits(0x3f317217u);
    const float ln2_lo = f32_from_bits(0x3377d1cfu);
    const float r = y * ln2_hi;
    float e = __builtin_fmaf(y, ln2_hi, -r);
    e = __builtin_fmaf(y, ln2_lo, e);
    return r + e;
}

// Correctly rounded sqrt for 0 <= x < 2^96 that is zero or normal: the backend's own IEEE expansion (v_sqrt_f32, then pick among
// the estimate and its two neighbours by the sign of the exact residuals) WITHOUT its pre-scaling of inputs below 2^-96 and its
// Inf / NaN pass-through -- 9 instead of 16 instructions.  For x = 0 the neighbours are NaN / the smallest subnormal and both
// tests fail, so 0 stays 0.  Equal to __builtin_sqrtf for every value bm_radius can feed it (all 2^32 inputs swept on the
// device: vrg_selftest_bm_radius, run by the GPU tests).
VRG_D float sqrt_normal_range(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = f32_from_bits(__float_as_uint(s) - 1u);
    const float up = f32_from_bits(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-dn, s, x);
    const float ru = __builtin_fmaf(-up, s, x);
    float r = (rd <= 0.0f) ? dn : s;
    r = (ru > 0.0f) ? up : r;
    return r;
#else
    return __builtin_sqrtf(x);
#endif
}

VRG_D float bm_radius_arg(uint32_t a) {
    const float two_m32 = f32_from_bits(0x2f800000u);  // ROCRAND_2POW32_INV
    const float u = __builtin_fmaf((float)a, two_m32, two_m32);
    return -2.0f * log_of_unit_uniform(u);             // in [0, 44.4]: zero (a = 2^32-1 rounds u to 1) or normal
}

VRG_D float bm_radius(uint32_t a) { return sqrt_normal_range(bm_radius_arg(a)); }

VRG_D float bm_angle_rev(uint32_t b) {
    const float two_pi_m32 = f32_from_bits(0x30c90fdbu);  // ROCRAND_2POW32_INV_2PI
    const float v = __builtin_fmaf((float)b, two_pi_m32, two_pi_m32);
    return v * f32_from_bits(0x3e22f983u);  // * 1/(2*pi): __ocml_native_sin/cos_f32
}

// torch's normal transform: rand * std + mean with std = 1, mean = 0, contracted to one FMA.
VRG_D float torch_normal_affine(float r) { return __builtin_fmaf(1.0f, r, 0.0f); }

VRG_D f32x2 box_muller(uint32_t a, uint32_t b) {
    const float s = bm_radius(a);
    const float w = bm_angle_rev(b);
    return f32x2{torch_normal_affine(VRG_HW_SIN_REV(w) * s), torch_normal_affine(VRG_HW_COS_REV(w) * s)};
}

// component `ii` (0..3) of normal_distribution4 (rocrand_normal.h:259-265)
VRG_D float normal_component(const u32x4& r, int ii) {
    const uint32_t a = (ii < 2) ? r.x : r.z;
    const uint32_t b = (ii < 2) ? r.y : r.w;
    const float s = bm_radius(a);
    const float w = bm_angle_rev(b);
    const float t = (ii & 1) ? VRG_HW_COS_REV(w) : VRG_HW_SIN_REV(w);
    return torch_normal_affine(t * s);
}

// The N(0,1) value torch.randn writes to element `li` of a chunk: the slow, fully general form
// (one Philox call per element).  seed/offset are the chunk's generator state, G = grid threads.
VRG_D float torch_randn_element(uint64_t seed, uint64_t offset, uint32_t G, uint64_t li) {
    const uint64_t q = li / G;
    const uint64_t idx = li - q * G;
    const u32x4 r = philox_for(seed, idx, (offset >> 2) + (q >> 2));
    return normal_component(r, (int)(q & 3));
}

// ------------------------------------------------------------------------------------------
// Film grain (nodes.py:53-60)
// ------------------------------------------------------------------------------------------
// One element: n_own = raw normal of this element, n_green = raw normal of the pixel's G element.
VRG_HD float grain_element(float x, float n_own, float n_green, int channel, float I, float S, float T) {
    const float gain = channel == 0 ? 2.0f : (channel == 2 ? 3.0f : 1.0f);
    const float scaled = n_own * gain;
    const float a = S * scaled;
    const float b = T * n_green;
    const float g = a + b;
    const float d = g * I;
    return clamp01(x + d);
}
// the same before the clamp
VRG_HD float grain_element_raw(float x, float n_own, float n_green, int channel, float I, float S, float T) {
    const float gain = channel == 0 ? 2.0f : (channel == 2 ? 3.0f : 1.0f);
    const float scaled = n_own * gain;
    const float a = S * scaled;
    const float b = T * n_green;
    const float g = a + b;
    const float d = g * I;
    return x + d;
}

VRG_HD void grain_pixel(const float x[3], const float n[3], float I, float S, float T, float o[3]) {
    o[0] = grain_element(x[0], n[0], n[1], 0, I, S, T);
    o[1] = grain_element(x[1], n[1], n[1], 1, I, S, T);
    o[2] = grain_element(x[2], n[2], n[1], 2, I, S, T);
}
// the same values with the pixel's three clamps behind ONE NaN test and a wave-uniform branch (clamp01_3): for kernels whose loop
// tolerates a conditional block (pass 1 of the colour transfer; the wave-march kernels count their memory operations by hand and keep grain_pixel)
VRG_HD void grain_pixel_nan_branch(const float x[3], const float n[3], float I, float S, float T, float o[3]) {
    const float r[3] = {grain_element_raw(x[0], n[0], n[1], 0, I, S, T), grain_element_raw(x[1], n[1], n[1], 1, I, S, T),
                        grain_element_raw(x[2], n[2], n[1], 2, I, S, T)};
    clamp01_3(r, o);
}

// ------------------------------------------------------------------------------------------
// 3D LUT (VRGDG_IV_Adjustments.py:293-336, blend :355-359)
// ------------------------------------------------------------------------------------------
// The table is consumed in a GATHER-FRIENDLY form: one 48-byte record per (b0, g0, r) -- for each output
// channel the four (g,b) corners {g0b0, g0b1, g1b0, g1b1} at red grid node r.  The records of r and r+1
// are adjacent, so a pixel reads ONE contiguous 96-byte run (6 x 16 B) instead of 8 scattered 12-byte corners
// of the [N][N][N][3] table: for incoherent colours that cuts L2->L1 traffic from ~8 cache lines per pixel to
// ~1.6, and the table (N*(N-1)^2*48 B = 1.6 MB for 33^3) stays L2 resident next to the streaming frames.
// Built once per LUT by lut_build_record(); values are copied, never re-rounded.
struct alignas(16) f32x4 { float x, y, z, w; };

constexpr int LUT_REC_FLOATS = 12;

// (Round 5 measured a cell-major twin of this table -- one 128-byte aligned record per cell, ONE line per pixel -- for cubes up to 28^3,
// read by the march's quad-cooperative fetch: the fetch alone gains 18 % on a 25^3 cube (96 -> 113 Gpix/s, profiles/r05_probe_gather_25.json),
// the fused kernel LOSES 3-5 % (profiles/r05_ab_cellmajor_25.json: 2.6x the table bytes in the L2 beside the streaming frames).  Not kept.)
struct LutParams {
    const float* cells;  // [(N-1)][(N-1)][N][12]
    int n;               // N
    float top;           // (float)(N-1)
    float dmin[3];
    float span[3];       // max(dmax - dmin, 1e-6f)
    int unit_domain;     // dmin == 0 and span == 1 for all channels: (x-0)/1 == x exactly
    int blend_mode;      // 1 = LUT only, 2 = x*(1-B) + y*B
    float blend, one_minus_blend;
};

// record (b0,g0,r) of an [N][N][N][3] table (index [b][g][r]) -> 12 floats: [ch][dg*2 + db]
VRG_HD void lut_build_record(const float* table, int n, int b0, int g0, int r, float out[LUT_REC_FLOATS]) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = b0 + (k & 1), g = g0 + (k >> 1);
            out[ch * 4 + k] = table[(size_t)((b * n + g) * n + r) * 3 + ch];
        }
}

struct LutAxis { int cell; float f, u; };

// Per-axis index / weights (VRGDG_IV_Adjustments.py:295-318).  i0 = floor(c), i1 = min(i0+1, N-1),
// f = c - i0.  The cell index is min(i0, N-2): only the top grid node (c == N-1, i.e. t == 1) is affected,
// and there f = (N-1) - (N-2) = 1 exactly, u = 0, which selects the same corner value the reference's
// i0 == i1 == N-1, f == 0 does (C[N-2]*0 + C[N-1]*1 == C[N-1]*1 + C[N-1]*0 for every finite table).
VRG_HD LutAxis lut_axis(float x, float dmin, float span, int unit_domain, float top) {
    float t;
    if (unit_domain) {
        t = x;
    } else {
        const float d = x - dmin;
        t = d / span;
    }
    t = clamp01_finite(t);                       // NaN input: the reference indexes with garbage; we use cell 0
    const float c = t * top;
    const float fl = __builtin_fminf(__builtin_floorf(c), top - 1.0f);
    LutAxis a;
    a.cell = (int)fl;
    a.f = c - fl;
    a.u = 1.0f - a.f;
    return a;
}

VRG_HD float lerp2(float a, float wa, float b, float wb) {
    const float pa = a * wa;
    const float pb = b * wb;
    return pa + pb;
}

// rgb in -> graded rgb out (before the strength blend); lerp order blue, green, red (:320-333).
// Split into "issue" (axes + the six 16-byte gathers) and "finish" (the 21 separately rounded ops per channel) so that a
// kernel can keep the gathers of one pixel in flight while it works on another (vrg_march.hip).
struct LutFetch { LutAxis R, G, B; f32x4 lo[3], hi[3]; };

VRG_HD void lut_fetch_issue(const LutParams& P, const float x[3], LutFetch& F) {
    F.R = lut_axis(x[0], P.dmin[0], P.span[0], P.unit_domain, P.top);
    F.G = lut_axis(x[1], P.dmin[1], P.span[1], P.unit_domain, P.top);
    F.B = lut_axis(x[2], P.dmin[2], P.span[2], P.unit_domain, P.top);
    const int nc = P.n - 1;
    const f32x4* q = reinterpret_cast<const f32x4*>(P.cells + (size_t)((F.B.cell * nc + F.G.cell) * P.n + F.R.cell) * LUT_REC_FLOATS);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        F.lo[ch] = q[ch];      // red node r0: (g0,b0) (g0,b1) (g1,b0) (g1,b1)
        F.hi[ch] = q[3 + ch];  // red node r0 + 1
    }
}

VRG_HD void lut_fetch_finish(const LutFetch& F, float y[3]) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const f32x4 lo = F.lo[ch], hi = F.hi[ch];
        const float c00 = lerp2(lo.x, F.B.u, lo.y, F.B.f);
        const float c01 = lerp2(lo.z, F.B.u, lo.w, F.B.f);
        const float c10 = lerp2(hi.x, F.B.u, hi.y, F.B.f);
        const float c11 = lerp2(hi.z, F.B.u, hi.w, F.B.f);
        const float c0 = lerp2(c00, F.G.u, c01, F.G.f);
        const float c1 = lerp2(c10, F.G.u, c11, F.G.f);
        y[ch] = clamp01_finite(lerp2(c0, F.R.u, c1, F.R.f));    // finite table => finite value
    }
}

VRG_HD void lut_pixel_raw(const LutParams& P, const float x[3], float y[3]) {
    LutFetch F;
    lut_fetch_issue(P, x, F);
    lut_fetch_finish(F, y);
}

VRG_HD void lut_blend(const LutParams& P, const float x[3], const float y[3], float o[3]) {
    if (P.blend_mode == 2) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[ch] = lerp2(x[ch], P.one_minus_blend, y[ch], P.blend);
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[ch] = y[ch];
    }
}

// Same pixel from a node table held in LDS (one float4 {R, G, B, pad} per grid node, index [b][g][r]): cubes up to 21^3.
// Corner order and arithmetic are those of lut_pixel_raw, so the result is bit-identical.
VRG_HD void lut_pixel_nodes(const LutParams& P, const f32x4* T, const float x[3], float o[3]) {
    const LutAxis R = lut_axis(x[0], P.dmin[0], P.span[0], P.unit_domain, P.top);
    const LutAxis G = lut_axis(x[1], P.dmin[1], P.span[1], P.unit_domain, P.top);
    const LutAxis B = lut_axis(x[2], P.dmin[2], P.span[2], P.unit_domain, P.top);
    const int n = P.n;
    const int base = (B.cell * n + G.cell) * n + R.cell;
    const f32x4 q000 = T[base], q001 = T[base + n * n];                    // (g0, b0), (g0, b1) at red r0
    const f32x4 q010 = T[base + n], q011 = T[base + n * n + n];            // (g1, b0), (g1, b1)
    const f32x4 q100 = T[base + 1], q101 = T[base + n * n + 1];            // red r0 + 1
    const f32x4 q110 = T[base + n + 1], q111 = T[base + n * n + n + 1];
    float y[3];
#define VRG_NODE_LERP(CH, DST)                                                         \
    {                                                                                  \
        const float c00 = lerp2(q000.CH, B.u, q001.CH, B.f);                           \
        const float c01 = lerp2(q010.CH, B.u, q011.CH, B.f);                           \
        const float c10 = lerp2(q100.CH, B.u, q101.CH, B.f);                           \
        const float c11 = lerp2(q110.CH, B.u, q111.CH, B.f);                           \
        const float c0 = lerp2(c00, G.u, c01, G.f);                                    \
        const float c1 = lerp2(c10, G.u, c11, G.f);                                    \
        DST = clamp01_finite(lerp2(c0, R.u, c1, R.f));                                 \
    }
    VRG_NODE_LERP(x, y[0])
    VRG_NODE_LERP(y, y[1])
    VRG_NODE_LERP(z, y[2])
#undef VRG_NODE_LERP
    if (P.blend_mode == 2) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[ch] = lerp2(x[ch], P.one_minus_blend, y[ch], P.blend);
    } else {
        o[0] = y[0]; o[1] = y[1]; o[2] = y[2];
    }
}

VRG_HD void lut_pixel(const LutParams& P, const float x[3], float o[3]) {
    float y[3];
    lut_pixel_raw(P, x, y);
    if (P.blend_mode == 2) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[ch] = lerp2(x[ch], P.one_minus_blend, y[ch], P.blend);
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[ch] = y[ch];
    }
}

// ------------------------------------------------------------------------------------------
// 3x3 stencils.  p[r][c] = tap at (y-1+r, x-1+c) after border handling (replicated or 0.0f).
// ------------------------------------------------------------------------------------------
// *_raw<UNIT>: the value before the final clamp.  UNIT: every tap is in [0, 1] or NaN (pixels that left a clamp, zero borders): the nine-tap
// sum cannot be infinite, so x / 9 needs no Inf pass-through (a NaN goes through the FMA form by itself).
template <bool UNIT = false>
VRG_HD float unsharp_raw(const float p[3][3], float strength) {
    // nodes.py:194-207 (numpy) and avg_pool2d's running sum share this left-assoc raster order.
    float s = p[0][0] + p[0][1];
    s = s + p[0][2];
    s = s + p[1][0];
    s = s + p[1][1];
    s = s + p[1][2];
    s = s + p[2][0];
    s = s + p[2][1];
    s = s + p[2][2];
    const float blur = UNIT ? VRG_DIVC(s, 9.0f) : div9(s);
    const float x = p[1][1];
    const float d = x - blur;
    const float e = strength * d;
    return x + e;
}
VRG_HD float unsharp_value(const float p[3][3], float strength) { return clamp01(unsharp_raw<false>(p, strength)); }

VRG_HD float laplacian_raw(const float p[3][3], float strength, int zero_border) {
    const float x = p[1][1];
    float lap;
    if (!zero_border) {
        // nodes.py:278-284: W + N + S + E - 4*x
        float s = p[1][0] + p[0][1];
        s = s + p[2][1];
        s = s + p[1][2];
        const float fx = 4.0f * x;
        lap = s - fx;
    } else {
        // conv2d cross-correlation with [[0,-1,0],[-1,4,-1],[0,-1,0]] (nodes.py:248-257), raster order
        float s = (-p[0][1]) - p[1][0];
        s = s + 4.0f * x;
        s = s - p[1][2];
        lap = s - p[2][1];
    }
    const float e = strength * lap;
    return x + e;
}
VRG_HD float laplacian_value(const float p[3][3], float strength, int zero_border) { return clamp01(laplacian_raw(p, strength, zero_border)); }

VRG_HD float sobel_raw(const float p[3][3], float strength, int zero_border) {
    const float x = p[1][1];
    float gx, gy, mag;
    if (!zero_border) {
        // nodes.py:369-377
        gx = (-p[0][0]) - 2.0f * p[1][0];
        gx = gx - p[2][0];
        gx = gx + p[0][2];
        gx = gx + 2.0f * p[1][2];
        gx = gx + p[2][2];
        gy = (-p[0][0]) - 2.0f * p[0][1];
        gy = gy - p[0][2];
        gy = gy + p[2][0];
        gy = gy + 2.0f * p[2][1];
        gy = gy + p[2][2];
        const float a = gx * gx;
        const float b = gy * gy;
        mag = __builtin_sqrtf(a + b);
    } else {
        // nodes.py:325-348, raster order over the non-zero taps, + 1e-6 under the root
        gx = (-p[0][0]) + p[0][2];
        gx = gx - 2.0f * p[1][0];
        gx = gx + 2.0f * p[1][2];
        gx = gx - p[2][0];
        gx = gx + p[2][2];
        gy = (-p[0][0]) - 2.0f * p[0][1];
        gy = gy - p[0][2];
        gy = gy + p[2][0];
        gy = gy + 2.0f * p[2][1];
        gy = gy + p[2][2];
        const float a = gx * gx;
        const float b = gy * gy;
        const float c = a + b;
        mag = __builtin_sqrtf(c + 1e-6f);
    }
    const float e = strength * mag;
    return x + e;
}
VRG_HD float sobel_value(const float p[3][3], float strength, int zero_border) { return clamp01(sobel_raw(p, strength, zero_border)); }

VRG_HD float stencil_value(int op, const float p[3][3], float strength, int zero_border) {
    if (op == 0) return unsharp_value(p, strength);
    if (op == 1) return laplacian_value(p, strength, zero_border);
    return sobel_value(p, strength, zero_border);
}
// The three channels of a pixel whose taps are in [0, 1] or NaN (the colour transfer's output): the same values with the three clamps behind
// one NaN test (clamp01_3) and the unsharp mean without its Inf pass-through
VRG_HD void stencil_value3_unit(int op, const float p[3][3][3], float strength, int zero_border, float o[3]) {
    float raw[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
        raw[ch] = op == 0 ? unsharp_raw<true>(p[ch], strength) : (op == 1 ? laplacian_raw(p[ch], strength, zero_border) : sobel_raw(p[ch], strength, zero_border));
    clamp01_3(raw, o);
}

// ------------------------------------------------------------------------------------------
// pow(x, y) for normal positive finite x and a constant exponent, in fp32 "double-word" arithmetic:
//   x = 2^e * m, m in [1,2);  i = top 7 mantissa bits;  r = fma(m, invc_i, -1)        (|r| <= 2^-8, one rounding)
//   log2 x = (e + logc_hi_i) + (logc_lo_i + r*(A0 + r*(A1 + r*A2)))                   = L_hi + L_lo, L_hi exact
//   E = y*log2 x = E_hi + E_lo with E_hi = RN(y*L_hi), E_lo = fma(y, L_lo, fma(y, L_hi, -E_hi))
//   n = rint(64*E_hi);  f = (E_hi - n/64) + E_lo  (|f| <= 2^-7);  q = 2^f - 1 by a degree-4 polynomial
//   x^y = 2^(n>>6) * (T_hi[n&63] + fma(T_hi, q, T_lo))                                the last add is THE rounding
// Error before that final rounding is < 0.06 ulp, so the result is within 0.56 ulp of the true power (ocml powf
// and Sleef powf, what torch uses on HIP / CPU, are "<= 1 ulp").  ~33 fp32 issue slots + 2 LDS reads instead of
// ocml powf's ~180 instructions; no fp64.
// ------------------------------------------------------------------------------------------
#include "vrg_pow_tables.inc"

constexpr int POW_TABLE_WORDS = 128 * 4 + 64 * 2;   // fp32 words

struct PowTables {
    const float* logt;  // [128][4] = {invc, logc_hi, logc_lo, pad}
    const float* expt;  // [64][2]  = {hi, lo}
};

// copy the constexpr tables into `dst` (LDS on the device, a static array in the host checker)
VRG_HD void pow_tables_fill(float* dst, int first, int stride) {
    for (int i = first; i < POW_TABLE_WORDS; i += stride)
        dst[i] = f32_from_bits(i < 512 ? VRG_POW_LOGT[i >> 2][i & 3] : VRG_POW_EXPT[(i - 512) >> 1][(i - 512) & 1]);
}

VRG_HD float pow_pos(float x, float y, const PowTables& T) {
    union { float f; uint32_t u; } cv;
    cv.f = x;
    const uint32_t b = cv.u;
    const float e = (float)((int)(b >> 23) - 127);
    const uint32_t i = (b >> 16) & 0x7fu;
    const float m = f32_from_bits((b & 0x007fffffu) | 0x3f800000u);
    const float invc = T.logt[4 * i], lhi = T.logt[4 * i + 1], llo = T.logt[4 * i + 2];
    const float r = __builtin_fmaf(m, invc, -1.0f);
    float p = __builtin_fmaf(r, f32_from_bits(VRG_POW_A[2]), f32_from_bits(VRG_POW_A[1]));
    p = __builtin_fmaf(r, p, f32_from_bits(VRG_POW_A[0]));
    const float L_hi = e + lhi;                               // exact: lhi is a multiple of 2^-16, |e| < 128
    const float L_lo = __builtin_fmaf(r, p, llo);
    const float E_hi = y * L_hi;
    const float E_lo = __builtin_fmaf(y, L_lo, __builtin_fmaf(y, L_hi, -E_hi));
    const float nf = __builtin_rintf(E_hi * 64.0f);
    const float f = __builtin_fmaf(nf, -0.015625f, E_hi) + E_lo;
    const int n = (int)nf;
    float q = __builtin_fmaf(f, f32_from_bits(VRG_POW_B[3]), f32_from_bits(VRG_POW_B[2]));
    q = __builtin_fmaf(f, q, f32_from_bits(VRG_POW_B[1]));
    q = __builtin_fmaf(f, q, f32_from_bits(VRG_POW_B[0]));
    q = f * q;
    const float thi = T.expt[2 * (n & 63)], tlo = T.expt[2 * (n & 63) + 1];
    const float v = thi + __builtin_fmaf(thi, q, tlo);
    const float res = __builtin_ldexpf(v, n >> 6);
    return (x != x) ? x : res;
}

// ------------------------------------------------------------------------------------------
// kornia.color Lab transforms (external to the reference, restated: oracle/restated.py).  Operation
// order and rounding points are kornia's.  TWO arithmetic policies, chosen by the type of the last argument:
//
//  DevMath   ("device" -- the default of the nodes): every op is what torch-ROCm executes for it on this GPU, i.e. what
//            the reference computes when ComfyUI runs it on the MI355X (nodes.py:98-115 under get_torch_device()):
//              * tensor / python_scalar  ->  x * fl32(1.0 / c)          (ATen BinaryDivTrueKernel: a * reciprocal(b), with the
//                                            reciprocal of the PYTHON DOUBLE formed in double and rounded once: for 1.055 that
//                                            is one ulp away from 1.0f / 1.055f; measured on the MI355X, profiles/r02_cm_parity.json)
//              * torch.pow(x, y)         ->  __ocml_pow_f32(x, y)       (ATen PowKernel: ::pow -> ocml; the exponent is a
//                                            RUNTIME value there, and is one here, so no specialisation can differ)
//              * torch.pow(x, 3.0)       ->  (x * x) * x                (PowKernel's d_exp == 3 special case)
//              * tensor / tensor         ->  IEEE quotient              (xyz / white, (lab - mean) / std)
//            tests/test_gpu_parity.py holds the element-wise path BIT-EQUAL to oracle/restated.py evaluated by torch on
//            the device (Lab image, and apply with injected statistics).
//  PowTables ("fast"): IEEE quotients (= torch on the CPU) and pow_pos / cbrt_pow below (0.53 / 0.50 ulp) instead of the
//            ~190-instruction ocml powf; within a few ulp of either reference, ~2.3x the throughput.
// ------------------------------------------------------------------------------------------
struct DevMath {
    float e24, e1_24, e1_3;     // 2.4f, (float)(1/2.4), (float)(1/3.0): kernel arguments, not literals
    const float* logt;          // dev_pow_ziv's log table in LDS ([128][4], filled by the kernel), or nullptr: transcription only
};

#if defined(__HIP_DEVICE_COMPILE__)
extern "C" __device__ float __ocml_pow_f32(float, float);
#define VRG_LIB_POWF(x, y) __ocml_pow_f32((x), (y))
#define VRG_HW_FREXP_MANT(x) __builtin_amdgcn_frexp_mantf(x)     /* v_frexp_mant_f32: [0.5, 1) */
#define VRG_HW_FREXP_EXP(x) __builtin_amdgcn_frexp_expf(x)       /* v_frexp_exp_i32_f32 */
#else
#define VRG_LIB_POWF(x, y) __builtin_powf((x), (y))      /* host checker only (tests/host_math) */
#define VRG_HW_FREXP_MANT(x) __builtin_frexpf((x), &vrg_frexp_dummy_)
#define VRG_HW_FREXP_EXP(x) ::vrg::host_frexp_exp(x)
static int vrg_frexp_dummy_;
VRG_HD int host_frexp_exp(float x) { int e; (void)__builtin_frexpf(x, &e); return e; }
#endif

// ------------------------------------------------------------------------------------------
// dev_pow(x, y) for x > 0 (finite, +Inf or NaN) and finite y: ocml's powf -- the function torch.pow evaluates on the
// device -- with its special-case scaffolding removed.  __ocml_pow_f32 is  expep(y * epln(|x|))  in fp32 double-word
// arithmetic (ROCm device-libs ocml: powF.cl / eplnF.cl / expepF.cl, read from the LLVM IR of /opt/rocm/amdgcn/bitcode/
// ocml.bc) wrapped in ~70 instructions of selects for x <= 0, integer / infinite y, signs and NaNs.  Below is the SAME
// operation sequence, op for op (every add, multiply and FMA of the gfx9 "has fast FMA" path in the same order, v_rcp_f32,
// and the backend's own lowering of exp through __builtin_expf), so the result is identical by construction wherever the
// scaffolding is inert: 120 instead of ~195 instructions.  tests/test_gpu_parity.py sweeps EVERY fp32 base of the Lab
// transforms' domains against torch.pow on the device (7.4e7 - 1.1e8 inputs per exponent) and against __ocml_pow_f32.
// ------------------------------------------------------------------------------------------
// GUARD selects how much of the scaffolding that is still left below is kept -- every flavour executes the SAME arithmetic,
// a lower GUARD only drops selects that a range argument proves inert for the stated inputs (so the results are identical
// there by construction, and the exhaustive sweeps in tests/test_gpu_parity.py run every flavour the kernels use):
//   DEV_POW_ANY   x > 0 down to subnormals, +Inf, NaN; any finite y          (everything kept)
//   DEV_POW_OVF   x >= 2^-20, +Inf or NaN; 0 < y <= 4: y ln x >= -56, so exp cannot underflow and no intermediate is
//                 infinite for finite x; overflow of the result (x^2.4 for x > 1.1e16) keeps its guards
//   DEV_POW_UNIT  x >= 2^-20, +Inf or NaN; 0 < y <= 0.5: |y ln x| <= 44.4, far from ln 2^128 = 88.72: nothing can overflow either
// The exp of the high word is the backend's own expansion of exp(x) for fp32 (AMDGPU lowerFEXP, the one ocml's expep gets
// inlined): exp2(x * log2e) with the product in double-word form, v_rndne / v_exp_f32 / v_ldexp_f32, then its two range
// selects (x < -103.28 -> 0, x > 88.72 -> Inf); written out here so that the selects can be dropped where they are inert.
// ------------------------------------------------------------------------------------------
constexpr int DEV_POW_ANY = 0, DEV_POW_OVF = 1, DEV_POW_UNIT = 2;

VRG_HD float dev_exp_core(float x) {
    const float c = f32_from_bits(0x3fb8aa3bu);                      // log2(e), high word
    const float ph = x * c;
    const float f0 = __builtin_fmaf(x, c, -ph);
    const float pl = __builtin_fmaf(x, f32_from_bits(0x32a5705fu), f0);       // + x * (log2(e) - c)
    const float e = __builtin_rintf(ph);
    const float a = (ph - e) + pl;
    return __builtin_ldexpf(VRG_HW_EXP2(a), (int)e);
}

// dev_exp_core for |x| < 2^20 whose result is a normal number -- every argument the Ziv route hands it from inside its domains (|y ln x|
// <= 12 there) --: the same value with three half-rate instructions less.  rint(ph) as (ph + 1.5 * 2^23) - 1.5 * 2^23 (round to nearest
// even, exact for |ph| < 2^22: what v_rndne_f32 returns), and ldexp(v_exp_f32(a), e) as an integer addition of e to the exponent field:
// the sum's low mantissa bits ARE e in two's complement (the bit pattern is 0x4B400000 + e, and 0x4B400000 << 23 vanishes modulo 2^32),
// v_exp_f32 of a in [-0.5, 0.5] is a normal number in [0.70, 1.42], and so is the scaled result.  One v_lshl_add_u32 instead of
// v_rndne_f32 + v_cvt_i32_f32 + v_ldexp_f32 (9.3 instead of 13.4 issue units).  A NaN or out-of-range argument yields garbage here: the
// callers' rounding / domain tests fail for exactly those lanes and send them to the transcription.
VRG_HD float dev_exp_core_normal(float x) {
    const float c = f32_from_bits(0x3fb8aa3bu);                      // log2(e), high word
    const float ph = x * c;
    const float f0 = __builtin_fmaf(x, c, -ph);
    const float pl = __builtin_fmaf(x, f32_from_bits(0x32a5705fu), f0);
    const float magic = f32_from_bits(0x4b400000u);                  // 1.5 * 2^23
    const float t = ph + magic;
    const float e = t - magic;
    const float a = (ph - e) + pl;
    return f32_from_bits(f32_bits(VRG_HW_EXP2(a)) + (f32_bits(t) << 23));
}

// ocml's epln: ln(x) = ln_hi + ln_lo
template <int GUARD>
VRG_HD void dev_epln(float x, float& ln_hi_out, float& ln_lo_out) {
    // x = m * 2^e with m in [2/3, 4/3)
    float m;
    int e;
    if (GUARD == DEV_POW_ANY) {
        m = VRG_HW_FREXP_MANT(x);
        const bool low = m < f32_from_bits(0x3f2aaaabu);             // 2/3
        m = m * (low ? 2.0f : 1.0f);
        e = VRG_HW_FREXP_EXP(x) - (low ? 1 : 0);
    } else {
        // the same (m, e) for a normal x, from its bit pattern: 4 integer ops instead of frexp x 2, compare, select, multiply, borrow
        const int32_t d = (int32_t)(f32_bits(x) - 0x3f2aaaabu);
        e = d >> 23;
        m = f32_from_bits((uint32_t)(d & 0x007fffff) + 0x3f2aaaabu);
    }
    const float a12 = m + -1.0f;
    const float a13 = m + 1.0f;
    const float a14 = a13 + -1.0f;
    const float a15 = m - a14;
#if defined(__HIP_DEVICE_COMPILE__)
    const float r16 = __builtin_amdgcn_rcpf(a13);
#else
    const float r16 = 1.0f / a13;
#endif
    const float a17 = a12 * r16;
    const float a18 = a13 * a17;
    const float a25 = __builtin_fmaf(a17, a13, -a18);
    const float a44 = __builtin_fmaf(a17, a15, a25);
    const float a49 = a18 + a44;
    const float a50 = a49 - a18;
    const float a51 = a44 - a50;
    // ocml forms the rounding error of a12 - a49 as well ((a12 - a52) - a49, two more subtractions and a negated add).  a49 is
    // a12 * (1 + eps) with |eps| < 2^-21 (v_rcp_f32's ulp plus the roundings of a17, a18, a44, a49), so the difference of the two
    // is exact (Sterbenz) and that error term is +0 for every input: a56 = a52 + (0 - a51) = a52 - a51, same rounding.
    const float a52 = a12 - a49;
    const float a56 = a52 - a51;
    const float a57 = r16 * a56;
    const float a58 = a17 + a57;
    const float a59 = a58 - a17;
    const float a60 = a57 - a59;
    const float a61 = a58 * a58;
    const float a65 = __builtin_fmaf(a58, a58, -a61);
    const float a80 = a60 * 2.0f;
    const float a81 = __builtin_fmaf(a58, a80, a65);
    const float a87 = a61 + a81;
    const float a88 = a87 - a61;
    const float a89 = a81 - a88;
    const float a90 = __builtin_fmaf(a87, f32_from_bits(0x3e76c4e1u), f32_from_bits(0x3e91f4c4u));
    const float a91 = __builtin_fmaf(a87, a90, f32_from_bits(0x3ecccdefu));
    const float a92 = (float)e;
    const float ln2h = f32_from_bits(0x3f317218u);
    const float a93 = a92 * ln2h;
    const float a97 = __builtin_fmaf(a92, ln2h, -a93);
    const float a112 = __builtin_fmaf(a92, f32_from_bits(0xb102e308u), a97);
    const float a117 = a58 * a87;
    const float a121 = __builtin_fmaf(a87, a58, -a117);
    const float a140 = __builtin_fmaf(a87, a60, a121);
    const float a141 = __builtin_fmaf(a89, a58, a140);
    const float a148 = a117 + a141;
    const float a149 = a148 - a117;
    const float a150 = a141 - a149;
    const float a151 = a87 * a91;
    const float a155 = __builtin_fmaf(a87, a91, -a151);
    const float a174 = __builtin_fmaf(a89, a91, a155);
    const float a179 = a151 + a174;
    const float a180 = a179 - a151;
    const float a181 = a174 - a180;
    const float a182 = a179 + f32_from_bits(0x3f2aaaaau);
    const float a183 = a182 + f32_from_bits(0xbf2aaaaau);
    const float a184 = a179 - a183;
    const float a185 = a181 + f32_from_bits(0x31739010u);
    const float a186 = a185 + a184;
    const float a187 = a182 + a186;
    const float a188 = a187 - a182;
    const float a189 = a186 - a188;
    const float a190 = a148 * a187;
    const float a194 = __builtin_fmaf(a148, a187, -a190);
    const float a213 = __builtin_fmaf(a148, a189, a194);
    const float a214 = __builtin_fmaf(a150, a187, a213);
    const float a221 = a60 * 2.0f;                                   // ldexp(., 1): exact
    const float a222 = a58 * 2.0f;
    const float a223 = a93 + a112;
    const float a224 = a223 - a93;
    const float a225 = a112 - a224;
    const float a226 = a190 + a214;
    const float a227 = a226 - a190;
    const float a228 = a214 - a227;
    const float a229 = a222 + a226;
    const float a230 = a229 - a222;
    const float a231 = a226 - a230;
    const float a232 = a221 + a228;
    const float a233 = a232 + a231;
    const float a234 = a229 + a233;
    const float a235 = a234 - a229;
    const float a236 = a233 - a235;
    // (a237, a242) = two-sum(a223, a234).  ocml spends the six-operation form here; a223 is the head of e * ln2 -- zero, or
    // at least 0.693 in magnitude -- and a234 the head of ln(m), |ln(m)| <= 0.405 for m in [2/3, 4/3): the three-operation
    // form is error-free under exactly that ordering (or a zero first operand) and returns the same pair.
    const float a237 = a223 + a234;
    const float a238 = a237 - a223;
    const float a242 = a234 - a238;
    const float a243 = a225 + a236;
    const float a244 = a243 - a225;
    const float a245 = a243 - a244;
    const float a246 = a225 - a245;
    const float a247 = a236 - a244;
    const float a248 = a247 + a246;
    const float a249 = a243 + a242;
    const float a250 = a237 + a249;
    const float a251 = a250 - a237;
    const float a252 = a249 - a251;
    const float a253 = a248 + a252;
    const float ln_hi = a250 + a253;
    const float a255 = ln_hi - a250;
    ln_hi_out = ln_hi;
    ln_lo_out = a253 - a255;
}

template <int GUARD>
VRG_HD float dev_pow_t(float x, float y) {
    float ln_hi, ln_lo;
    dev_epln<GUARD>(x, ln_hi, ln_lo);
    // ---- y * ln(x) = ph + pl
    const float p17 = y * ln_hi;
    const float p24 = __builtin_fmaf(y, ln_hi, -p17);
    const float p44 = __builtin_fmaf(y, ln_lo, p24);
    const float p50 = p17 + p44;
    const float p51 = p50 - p17;
    const float p52 = p44 - p51;
    const float inf = __builtin_inff();
    const float ovf = f32_from_bits(0x42b17218u);                    // ln 2^128
    float ph = p50, pl = p52;
    if (GUARD == DEV_POW_ANY) {                                      // y * ln_hi overflowed
        ph = (__builtin_fabsf(p17) == inf) ? p17 : p50;
        pl = (__builtin_fabsf(ph) == inf) ? 0.0f : p52;
    }
    // ---- expep
    float h5 = ph, l7 = pl;
    if (GUARD != DEV_POW_UNIT) {
        const float c4 = (ph == ovf) ? f32_from_bits(0x37000000u) : 0.0f;
        h5 = ph - c4;
        l7 = pl + c4;
    }
    float e8 = dev_exp_core(h5);
    if (GUARD == DEV_POW_ANY) e8 = (h5 < f32_from_bits(0xc2ce8ed0u)) ? 0.0f : e8;       // -103.279: the backend's underflow select
    if (GUARD != DEV_POW_UNIT) e8 = (h5 > ovf) ? inf : e8;
    const float r9 = __builtin_fmaf(e8, l7, e8);
    const float r = (GUARD != DEV_POW_UNIT && __builtin_fabsf(e8) == inf) ? e8 : r9;
    // ocml's scaffolding for the inputs this function admits: pow(+Inf, y) = y > 0 ? Inf : 0; NaN propagates by itself
    // through the frexp form, and is passed on explicitly by the bit-pattern form
    if (GUARD == DEV_POW_ANY) return (x == inf) ? (y > 0.0f ? inf : 0.0f) : r;
    return (x < inf) ? r : ((x == inf) ? (y > 0.0f ? inf : 0.0f) : x);
}
VRG_HD float dev_pow(float x, float y) { return dev_pow_t<DEV_POW_ANY>(x, y); }

// ------------------------------------------------------------------------------------------
// dev_pow_ziv: the SAME value as dev_pow_t -- ocml's powf -- through a cheaper route wherever that route provably cannot
// differ, and through the transcription itself everywhere else (a Ziv-style rounding test, per lane).
//
// ocml's result is  RN(e8 * (1 + pl))  with  e8 = exp(ph),  (ph, pl) = the double-word y * ln x:  it depends on the double-word
// logarithm only through (a) the fp32 head ph = RN(y ln x) and (b) the last rounding; everything ocml spends on carrying ln x to
// ~2^-45 matters only for the rare arguments where one of those two roundings sits within the logarithm's error of a tie.
// So: ln x from a 128-entry table (tools/make_ziv_log_table.py: ln x = e ln2 + T_j + log1p(r), r = m c_j - 1 EXACT in one FMA,
// r^2 kept as a double word, the r^3.. tail in fp32; |error| <= 2^-37.2 |ln x| and <= 2^-39 absolute on the domains below, measured
// over every fp32 argument against a float64 log by tools/ziv_log_accuracy.py -- ocml's own epln: 2^-34.7 / 2^-36), then ocml's own
// y * ln x product, backend exp and final FMA, with the two roundings evaluated at BOTH ends of the interval
// [y ln x - delta, y ln x + delta], delta = the measured maximum distance between ocml's logarithm and this one (ziv_delta):
// if the heads agree and the results agree, ocml's head and result lie between equal numbers.  Otherwise -- 0.1-0.3 % of the
// lanes -- and for arguments outside [lo, hi] (the ranges the tests sweep exhaustively, per exponent), the lane runs
// dev_pow_t.  tests/test_gpu_parity.py compares this function with torch.pow for EVERY fp32 base of [lo, hi] for each of the
// three exponents, so inside the fast path's domain equality is established by enumeration, outside it by construction.
// ------------------------------------------------------------------------------------------
#include "vrg_ziv_log_table.inc"

constexpr int ZIV_TABLE_WORDS = 128 * 4;

VRG_HD void ziv_table_fill(float* dst, int first, int stride) {
    for (int i = first; i < ZIV_TABLE_WORDS; i += stride) dst[i] = f32_from_bits(VRG_ZIV_LOGT[i >> 2][i & 3]);
}

// (Lh, Ll) = ln x for a normal positive x; T = the table in LDS.  The pair is NOT normalised (Lh is not the rounded head: |Ll| can
// reach 2^-20 |Lh|): its only consumer is the double-word product y * (Lh + Ll) of ziv_try, which does not need that, and the
// three operations of the final renormalisation are saved.  Both two-sums use the three-operation form where its ordering
// condition holds for every argument: |e ln2 + T_j| >= |r - r^2/2| whenever the former is not zero (e != 0: >= 0.28; e == 0: the
// table's smallest non-zero |T_j| is 0.0078 against max |r| 0.0051 in those intervals -- asserted for every (e, j) by
// tools/make_ziv_log_table.py), and |r| >= r^2/2.
VRG_HD void ziv_log(float x, const float* T, float& Lh, float& Ll, float& Eh_out, float& A_out) {
    const int32_t d = (int32_t)(f32_bits(x) - 0x3f2aaaabu);
    // e * 2^23 = d with its low 23 bits cleared (one full-rate v_and_b32 for the half-rate shift), converted exactly; the factor 2^-23
    // sits in the two constants it is multiplied by -- the same products
    const uint32_t e23 = (uint32_t)d & 0xff800000u;
    const float ef23 = (float)(int32_t)e23;
    const float m = f32_from_bits(f32_bits(x) - e23);                // x * 2^-e in [2/3, 4/3): (d - e23) + 0x3f2aaaab in one subtraction
    const float* t = T + (((uint32_t)d >> 16) & 0x7fu) * 4;          // index j = bits 16 .. 22 of d
    const float c = t[0], th = t[1], tl = t[2];
    A_out = t[3];
    const float r = __builtin_fmaf(m, c, -1.0f);                     // exact (7-bit c)
    const float Eh = ef23 * f32_from_bits(0x33b17200u);              // e * ln2 head (15 bits: exact product); 0x33b17200 = 0x3f317200 * 2^-23
    const float h = r * r;
    const float l = __builtin_fmaf(r, r, -h);                        // r^2 = h + l
    float P = __builtin_fmaf(r, (float)(-1.0 / 6.0), 0.2f);
    P = __builtin_fmaf(r, P, -0.25f);
    P = __builtin_fmaf(r, P, (float)(1.0 / 3.0));
    const float tail = __builtin_fmaf(-0.5f, l, (h * r) * P);        // -l/2 + r^3/3 - r^4/4 + r^5/5 - r^6/6
    const float s1 = Eh + th;                                        // EXACT for the exponents of the domains (e = -8 .. 2): th lies on the 2^-21 grid (make_ziv_log_table.py)
    const float s2 = __builtin_fmaf(-0.5f, h, r);                    // r - h/2 (h/2 is exact): |r| >= |h/2|, fast two-sum
    const float e2 = __builtin_fmaf(-0.5f, h, r - s2);               // (-h/2) - (s2 - r)
    const float s3 = s1 + s2;                                        // s1 = 0 or |s1| >= |s2|: fast two-sum
    const float e3 = s2 - (s3 - s1);
    float low = __builtin_fmaf(ef23, f32_from_bits(0x2a3fbe8eu), tl);       // e * (ln2 - head) + T_lo; 0x2a3fbe8e = 0x35bfbe8e * 2^-23
    low = low + e3;
    low = low + e2;
    low = low + tail;
    Lh = s3;
    Ll = low;
    Eh_out = Eh;
}

// Half-width of the interval that contains ocml's y ln x around this function's: y * |ln x (ocml) - ln x (table)|.  The two
// logarithms are deterministic functions of x, so their distance has an exact maximum over a finite domain.  Almost all of it is
// ocml's own error (its epln is good to 2^-34.7, the table log to 2^-37.3), a smooth function of m: the table's T_lo_j carries the
// midpoint of that distance over the arguments with table index j (so the table tracks OCML's logarithm), and its fourth word holds A_j =
// 1.25 x the largest distance left for index j (divided by the relative bound's constant, see VRG_ZIV_REL_BITS), measured over EVERY
// fp32 of [0.0031308, 4] (tools/ziv_per_index.py ->
// tools/ziv_calibration.json; 2^-38.8 for the median index, 2^-36.2 for the worst, where one global bound used to stand at
// 2^-35.7).  Near x = 1 -- the indexes around m = 1 carry no bias -- the distance is relative to |ln x|: the second bound, 1.25 x the
// measured global maximum relative to max(|e ln2|, |ln x|) (2^-35.25 with this table).  It is not optional: saturated pixels
// (v = 1.0 -> q = 0.9999995, 2 % of the lanes of a graded frame) sit there, and without it pass 1 ran the transcription in every
// wave (VRG_ZIV_REL = 0: 928 instead of 709 instructions per pixel).  (The +-delta additions and ocml's own last roundings are
// five orders of magnitude smaller.)
#define VRG_ZIV_REL 1
// The relative bound's constant C = 2^-34.92 = 1.25 x the measured maximum.  The table's fourth word holds A_j / C (rounded up), so that
// the half-width is C * min(|Lh| + 64 |Eh|, A_j / C) and the factor C joins y in ONE loop-invariant product (ziv_try): no multiply here.
#define VRG_ZIV_REL_BITS 0x2e06f428u
// (returns the half-width of the LOGARITHM's interval in units of C; ziv_try scales it by y * C inside the two FMAs that form the interval's ends)
VRG_HD float ziv_delta(float Lh, float Eh, float A_over_C) {
#if VRG_ZIV_REL
    // max(|Lh|, |Eh|) only decides for e = 0 (Eh = 0): for e != 0 |Eh| >= 0.69, the relative bound is above every A_j (<= 2^-36) and the minimum
    // picks A_j.  |Lh| + 64 |Eh| -- one full-rate FMA with |.| modifiers for the half-rate v_max_f32 -- is |Lh| for e = 0 and >= 44 otherwise: the same bound.
    // (a NaN here fails the rounding test anyway)
    const float rel = __builtin_fmaf(__builtin_fabsf(Eh), 64.0f, __builtin_fabsf(Lh));
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(rel, A_over_C, 0.0f);    // min of two non-negative numbers; v_min_f32 would first canonicalise the table word (a v_max_f32 A, A)
#else
    return __builtin_fminf(rel, A_over_C);
#endif
#else
    (void)Lh; (void)Eh;
    return A_over_C;
#endif
}

// The fast route alone: returns the candidate and whether the rounding test (and the domain test) passed.
// IN_DOMAIN: the caller guarantees that x lies in [lo, hi] or is NaN (a NaN fails the rounding test by itself): no domain test.
template <bool IN_DOMAIN = false>
VRG_HD bool ziv_try(float x, float y, const float* T, uint32_t lo_bits, uint32_t hi_bits, float& out) {
    float Lh, Ll, Eh, A;
    ziv_log(x, T, Lh, Ll, Eh, A);
    const float p17 = y * Lh;
    const float p24 = __builtin_fmaf(y, Lh, -p17);
    const float p44 = __builtin_fmaf(y, Ll, p24);
    const float yc = y * f32_from_bits(VRG_ZIV_REL_BITS);            // loop-invariant (y is a kernel argument)
    const float dl = ziv_delta(Lh, Eh, A);
    const float up = __builtin_fmaf(yc, dl, p44), dn = __builtin_fmaf(-yc, dl, p44);    // p44 +- y * half-width, one rounding each
    const float php = p17 + up;
    const float phm = p17 + dn;
    const float t = php - p17;                                       // exact; ocml's tail is y ln x - head = (p44 -+ ...) - t
    const float e8 = dev_exp_core_normal(php);
    const float rp = __builtin_fmaf(e8, up - t, e8);
    const float rm = __builtin_fmaf(e8, dn - t, e8);
    out = rp;
    if (IN_DOMAIN) return (php == phm) & (rp == rm);
    return ((f32_bits(x) - lo_bits) <= (hi_bits - lo_bits)) & (php == phm) & (rp == rm);      // bitwise: no control flow here
}

template <int GUARD>
VRG_HD float dev_pow_ziv(float x, float y, const float* T, uint32_t lo_bits, uint32_t hi_bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (T) {
        float r;
        if (ziv_try(x, y, T, lo_bits, hi_bits, r)) return r;
#if defined(VRG_LAB_VARIANT_SOURCE) && defined(LAB_ZIV_NO_FALLBACK)      /* tools/ab: the cost of the transcription branches (wrong bits in 0.1-0.3 % of the lanes) */
        return r;
#endif
    }
#else
    (void)T; (void)lo_bits; (void)hi_bits;
#endif
    return dev_pow_t<GUARD>(x, y);
}

// Three powers with one exponent (the three channels of a Lab transform): the three fast routes first, as straight-line code --
// three independent chains for the scheduler to interleave, the three table reads in flight together -- then the (rare)
// transcription per channel.  Same values as three dev_pow_ziv calls.
// Contract: every x[c] >= the domain's lower end lo (the callers' pow_base_min(., lo)); IN_DOMAIN: also <= its upper end.
template <int GUARD, bool IN_DOMAIN = false>
VRG_HD void dev_pow_ziv3(const float x[3], float y, const float* T, uint32_t lo_bits, uint32_t hi_bits, float o[3]) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (T) {
        // the three bases arrive clamped from below to the domain's lower end (pow_base_min at every call site): what is left of the domain
        // test is ONE comparison of the largest of them with the upper end -- two instructions for the triple instead of two per channel
        // (a triple with one base above the domain sends its three channels to the transcription: same values)
        float r0, r1, r2;
        const bool dom = IN_DOMAIN || __builtin_fmaxf(__builtin_fmaxf(x[0], x[1]), x[2]) <= f32_from_bits(hi_bits);
        const bool s0 = ziv_try<true>(x[0], y, T, lo_bits, hi_bits, r0) & dom;
        const bool s1 = ziv_try<true>(x[1], y, T, lo_bits, hi_bits, r1) & dom;
        const bool s2 = ziv_try<true>(x[2], y, T, lo_bits, hi_bits, r2) & dom;
        o[0] = r0; o[1] = r1; o[2] = r2;
#if defined(VRG_LAB_VARIANT_SOURCE) && defined(LAB_ZIV_NO_FALLBACK)
        if (s0 | s1 | s2 | true) return;
#endif
        if (!s0) o[0] = dev_pow_t<GUARD>(x[0], y);
        if (!s1) o[1] = dev_pow_t<GUARD>(x[1], y);
        if (!s2) o[2] = dev_pow_t<GUARD>(x[2], y);
        return;
    }
#else
    (void)T; (void)lo_bits; (void)hi_bits;
#endif
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = dev_pow_t<GUARD>(x[c], y);
}

// x / c for a Python-scalar c (written as a double literal): fast = the IEEE quotient by (float)c; device = x * (float)(1.0 / c)
VRG_HD float cm_div_scalar(float x, float c, float rc, float, const PowTables&) { return div_const(x, c, rc); }
VRG_HD float cm_div_scalar(float x, float, float, float rc_dev, const DevMath&) { return x * rc_dev; }
#define VRG_CM_DIVS(x, c, M) ::vrg::cm_div_scalar((x), (float)(c), 1.0f / (float)(c), (float)(1.0 / (double)(c)), (M))
// x / c for a tensor-valued constant c (the D65 white point): the IEEE quotient under both policies.  The fast policy takes the
// FMA form outright; the device policy takes it for a whole wave when every active lane's |x| lies in the range over which the
// form is PROVEN equal to x / c for these constants (1e-30 .. 1e30: all 2^32 inputs swept on the device, vrg_selftest_divconst) or
// is zero (0 / c = 0 either way; the sign of a zero does not survive lab_f's 7.787 * t + 4/29), and the backend's IEEE division
// sequence (~10 instructions plus v_rcp_f32, 13 issue slots) otherwise -- 6 slots instead of 13 for every ordinary pixel.
VRG_HD float cm_div_tensor(float x, float c, float rc, const PowTables&) { return div_const(x, c, rc); }
VRG_HD float cm_div_tensor(float x, float c, float rc, const DevMath&) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool proven = ((__builtin_fabsf(x) - 1e-30f) <= (1e30f - 1e-30f)) | (x == 0.0f);
    if (__builtin_amdgcn_ballot_w64(!proven) == 0) return div_const(x, c, rc);
#else
    (void)rc;
#endif
    return x / c;
}
#define VRG_CM_DIVT(x, c, M) ::vrg::cm_div_tensor((x), (c), 1.0f / (c), (M))
// X / Xn and Z / Zn of a pixel whose RGB is in [0, 1 + 2^-22] or NaN (rgb_to_lab_unit): both numerators are sums of non-negative products, at
// most 1.0001 -- never negative, never -0, never above the proven range.  What is left of the range test is "not in (0, 1e-30)": on the bit
// patterns of non-negative numbers, (bits - 1) >= bits(1e-30) - 1 as UNSIGNED integers (+0 wraps to the top; a NaN passes, and the FMA form
// hands a NaN on as the division does): an add and a compare per value instead of a subtraction and two compares; one ballot for the pair.
VRG_HD void cm_div_white_unit(float X, float Z, float& Xo, float& Zo, const PowTables& M) {
    Xo = cm_div_tensor(X, 0.95047f, 1.0f / 0.95047f, M);
    Zo = cm_div_tensor(Z, 1.08883f, 1.0f / 1.08883f, M);
}
VRG_HD void cm_div_white_unit(float X, float Z, float& Xo, float& Zo, const DevMath&) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t k = 0x0da24260u - 1u;                             // bits(1e-30f) - 1
    const bool proven = ((f32_bits(X) - 1u) >= k) & ((f32_bits(Z) - 1u) >= k);
    if (__builtin_amdgcn_ballot_w64(!proven) == 0) {
        Xo = div_const(X, 0.95047f, 1.0f / 0.95047f);
        Zo = div_const(Z, 1.08883f, 1.0f / 1.08883f);
        return;
    }
#endif
    Xo = X / 0.95047f;
    Zo = Z / 1.08883f;
}

VRG_HD float srgb_to_linear(float v, const PowTables& T) {
    const float t = v + 0.055f;
    const float q = VRG_DIVC(t, 1.055f);
    // the power is only selected for v > 0.04045 (q > 0.09): keep its argument in pow_pos's domain
    const float hi = pow_pos(pow_base_min(q, 0.0625f), 2.4f, T);
    const float lo = VRG_DIVC(v, 12.92f);
    return v > 0.04045f ? hi : lo;
}
VRG_HD float srgb_to_linear(float v, const DevMath& M) {
    const float t = v + 0.055f;
    const float q = VRG_CM_DIVS(t, 1.055, M);
    // (the reference evaluates pow on every element and selects afterwards: for v <= 0.04045 the value is discarded, so the
    //  base only has to stay in dev_pow's domain there)
    const float hi = dev_pow_ziv<DEV_POW_OVF>(pow_base_min(q, 0.0625f), M.e24, M.logt, 0x3d800000u, 0x40000000u);        // fast path on [0.0625, 2]
    const float lo = VRG_CM_DIVS(v, 12.92, M);
    return v > 0.04045f ? hi : lo;
}

// the three channels at once (same values as three calls; the device policy phases its powers, see dev_pow_ziv3)
// UNIT (device policy): the caller guarantees v in [0, 1 + 2^-22] or NaN -- the output of grain's or the cube's clamp, blended or not --,
// so the power's base lies in its fast-path domain [0.0625, 2] (at most 1.0000003) and the domain test is dropped (ziv_try<IN_DOMAIN>)
template <bool UNIT = false>
VRG_HD void srgb_to_linear3(const float v[3], float o[3], const PowTables& T) {
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = srgb_to_linear(v[c], T);
}
template <bool UNIT = false>
VRG_HD void srgb_to_linear3(const float v[3], float o[3], const DevMath& M) {
    float q[3], hi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = pow_base_min(VRG_CM_DIVS(v[c] + 0.055f, 1.055, M), 0.0625f);
    dev_pow_ziv3<DEV_POW_OVF, UNIT>(q, M.e24, M.logt, 0x3d800000u, 0x40000000u, hi);                                  // fast path on [0.0625, 2]
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = v[c] > 0.04045f ? hi[c] : VRG_CM_DIVS(v[c], 12.92, M);
}

VRG_HD float linear_to_srgb(float v, const PowTables& T) {
    const float thr = 0.0031308f;
    const float base = pow_base_min(v, thr);
    const float pw = pow_pos(base, (float)(1.0 / 2.4), T);
    const float hi = 1.055f * pw - 0.055f;
    const float lo = 12.92f * v;
    return v > thr ? hi : lo;
}
VRG_HD float linear_to_srgb(float v, const DevMath& M) {
    const float thr = 0.0031308f;
    const float base = pow_base_min(v, thr);
    const float pw = dev_pow_ziv<DEV_POW_UNIT>(base, M.e1_24, M.logt, 0x3b4d2e1cu, 0x40800000u);                      // [0.0031308, 4]
    const float hi = 1.055f * pw - 0.055f;
    const float lo = 12.92f * v;
    return v > thr ? hi : lo;
}

// x ** float(1/3) for x in [0.008856, ~1.2] (the Lab cube root: kornia's torch.pow(xyz, 1/3) with the exponent rounded
// to fp32).  Hardware estimate t0 = exp2(y * log2 x) (v_log_f32 / v_exp_f32, ~1e-6 relative), then ONE Newton step
// on t^3 = x * x^(3*eps) with the residual x - t0^3 formed exactly (FMA, with the rounding error of t0*t0 carried
// along) and eps = float(1/3) - 1/3 folded in through the log already at hand.  The step is quadratic, so the result
// is the estimate's error squared away from the true power and then rounded once: 0.500 ulp maximum error, equal to
// the correctly rounded power in every one of 4e6 sampled inputs (pow_pos: 0.534 ulp, 0.2-0.3 % one-ulp differences),
// in 20 issue units instead of 33 and without the LDS tables.
VRG_HD float cbrt_pow(float x) {
    const float y = (float)(1.0 / 3.0);
    const float L = VRG_HW_LOG2(x);
    const float t0 = VRG_HW_EXP2(L * y);
    const float t2 = t0 * t0;
    const float e2 = __builtin_fmaf(t0, t0, -t2);                   // t0^2 = t2 + e2 exactly
    float r = __builtin_fmaf(-t2, t0, x);                           // x - t2*t0, one rounding
    r = __builtin_fmaf(-e2, t0, r);                                 // ... - e2*t0
    const float k = (float)(3.0 * ((double)(float)(1.0 / 3.0) - 1.0 / 3.0) * 0.6931471805599453);   // 3*eps*ln2
    r = __builtin_fmaf(x, L * k, r);                                // target is x^(1+3*eps) = x*(1 + 3*eps*ln x)
    const float d = (r * VRG_HW_RCP(x)) * (float)(1.0 / 3.0);       // relative correction r / (3*t0^3), t0^3 ~ x
    return __builtin_fmaf(t0, d, t0);
}

VRG_HD void linear_to_srgb3(const float v[3], float o[3], const PowTables& T) {
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = linear_to_srgb(v[c], T);
}
VRG_HD void linear_to_srgb3(const float v[3], float o[3], const DevMath& M) {
    const float thr = 0.0031308f;
    float base[3], pw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) base[c] = pow_base_min(v[c], thr);
    dev_pow_ziv3<DEV_POW_UNIT>(base, M.e1_24, M.logt, 0x3b4d2e1cu, 0x40800000u, pw);                                  // [0.0031308, 4]
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = v[c] > thr ? 1.055f * pw[c] - 0.055f : 12.92f * v[c];
}

VRG_HD float lab_cbrt(float t, const PowTables&) { return cbrt_pow(t); }
VRG_HD float lab_cbrt(float t, const DevMath& M) { return dev_pow_ziv<DEV_POW_UNIT>(t, M.e1_3, M.logt, 0x3c1118c2u, 0x40800000u); }   // [0.008856, 4]

template <class MATH>
VRG_HD float lab_f(float t, const MATH& T) {
    const float thr = 0.008856f;
    const float pw = lab_cbrt(pow_base_min(t, thr), T);
    const float sc = 7.787f * t + (float)(4.0 / 29.0);
    return t > thr ? pw : sc;
}

// UNIT (device policy): t = XYZ / white of a pixel whose RGB is in [0, 1 + 2^-22] or NaN: at most 1.000001, inside [0.008856, 4] after the clamp
template <bool UNIT = false>
VRG_HD void lab_f3(const float t[3], float o[3], const PowTables& T) {
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = lab_f(t[c], T);
}
template <bool UNIT = false>
VRG_HD void lab_f3(const float t[3], float o[3], const DevMath& M) {
    const float thr = 0.008856f;
    float base[3], pw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) base[c] = pow_base_min(t[c], thr);
    dev_pow_ziv3<DEV_POW_UNIT, UNIT>(base, M.e1_3, M.logt, 0x3c1118c2u, 0x40800000u, pw);                             // [0.008856, 4]
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = t[c] > thr ? pw[c] : 7.787f * t[c] + (float)(4.0 / 29.0);
}

VRG_HD float dot3(float a, float x, float b, float y, float c, float z) {
    const float p = a * x;
    const float q = b * y;
    const float r = c * z;
    const float s = p + q;
    return s + r;
}

template <bool UNIT, class MATH>
VRG_HD void rgb_to_lab_t(const float rgb[3], float lab[3], const MATH& T) {
    float lin[3];
    srgb_to_linear3<UNIT>(rgb, lin, T);
    const float r = lin[0], g = lin[1], b = lin[2];
    float X, Z;
    const float Xs = dot3(0.412453f, r, 0.357580f, g, 0.180423f, b), Zs = dot3(0.019334f, r, 0.119193f, g, 0.950227f, b);
    if (UNIT) {
        cm_div_white_unit(Xs, Zs, X, Z, T);
    } else {
        X = VRG_CM_DIVT(Xs, 0.95047f, T);
        Z = VRG_CM_DIVT(Zs, 1.08883f, T);
    }
    const float Y = dot3(0.212671f, r, 0.715160f, g, 0.072169f, b);   // / 1.0
    const float xyz[3] = {X, Y, Z};
    float f[3];
    lab_f3<UNIT>(xyz, f, T);
    const float fx = f[0], fy = f[1], fz = f[2];
    lab[0] = 116.0f * fy - 16.0f;
    const float dxy = fx - fy;
    const float dyz = fy - fz;
    lab[1] = 500.0f * dxy;
    lab[2] = 200.0f * dyz;
}
template <class MATH>
VRG_HD void rgb_to_lab(const float rgb[3], float lab[3], const MATH& T) { rgb_to_lab_t<false>(rgb, lab, T); }
// the same values for a pixel that left grain's or the cube's clamp (RGB in [0, 1 + 2^-22] or NaN): the powers skip their domain tests
template <class MATH>
VRG_HD void rgb_to_lab_unit(const float rgb[3], float lab[3], const MATH& T) { rgb_to_lab_t<true>(rgb, lab, T); }

template <class MATH>
VRG_HD float lab_finv(float f, const MATH& T) {
    const float cube = (f * f) * f;
    const float d = f - (float)(4.0 / 29.0);
    const float sc = VRG_CM_DIVS(d, 7.787, T);
    return f > 0.2068966f ? cube : sc;
}

template <class MATH>
VRG_HD void lab_to_rgb(const float lab[3], float rgb[3], const MATH& T) {
    const float l16 = lab[0] + 16.0f;
    const float fy = VRG_CM_DIVS(l16, 116.0, T);
    const float a5 = VRG_CM_DIVS(lab[1], 500.0, T);
    const float fx = a5 + fy;
    const float b2 = VRG_CM_DIVS(lab[2], 200.0, T);
    const float fzr = fy - b2;
    const float fz = clamp_min(fzr, 0.0f);
    const float X = lab_finv(fx, T) * 0.95047f;
    const float Y = lab_finv(fy, T);   // * 1.0
    const float Z = lab_finv(fz, T) * 1.08883f;
    const float lr = dot3((float)3.2404813432005266, X, (float)-1.5371515162713185, Y, (float)-0.4985363261688878, Z);
    const float lg = dot3((float)-0.9692549499965682, X, (float)1.8759900014898907, Y, (float)0.0415559265582928, Z);
    const float lb = dot3((float)0.0556466391351772, X, (float)-0.2040413383665112, Y, (float)1.0573110696453443, Z);
    const float lin[3] = {lr, lg, lb};
    float s3[3];
    linear_to_srgb3(lin, s3, T);
    clamp01_3(s3, rgb);
}

// matched = (lab-mu)/sigma*sigma_ref + mu_ref ; blended = K*matched + T*lab (nodes.py:112-113)
// fast: d / sigma with sigma = std + 1e-5 (a per-frame value in [1e-5, ~100]) through the FMA form of div_const, with the
// reciprocal from v_rcp_f32 + one Newton step: 8 issue units instead of the 14 of the IEEE sequence, and the same
// quotient in 2e7 random (d, sigma) pairs.  device: the IEEE quotient itself (tensor / tensor).
VRG_HD float recip_newton(float s) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float e = __builtin_fmaf(-s, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
#else
    return 1.0f / s;
#endif
}
VRG_HD float cm_div_sigma(float d, float sigma, const PowTables&) { return div_const(d, sigma, recip_newton(sigma)); }
VRG_HD float cm_div_sigma(float d, float sigma, const DevMath&) { return d / sigma; }

// ------------------------------------------------------------------------------------------
// (lab - mu) / sigma under the device policy is the IEEE quotient: the backend's sequence (AMDGPU LowerFDIV32, as hipcc emits it and as
// torch's kernel runs it) is
//     bs = v_div_scale(sigma), as = v_div_scale(d);  y0 = v_rcp_f32(bs);  e = fma(-bs, y0, 1);  y1 = fma(e, y0, y0);
//     q = as * y1;  r = fma(-bs, q, as);  q1 = fma(r, y1, q);  r1 = fma(-bs, q1, as);  v_div_fmas(r1, y1, q1);  v_div_fixup
// i.e. 12 instructions of which two scalings, the v_div_fmas and the fix-up issue at half rate and the reciprocal at a quarter (37 issue
// units, profiles/r06_valu_instruction_costs.json).  sigma is ONE value per (frame, channel): y1 is a per-frame constant, and when the
// operands cannot trigger a scaling or a fix-up -- v_div_scale_f32 returns its operand and clears VCC, v_div_fmas_f32 is then a plain FMA,
// v_div_fixup_f32 returns its first operand (ISA pseudocode) -- the quotient is the five full-rate operations q .. fma(r1, y1, q1) on
// the unscaled operands: the same roundings, 12 issue units.  That holds when
//     2^-40 <= sigma <= 2^30 and 2^-60 <= |mu| <= 2^40      (per frame: SigmaRecip::usable; then d = lab - mu is +0 or |d| >= 2^-85 --
//                                                            Sterbenz inside [mu/2, 2 mu], >= |mu| / 2 outside -- never -0, never tiny:
//                                                            exponent(d) > 23, d / sigma >= 2^-115 is normal, 1 / sigma is normal)
//     |d_L| + |d_a| + |d_b| < 2^40                          (per pixel, one wave-uniform branch: exponent(d) - exponent(sigma) < 96, nothing
//                                                            overflows; an Inf or NaN in any channel fails the comparison)
// and d == +0 gives +0 through the FMAs as through the fix-up.  Everything else takes the division itself.  tests/test_gpu_parity.py
// compares the two forms on 2^27 random (d, sigma, mu) triples and at the edges of the conditions.
// ------------------------------------------------------------------------------------------
struct SigmaRecip { float y1[3]; bool usable; };

VRG_HD SigmaRecip sigma_recip(const float* img_ms) {
    SigmaRecip R;
    R.usable = true;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float mu = __builtin_fabsf(img_ms[2 * c]), s = img_ms[2 * c + 1];
#if defined(__HIP_DEVICE_COMPILE__)
        const float y0 = __builtin_amdgcn_rcpf(s);
#else
        const float y0 = 1.0f / s;
#endif
        const float e = __builtin_fmaf(-s, y0, 1.0f);
        R.y1[c] = __builtin_fmaf(e, y0, y0);
        R.usable = R.usable && s >= 0x1p-40f && s <= 0x1p30f && mu >= 0x1p-60f && mu <= 0x1p40f;
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    R.usable = false;        // the host build (tests/host_math) has no v_rcp_f32: the division itself
#endif
    return R;
}

VRG_HD float div_sigma_unscaled(float d, float s, float y1) {
    const float q = d * y1;
    const float r = __builtin_fmaf(-s, q, d);
    const float q1 = __builtin_fmaf(r, y1, q);
    const float r1 = __builtin_fmaf(-s, q1, d);
    return __builtin_fmaf(r1, y1, q1);
}

// z = (lab - mu) / sigma for the three channels
VRG_HD void cm_normalise3(const float lab[3], const float* img_ms, float z[3], const PowTables& M, const SigmaRecip*) {
#pragma unroll
    for (int c = 0; c < 3; ++c) z[c] = cm_div_sigma(lab[c] - img_ms[2 * c], img_ms[2 * c + 1], M);
}
VRG_HD void cm_normalise3(const float lab[3], const float* img_ms, float z[3], const DevMath& M, const SigmaRecip* SR) {
    float d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = lab[c] - img_ms[2 * c];
#if defined(__HIP_DEVICE_COMPILE__)
    if (SR && SR->usable) {
        const float mag = (__builtin_fabsf(d[0]) + __builtin_fabsf(d[1])) + __builtin_fabsf(d[2]);
        if (__builtin_amdgcn_ballot_w64(!(mag < 0x1p40f)) == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) z[c] = div_sigma_unscaled(d[c], img_ms[2 * c + 1], SR->y1[c]);
            return;
        }
    }
#else
    (void)SR;
#endif
#pragma unroll
    for (int c = 0; c < 3; ++c) z[c] = cm_div_sigma(d[c], img_ms[2 * c + 1], M);
}

// ms: {mean, std+1e-5} per channel.  Lab of the pixel -> matched, blended, back to RGB.
// SR: the frame's SigmaRecip (device policy only), or nullptr: every quotient by the division
template <class MATH>
VRG_HD void colormatch_from_lab(const float lab[3], const float* img_ms, const float* ref_ms, float K, float T, float o[3],
                                const MATH& PT, const SigmaRecip* SR = nullptr) {
    float z[3], bl[3];
    cm_normalise3(lab, img_ms, z, PT, SR);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float w = z[c] * ref_ms[2 * c + 1];
        const float m = w + ref_ms[2 * c];
        const float a = K * m;
        const float b = T * lab[c];
        bl[c] = a + b;
    }
    lab_to_rgb(bl, o, PT);
    // final .clamp(0,1) of nodes.py:121 is idempotent after lab_to_rgb's clip
}

template <class MATH>
VRG_HD void colormatch_pixel(const float rgb[3], const float* img_ms, const float* ref_ms, float K, float T, float o[3],
                             const MATH& PT, const SigmaRecip* SR = nullptr) {
    float lab[3];
    rgb_to_lab(rgb, lab, PT);
    colormatch_from_lab(lab, img_ms, ref_ms, K, T, o, PT, SR);
}

}  // namespace vrg
