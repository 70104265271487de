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
#endif

namespace vrg {

VRG_HD float f32_from_bits(uint32_t b) {
    union { uint32_t u; float f; } c;
    c.u = b;
    return c.f;
}

// clamp(v, 0, 1) with torch.clamp / np.clip NaN behaviour (NaN stays NaN).
VRG_HD float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
// clamp(v, min=lo)
VRG_HD float clamp_min(float v, float lo) { return v < lo ? lo : v; }

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

VRG_D float bm_radius(uint32_t a) {
    const float two_m32 = f32_from_bits(0x2f800000u);  // ROCRAND_2POW32_INV
    const float u = __builtin_fmaf((float)a, two_m32, two_m32);
    return __builtin_sqrtf(-2.0f * log_of_unit_uniform(u));
}

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

VRG_HD void grain_pixel(const float x[3], const float n[3], float I, float S, float T, float o[3]) {
    o[0] = grain_element(x[0], n[0], n[1], 0, I, S, T);
    o[1] = grain_element(x[1], n[1], n[1], 1, I, S, T);
    o[2] = grain_element(x[2], n[2], n[1], 2, I, S, T);
}

// ------------------------------------------------------------------------------------------
// 3D LUT (VRGDG_IV_Adjustments.py:293-336, blend :355-359)
// ------------------------------------------------------------------------------------------
struct LutParams {
    const float* table;  // [N][N][N][3], index [b][g][r]
    int n;
    float top;           // (float)(N-1)
    float dmin[3];
    float span[3];       // max(dmax - dmin, 1e-6f)
    int unit_domain;     // dmin == 0 and span == 1 for all channels: (x-0)/1 == x exactly
    int blend_mode;      // 1 = LUT only, 2 = x*(1-B) + y*B
    float blend, one_minus_blend;
};

struct LutAxis { int i0, i1; float f, u; };

VRG_HD LutAxis lut_axis(float x, float dmin, float span, int unit_domain, float top, int n) {
    float t;
    if (unit_domain) {
        t = x;
    } else {
        const float d = x - dmin;
        t = d / span;
    }
    t = clamp01(t);
    const float c = t * top;
    const float fl = __builtin_floorf(c);
    int i0 = (int)fl;
    i0 = i0 < 0 ? 0 : (i0 > n - 1 ? n - 1 : i0);  // only reachable for NaN input (reference: undefined)
    LutAxis a;
    a.i0 = i0;
    a.i1 = i0 + 1 > n - 1 ? n - 1 : i0 + 1;
    a.f = c - (float)i0;
    a.u = 1.0f - a.f;
    return a;
}

VRG_HD float lerp2(float a, float wa, float b, float wb) {
    const float pa = a * wa;
    const float pb = b * wb;
    return pa + pb;
}

// rgb in -> graded rgb out (before the strength blend)
VRG_HD void lut_pixel_raw(const LutParams& P, const float x[3], float y[3]) {
    const LutAxis R = lut_axis(x[0], P.dmin[0], P.span[0], P.unit_domain, P.top, P.n);
    const LutAxis Gx = lut_axis(x[1], P.dmin[1], P.span[1], P.unit_domain, P.top, P.n);
    const LutAxis B = lut_axis(x[2], P.dmin[2], P.span[2], P.unit_domain, P.top, P.n);
    const int n = P.n;
    const float* b0g0 = P.table + (size_t)((B.i0 * n + Gx.i0) * n) * 3;
    const float* b1g0 = P.table + (size_t)((B.i1 * n + Gx.i0) * n) * 3;
    const float* b0g1 = P.table + (size_t)((B.i0 * n + Gx.i1) * n) * 3;
    const float* b1g1 = P.table + (size_t)((B.i1 * n + Gx.i1) * n) * 3;
    const int r0 = R.i0 * 3, r1 = R.i1 * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float c00 = lerp2(b0g0[r0 + ch], B.u, b1g0[r0 + ch], B.f);
        const float c01 = lerp2(b0g1[r0 + ch], B.u, b1g1[r0 + ch], B.f);
        const float c10 = lerp2(b0g0[r1 + ch], B.u, b1g0[r1 + ch], B.f);
        const float c11 = lerp2(b0g1[r1 + ch], B.u, b1g1[r1 + ch], B.f);
        const float c0 = lerp2(c00, Gx.u, c01, Gx.f);
        const float c1 = lerp2(c10, Gx.u, c11, Gx.f);
        y[ch] = clamp01(lerp2(c0, R.u, c1, R.f));
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
VRG_HD float unsharp_value(const float p[3][3], float strength) {
    // nodes.py:194-207 (numpy) and avg_pool2d's running sum share this left-assoc raster order.
    float s = p[0][0] + p[0][1];
    s = s + p[0][2];
    s = s + p[1][0];
    s = s + p[1][1];
    s = s + p[1][2];
    s = s + p[2][0];
    s = s + p[2][1];
    s = s + p[2][2];
    const float blur = s / 9.0f;
    const float x = p[1][1];
    const float d = x - blur;
    const float e = strength * d;
    return clamp01(x + e);
}

VRG_HD float laplacian_value(const float p[3][3], float strength, int zero_border) {
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
    return clamp01(x + e);
}

VRG_HD float sobel_value(const float p[3][3], float strength, int zero_border) {
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
    return clamp01(x + e);
}

VRG_HD float stencil_value(int op, const float p[3][3], float strength, int zero_border) {
    if (op == 0) return unsharp_value(p, strength);
    if (op == 1) return laplacian_value(p, strength, zero_border);
    return sobel_value(p, strength, zero_border);
}

// ------------------------------------------------------------------------------------------
// kornia.color Lab transforms (external to the reference, restated: oracle/restated.py)
// ------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define VRG_POWF(a, b) powf(a, b)  /* ocml pow, what torch-HIP's pow kernel calls */
#else
#define VRG_POWF(a, b) __builtin_powf(a, b)
#endif

VRG_HD float srgb_to_linear(float v) {
    const float t = v + 0.055f;
    const float q = t / 1.055f;
    const float hi = VRG_POWF(q, 2.4f);
    const float lo = v / 12.92f;
    return v > 0.04045f ? hi : lo;
}

VRG_HD float linear_to_srgb(float v) {
    const float thr = 0.0031308f;
    const float base = clamp_min(v, thr);
    const float pw = VRG_POWF(base, (float)(1.0 / 2.4));
    const float hi = 1.055f * pw - 0.055f;
    const float lo = 12.92f * v;
    return v > thr ? hi : lo;
}

VRG_HD float lab_f(float t) {
    const float thr = 0.008856f;
    const float pw = VRG_POWF(clamp_min(t, thr), (float)(1.0 / 3.0));
    const float sc = 7.787f * t + (float)(4.0 / 29.0);
    return t > thr ? pw : sc;
}

VRG_HD float dot3(float a, float x, float b, float y, float c, float z) {
    const float p = a * x;
    const float q = b * y;
    const float r = c * z;
    const float s = p + q;
    return s + r;
}

VRG_HD void rgb_to_lab(const float rgb[3], float lab[3]) {
    const float r = srgb_to_linear(rgb[0]);
    const float g = srgb_to_linear(rgb[1]);
    const float b = srgb_to_linear(rgb[2]);
    const float X = dot3(0.412453f, r, 0.357580f, g, 0.180423f, b) / 0.95047f;
    const float Y = dot3(0.212671f, r, 0.715160f, g, 0.072169f, b) / 1.0f;
    const float Z = dot3(0.019334f, r, 0.119193f, g, 0.950227f, b) / 1.08883f;
    const float fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
    lab[0] = 116.0f * fy - 16.0f;
    const float dxy = fx - fy;
    const float dyz = fy - fz;
    lab[1] = 500.0f * dxy;
    lab[2] = 200.0f * dyz;
}

VRG_HD float lab_finv(float f) {
    const float cube = (f * f) * f;
    const float d = f - (float)(4.0 / 29.0);
    const float sc = d / 7.787f;
    return f > 0.2068966f ? cube : sc;
}

VRG_HD void lab_to_rgb(const float lab[3], float rgb[3]) {
    const float l16 = lab[0] + 16.0f;
    const float fy = l16 / 116.0f;
    const float a5 = lab[1] / 500.0f;
    const float fx = a5 + fy;
    const float b2 = lab[2] / 200.0f;
    const float fzr = fy - b2;
    const float fz = clamp_min(fzr, 0.0f);
    const float X = lab_finv(fx) * 0.95047f;
    const float Y = lab_finv(fy) * 1.0f;
    const float Z = lab_finv(fz) * 1.08883f;
    const float lr = dot3((float)3.2404813432005266, X, (float)-1.5371515162713185, Y, (float)-0.4985363261688878, Z);
    const float lg = dot3((float)-0.9692549499965682, X, (float)1.8759900014898907, Y, (float)0.0415559265582928, Z);
    const float lb = dot3((float)0.0556466391351772, X, (float)-0.2040413383665112, Y, (float)1.0573110696453443, Z);
    rgb[0] = clamp01(linear_to_srgb(lr));
    rgb[1] = clamp01(linear_to_srgb(lg));
    rgb[2] = clamp01(linear_to_srgb(lb));
}

// matched = (lab-mu)/sigma*sigma_ref + mu_ref ; blended = K*matched + T*lab (nodes.py:112-113)
VRG_HD float colormatch_channel(float lab, float mu, float sigma, float mu_ref, float sigma_ref, float K, float T) {
    const float d = lab - mu;
    const float z = d / sigma;
    const float w = z * sigma_ref;
    const float m = w + mu_ref;
    const float a = K * m;
    const float b = T * lab;
    return a + b;
}

// ms: {mean, std+1e-5} per channel
VRG_HD void colormatch_pixel(const float rgb[3], const float* img_ms, const float* ref_ms, float K, float T, float o[3]) {
    float lab[3], bl[3];
    rgb_to_lab(rgb, lab);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        bl[c] = colormatch_channel(lab[c], img_ms[2 * c], img_ms[2 * c + 1], ref_ms[2 * c], ref_ms[2 * c + 1], K, T);
    lab_to_rgb(bl, o);
    // final .clamp(0,1) of nodes.py:121 is idempotent after lab_to_rgb's clip
}

}  // namespace vrg
