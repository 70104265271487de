// vrg_adjust_math.hpp -- per-pixel arithmetic of the 13-slider Adjust (_apply_adjust_tensor,
// VRGDG_LUTVideoTools.py:307-391 of the reference), one fp32 rounding per reference op.  Shared by the HIP
// kernels (vrg_adjust.hip) and by the host arithmetic-order check (tests/host_math).
#pragma once
#include "vrg_pixel_math.hpp"

namespace vrg {

struct AdjustK {
    int32_t enabled;
    float shift[3];
    float exposure, contrast, saturation;
    float highlights, shadows, whites, blacks;      // slider/220, slider/240
    int32_t has_clarity, has_sharpen, has_fade, has_vignette;
    float clarity, sharpen;                          // slider/100
    float fade_mul, fade_add;                        // (1 - fade*0.35), fade*0.18
    float vignette;                                  // slider/100
    int32_t div_device;                              // tensor / python scalar as torch runs it on the GPU: x * fl32(1.0 / c)
    int32_t box;                                     // clarity box size k (odd, 3..9) or < 3: no blur
    float step_y, step_x;                            // torch.linspace steps 2/(H-1), 2/(W-1) (fp32 quotients)
};

// x / c for the Adjust constants 0.45, 1.05 and the box areas 9, 25, 49, 81 through the FMA form of
// vrg_pixel_math.hpp (div_const): equal to the IEEE quotient for every x with 1e-30 <= |x| <= 1e30 and for 0
// (up to the sign of zero, which no later op observes) -- swept over all 2^32 inputs for each constant.
// Outside that range (never reached from frames in [0,1], kept for completeness) the IEEE division runs.
#define VRG_ADJ_DIV(x, c) ::vrg::adjust_div((x), (c), 1.0f / (c))
VRG_HD float adjust_div(float x, float c, float rc) {
    const float ax = __builtin_fabsf(x);
    if (__builtin_expect(!((ax >= 1e-30f && ax <= 1e30f) || ax == 0.0f), 0)) return x / c;   // also NaN
    return div_const(x, c, rc);
}

// The three `tensor / python scalar` divisions of the reference that are not by a power of two (/ 0.45 twice, / 1.05).
// The reference runs where its `device` argument says: on the CPU torch divides (IEEE quotient); on the GPU ATen multiplies
// by the reciprocal of the Python double rounded to fp32 (BinaryDivTrueKernel; measured on the MI355X, profiles/
// r02_cm_parity.json) -- one ulp apart now and then.  div_device selects which of the two references is reproduced.
#define VRG_ADJ_DIVS(A, x, c) ((A).div_device ? (x) * (float)(1.0 / (double)(c)) : ::vrg::adjust_div((x), (float)(c), 1.0f / (float)(c)))

VRG_HD float luma3(float r, float g, float b) {
    const float a = r * 0.2126f;
    const float c = g * 0.7152f;
    const float d = b * 0.0722f;
    const float s = a + c;
    return s + d;
}

// point stage (:316-340)
VRG_HD void adjust_point(const AdjustK& A, const float x[3], float o[3]) {
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = clamp01(x[c]);
        t = t + A.shift[c];
        t = t * A.exposure;
        t = t - 0.5f;
        t = t * A.contrast;
        v[c] = t + 0.5f;
    }
    const float gray = luma3(v[0], v[1], v[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float d = v[c] - gray;
        const float e = d * A.saturation;
        v[c] = gray + e;
    }
    const float luma = luma3(v[0], v[1], v[2]);
    // x / 0.25 == x * 4 exactly (power of two)
    const float hm = clamp01(VRG_ADJ_DIVS(A, luma - 0.55f, 0.45)) * A.highlights;
    const float sm = clamp01(VRG_ADJ_DIVS(A, 0.45f - luma, 0.45)) * A.shadows;
    const float wm = clamp01((luma - 0.75f) * 4.0f) * A.whites;
    const float bm = clamp01((0.25f - luma) * 4.0f) * A.blacks;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = v[c] + hm;
        t = t + sm;
        t = t + wm;
        o[c] = t + bm;
    }
}

// torch.linspace(-1, 1, n)[i] in fp32: ATen evaluates from both ends (RangeFactories: idx < n/2 ? start + step*idx
// : end - step*(n-idx-1)), step = 2/(n-1) as an fp32 quotient, and its build contracts the multiply-add into one
// FMA -- verified against torch.linspace for every n <= 300 and the video heights / widths (tests/test_host_math.py).
VRG_HD float linspace_step(int n) { return n > 1 ? 2.0f / (float)(n - 1) : 0.0f; }
VRG_HD float linspace_pm1(int i, int n, float step) {
    if (n <= 1) return -1.0f;
    if (i < n / 2) return __builtin_fmaf(step, (float)i, -1.0f);
    return __builtin_fmaf(-step, (float)(n - i - 1), 1.0f);
}

// tail (:377-391): fade, vignette, clamp
VRG_HD void adjust_tail(const AdjustK& A, int y, int x, int H, int W, float v[3]) {
    if (A.has_fade) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = v[c] * A.fade_mul;
            v[c] = t + A.fade_add;
        }
    }
    if (A.has_vignette) {
        const float yy = linspace_pm1(y, H, A.step_y), xx = linspace_pm1(x, W, A.step_x);
        const float a = xx * xx;
        const float b = yy * yy;
        const float dist = __builtin_sqrtf(a + b);
        const float e = dist - 0.35f;
        const float q = clamp01(VRG_ADJ_DIVS(A, e, 1.05));
        const float m0 = q * A.vignette;
        const float m1 = m0 * 0.75f;
        const float mask = 1.0f - m1;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = v[c] * mask;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = clamp01(v[c]);
}

// clarity mix (:346-356): x + (x - blur) * clarity * 1.55 * (0.35 + midtone * 0.65)
VRG_HD void adjust_clarity_mix(const AdjustK& A, const float ctr[3], const float blur[3], float v[3]) {
    const float lum = luma3(ctr[0], ctr[1], ctr[2]);
    const float t = __builtin_fabsf(lum - 0.5f) * 2.0f;      // x / 0.5 == x * 2 exactly
    const float mid = 1.0f - clamp01(t);
    const float w0 = mid * 0.65f;
    const float w = 0.35f + w0;
    for (int c = 0; c < 3; ++c) {
        const float d = ctr[c] - blur[c];
        const float e = d * A.clarity;
        const float f = e * 1.55f;
        const float g = f * w;
        v[c] = ctr[c] + g;
    }
}

// sharpen mix (:358-361): x + (x - blur) * sharpen * 5
VRG_HD void adjust_sharpen_mix(const AdjustK& A, const float ctr[3], const float blur[3], float v[3]) {
    for (int c = 0; c < 3; ++c) {
        const float d = ctr[c] - blur[c];
        const float e = d * A.sharpen;
        const float f = e * 5.0f;
        v[c] = ctr[c] + f;
    }
}

// clarity box size (:349-352): min(9, odd(H), odd(W)); < 3 means "no blur"
VRG_HD int adjust_box_size(int H, int W) {
    const int hk = H % 2 ? H : H - 1, wk = W % 2 ? W : W - 1;
    int box = 9;
    if (hk < box) box = hk;
    if (wk < box) box = wk;
    return box;
}

}  // namespace vrg
