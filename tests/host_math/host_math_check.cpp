// Arithmetic-order checker (TEST SCAFFOLDING, built by tests/test_host_math.py with g++, no GPU).
//
// Instantiates the kernels' per-pixel math header (csrc/vrg_pixel_math.hpp) on the host so that
// operation order, rounding points, index arithmetic and the Philox integer pipeline can be compared
// with the oracle before any GPU time is spent.  It is NOT a CPU fallback: nothing in the package can
// load it, and the three hardware transcendentals of the Box-Muller transform are replaced by libm
// stand-ins (so normals are only checked to a tolerance here; the bit-exact check is -m gpu).
#include <math.h>
#include <stdint.h>
#include <string.h>
#define VRG_HW_LOG2(x) log2f(x)
#define VRG_HW_SIN_REV(x) sinf((x) * 6.28318530717958647692f)
#define VRG_HW_COS_REV(x) cosf((x) * 6.28318530717958647692f)
#define VRG_HW_EXP2(x) exp2f(x)
#define VRG_HW_RCP(x) (1.0f / (x))
#include "vrg_pixel_math.hpp"
#include "vrg_adjust_math.hpp"

using namespace vrg;

static const PowTables& host_tables() {
    static float store[POW_TABLE_WORDS];
    static PowTables T{store, store + 512};
    static bool init = false;
    if (!init) { pow_tables_fill(store, 0, 1); init = true; }
    return T;
}

extern "C" {

void hm_pow(const float* x, float* o, int64_t n, double y) {
    const PowTables& T = host_tables();
    for (int64_t i = 0; i < n; ++i) o[i] = pow_pos(x[i], (float)y, T);
}

// Markstein division vs IEEE: returns the number of bit mismatches among n inputs (NaN == NaN)
int64_t hm_divc_mismatches(const float* x, int64_t n, int which) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        float got, want;
        const float v = x[i];
        switch (which) {
            case 0: got = VRG_DIVC(v, 1.055f); want = v / 1.055f; break;
            case 1: got = VRG_DIVC(v, 12.92f); want = v / 12.92f; break;
            case 2: got = VRG_DIVC(v, 0.95047f); want = v / 0.95047f; break;
            case 3: got = VRG_DIVC(v, 1.08883f); want = v / 1.08883f; break;
            case 4: got = VRG_DIVC(v, 116.0f); want = v / 116.0f; break;
            case 5: got = VRG_DIVC(v, 500.0f); want = v / 500.0f; break;
            case 6: got = VRG_DIVC(v, 200.0f); want = v / 200.0f; break;
            case 7: got = VRG_DIVC(v, 7.787f); want = v / 7.787f; break;
            case 9: got = VRG_ADJ_DIV(v, 0.45f); want = v / 0.45f; break;
            case 10: got = VRG_ADJ_DIV(v, 1.05f); want = v / 1.05f; break;
            case 11: got = VRG_ADJ_DIV(v, 81.0f); want = v / 81.0f; break;
            default: got = div9(v); want = v / 9.0f; break;
        }
        uint32_t a, b;
        memcpy(&a, &got, 4); memcpy(&b, &want, 4);
        if (a != b && !(got != got && want != want) && !(got == 0.0f && want == 0.0f)) ++bad;
    }
    return bad;
}


void hm_philox(uint64_t seed, uint64_t subsequence, uint64_t counter, uint32_t out[4]) {
    const u32x4 r = philox_for(seed, subsequence, counter);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

void hm_randn(uint64_t seed, uint64_t offset, uint32_t G, int64_t numel, float* out) {
    for (int64_t i = 0; i < numel; ++i) out[i] = torch_randn_element(seed, offset, G, (uint64_t)i);
}

void hm_grain(const float* x, const float* n, float* o, int64_t pixels, float I, float S, float T) {
    for (int64_t p = 0; p < pixels; ++p) grain_pixel(x + 3 * p, n + 3 * p, I, S, T, o + 3 * p);
}

void hm_lut(const float* x, float* o, int64_t pixels, const float* table, int n, const float* dmin, const float* dmax,
            int blend_mode, float blend, float one_minus_blend) {
    const int nc = n - 1;
    float* cells = new float[(size_t)nc * nc * n * LUT_REC_FLOATS + 4];
    float* aligned = (float*)(((uintptr_t)cells + 15) & ~(uintptr_t)15);
    for (int b = 0; b < nc; ++b)
        for (int g = 0; g < nc; ++g)
            for (int r = 0; r < n; ++r) lut_build_record(table, n, b, g, r, aligned + (size_t)((b * nc + g) * n + r) * LUT_REC_FLOATS);
    LutParams P;
    P.cells = aligned; P.n = n; P.top = (float)(n - 1); P.unit_domain = 1;
    for (int c = 0; c < 3; ++c) {
        P.dmin[c] = dmin[c];
        const float span = dmax[c] - dmin[c];
        P.span[c] = span < 1e-6f ? 1e-6f : span;
        if (!(P.dmin[c] == 0.0f && P.span[c] == 1.0f)) P.unit_domain = 0;
    }
    P.blend_mode = blend_mode; P.blend = blend; P.one_minus_blend = one_minus_blend;
    for (int64_t p = 0; p < pixels; ++p) lut_pixel(P, x + 3 * p, o + 3 * p);
    delete[] cells;
}

void hm_stencil(const float* in, float* out, int F, int H, int W, int C, int op, int zero_border, float strength) {
    for (int f = 0; f < F; ++f)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < C; ++c) {
                    float p[3][3];
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            int yy = y + dy, xx = x + dx;
                            const bool inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
                            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                            const float v = in[(((int64_t)f * H + yy) * W + xx) * C + c];
                            p[dy + 1][dx + 1] = (zero_border && !inside) ? 0.0f : v;
                        }
                    out[(((int64_t)f * H + y) * W + x) * C + c] = stencil_value(op, p, strength, zero_border);
                }
}

// device colour-match policy (DevMath) compiled for the host: v_rcp_f32 / the backend's exp lowering are replaced by IEEE 1/x and
// libm expf, so this checks the STRUCTURE of the ocml powf transcription and of the reciprocal-multiply divisions (a wrong
// constant, a dropped term or a swapped operand shows as an error of many ulps), not bit-equality -- that is the GPU suite's job.
static vrg::DevMath host_dev_math_() { return vrg::DevMath{(float)2.4, (float)(1.0 / 2.4), (float)(1.0 / 3.0)}; }
void hm_dev_pow(const float* x, float* o, int64_t n, float y) {
    for (int64_t i = 0; i < n; ++i) o[i] = vrg::dev_pow(x[i], y);
}
// dev_exp_core against its integer-scaling form (round 6): the magic-number rint and the exponent-field add are plain C, so the two must be
// bit-equal on the host wherever the form's preconditions hold
void hm_exp_cores(const float* x, float* plain, float* normal, int64_t n) {
    for (int64_t i = 0; i < n; ++i) { plain[i] = vrg::dev_exp_core(x[i]); normal[i] = vrg::dev_exp_core_normal(x[i]); }
}
// The Ziv route on the host: the candidate of the table logarithm, whether its rounding test passes, and the transcription's value.  The
// half-widths A_j are calibrated against the DEVICE's logarithm (v_rcp_f32 inside ocml's epln; here an IEEE reciprocal), so equality of
// the passing lanes is expected in all but a handful of arguments: what the CPU suite holds is the table and the arithmetic around it.
void hm_ziv(const float* x, float* cand, float* passed, float* transcription, int64_t n, float y, uint32_t lo_bits, uint32_t hi_bits) {
    static float table[ZIV_TABLE_WORDS];
    static bool init = false;
    if (!init) { ziv_table_fill(table, 0, 1); init = true; }
    for (int64_t i = 0; i < n; ++i) {
        float r;
        passed[i] = vrg::ziv_try(x[i], y, table, lo_bits, hi_bits, r) ? 1.0f : 0.0f;
        cand[i] = r;
        transcription[i] = vrg::dev_pow_t<vrg::DEV_POW_UNIT>(x[i], y);
    }
}
// (lab - mean) / std in the unscaled form (round 6), with the host's IEEE reciprocal where the device has v_rcp_f32
void hm_div_sigma(const float* d, const float* sd, float* unscaled, float* ieee, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        const float ms[6] = {1.0f, sd[i], 1.0f, sd[i], 1.0f, sd[i]};
        const vrg::SigmaRecip R = vrg::sigma_recip(ms);
        unscaled[i] = vrg::div_sigma_unscaled(d[i], sd[i], R.y1[0]);
        ieee[i] = d[i] / sd[i];
    }
}
void hm_rgb_to_lab_dev(const float* x, float* o, int64_t pixels) {
    for (int64_t p = 0; p < pixels; ++p) rgb_to_lab(x + 3 * p, o + 3 * p, host_dev_math_());
}
void hm_lab_to_rgb_dev(const float* x, float* o, int64_t pixels) {
    for (int64_t p = 0; p < pixels; ++p) lab_to_rgb(x + 3 * p, o + 3 * p, host_dev_math_());
}

void hm_rgb_to_lab(const float* x, float* o, int64_t pixels) {
    for (int64_t p = 0; p < pixels; ++p) rgb_to_lab(x + 3 * p, o + 3 * p, host_tables());
}

void hm_lab_to_rgb(const float* x, float* o, int64_t pixels) {
    for (int64_t p = 0; p < pixels; ++p) lab_to_rgb(x + 3 * p, o + 3 * p, host_tables());
}

// ms arrays: [3][2] = {mean, std+1e-5}
void hm_colormatch(const float* x, float* o, int64_t pixels, const float* img_ms, const float* ref_ms, float K, float T) {
    for (int64_t p = 0; p < pixels; ++p) colormatch_pixel(x + 3 * p, img_ms, ref_ms, K, T, o + 3 * p, host_tables());
}

// Adjust, frame by frame with plain loops: point stage -> (clarity: k x k reflect box, raster-order sum) ->
// (sharpen: 3x3 replicate box) -> tail.  t = the 20 descriptor terms in vrg_adjust_desc order.
void hm_adjust(const float* in, float* out, int F, int H, int W, const float* t) {
    AdjustK A{};
    A.enabled = (int)t[0];
    A.shift[0] = t[1]; A.shift[1] = t[2]; A.shift[2] = t[3];
    A.exposure = t[4]; A.contrast = t[5]; A.saturation = t[6];
    A.highlights = t[7]; A.shadows = t[8]; A.whites = t[9]; A.blacks = t[10];
    A.has_clarity = (int)t[11]; A.clarity = t[12]; A.has_sharpen = (int)t[13]; A.sharpen = t[14];
    A.has_fade = (int)t[15]; A.fade_mul = t[16]; A.fade_add = t[17]; A.has_vignette = (int)t[18]; A.vignette = t[19];
    A.box = adjust_box_size(H, W);
    A.step_y = linspace_step(H); A.step_x = linspace_step(W);
    const int64_t n = (int64_t)H * W * 3;
    float* a = new float[n];
    float* b = new float[n];
    for (int f = 0; f < F; ++f) {
        const float* src = in + f * n;
        float* dst = out + f * n;
        if (!A.enabled) {
            for (int64_t i = 0; i < n; ++i) dst[i] = clamp01(src[i]);
            continue;
        }
        for (int64_t p = 0; p < (int64_t)H * W; ++p) adjust_point(A, src + 3 * p, a + 3 * p);
        if (A.has_clarity && A.box >= 3) {
            const int r = A.box / 2;
            const float kk = (float)(A.box * A.box);
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float blur[3];
                    for (int c = 0; c < 3; ++c) {
                        float acc = 0.0f;
                        for (int dy = -r; dy <= r; ++dy)
                            for (int dx = -r; dx <= r; ++dx) {
                                int yy = y + dy, xx = x + dx;
                                if (yy < 0) yy = -yy;
                                if (yy > H - 1) yy = 2 * (H - 1) - yy;
                                if (xx < 0) xx = -xx;
                                if (xx > W - 1) xx = 2 * (W - 1) - xx;
                                acc = acc + a[((int64_t)yy * W + xx) * 3 + c];
                            }
                        blur[c] = A.box == 9 ? VRG_ADJ_DIV(acc, 81.0f) : acc / kk;
                    }
                    adjust_clarity_mix(A, a + ((int64_t)y * W + x) * 3, blur, b + ((int64_t)y * W + x) * 3);
                }
            memcpy(a, b, n * sizeof(float));
        }
        if (A.has_sharpen) {
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float blur[3];
                    for (int c = 0; c < 3; ++c) {
                        float acc = 0.0f;
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                int yy = y + dy, xx = x + dx;
                                yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                                xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                                acc = acc + a[((int64_t)yy * W + xx) * 3 + c];
                            }
                        blur[c] = div9(acc);
                    }
                    adjust_sharpen_mix(A, a + ((int64_t)y * W + x) * 3, blur, b + ((int64_t)y * W + x) * 3);
                }
            memcpy(a, b, n * sizeof(float));
        }
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float* v = a + ((int64_t)y * W + x) * 3;
                adjust_tail(A, y, x, H, W, v);
                dst[((int64_t)y * W + x) * 3 + 0] = v[0];
                dst[((int64_t)y * W + x) * 3 + 1] = v[1];
                dst[((int64_t)y * W + x) * 3 + 2] = v[2];
            }
    }
    delete[] a;
    delete[] b;
}

void hm_cbrt_pow(const float* x, float* o, int64_t n) { for (int64_t i = 0; i < n; ++i) o[i] = cbrt_pow(x[i]); }

// uint8 codec edge
void hm_u8_to_unit(const uint8_t* in, float* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = unit_from_u8(in[i]); }
void hm_unit_to_u8(const float* in, uint8_t* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = u8_from_unit(in[i]); }

}  // extern "C"
