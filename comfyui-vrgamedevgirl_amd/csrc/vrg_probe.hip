// vrg_probe.hip -- self-tests, element-wise probes and timing probes (include/vrgdg_hip_debug.h).  NOT part of the drop-in boundary:
// exhaustive device sweeps that prove an arithmetic substitution (constant divisions, the Box-Muller radius' square root), the pieces
// of the colour-match arithmetic one value at a time (compared with the torch op the reference executes), and the measurement probes
// DESIGN.md / LABNOTES.md quote (LUT record fetch patterns, streaming-copy ceiling, VALU issue rates).  gfx950 only.
#include "vrg_common.hpp"
#include "vrg_lanes.hpp"
#include "vrg_tstats_config.hpp"

namespace vrg {

// Exhaustive device check of div_const / div9 against the IEEE quotient: one thread per fp32 bit pattern.
// counts[w]   = mismatches with 1e-30 <= |x| <= 1e30 for constant w (must be 0),
// counts[9+w] = mismatches outside that range (documented: tiny / huge / Inf inputs).
__global__ __launch_bounds__(256) void k_selftest_divconst(unsigned long long* counts, uint32_t first) {
    const uint32_t bits = first + blockIdx.x * 256u + threadIdx.x;
    const float x = f32_from_bits(bits);
    if (x != x) return;
    const float a = __builtin_fabsf(x);
    const bool mid = a >= 1e-30f && a <= 1e30f;
    float got[9], want[9];
    got[0] = VRG_DIVC(x, 1.055f);   want[0] = x / 1.055f;
    got[1] = VRG_DIVC(x, 12.92f);   want[1] = x / 12.92f;
    got[2] = VRG_DIVC(x, 0.95047f); want[2] = x / 0.95047f;
    got[3] = VRG_DIVC(x, 1.08883f); want[3] = x / 1.08883f;
    got[4] = VRG_DIVC(x, 116.0f);   want[4] = x / 116.0f;
    got[5] = VRG_DIVC(x, 500.0f);   want[5] = x / 500.0f;
    got[6] = VRG_DIVC(x, 200.0f);   want[6] = x / 200.0f;
    got[7] = VRG_DIVC(x, 7.787f);   want[7] = x / 7.787f;
    got[8] = div9(x);               want[8] = x / 9.0f;
#pragma unroll
    for (int w = 0; w < 9; ++w) {
        const bool same = (got[w] == want[w]) || (got[w] != got[w] && want[w] != want[w]);
        if (!same) atomicAdd(&counts[(mid ? 0 : 9) + w], 1ull);
    }
}

// Exhaustive device check of the trimmed square root of the Box-Muller radius: one thread per 32-bit Philox word.
// counts[0] = inputs where bm_radius (sqrt_normal_range) != the backend's IEEE sqrt of the same argument (must be 0).
__global__ __launch_bounds__(256) void k_selftest_bm_radius(unsigned long long* counts, uint32_t first) {
    const uint32_t a = first + blockIdx.x * 256u + threadIdx.x;
    const float x = bm_radius_arg(a);
    const float got = bm_radius(a), want = __builtin_sqrtf(x);
    if (!(got == want) || !(x >= 0.0f)) atomicAdd(&counts[0], 1ull);
}

// ---- micro-benchmark of LUT record fetch patterns (timing only; results are checksums, not pixels) ----------
//  mode 0: every lane reads its own 96-B record as 6 x 16 B                       (what k_lut3d does)
//  mode 1: every lane reads half of its record (3 x 16 B)                          (is the L1 request rate the bound?)
//  mode 2: quad-cooperative: the 4 lanes of a quad read 64 contiguous bytes of ONE record per round
//          (rounds: chunks 0-3 of pixels 0,1,2,3; then chunks 4,5 of pixels 0|1 and 2|3) -- same bytes as mode 0
//  mode 3: like 0 with 64-B record stride (cells buffer must be sized for it)
__device__ __forceinline__ int dbg_quad_bcast(int v, int pattern) {   // pattern: compile-time quad_perm
    switch (pattern) {
        case 0: return __builtin_amdgcn_update_dpp(0, v, 0x00, 0xf, 0xf, false);   // [0,0,0,0]
        case 1: return __builtin_amdgcn_update_dpp(0, v, 0x55, 0xf, 0xf, false);   // [1,1,1,1]
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0xAA, 0xf, 0xf, false);   // [2,2,2,2]
        case 3: return __builtin_amdgcn_update_dpp(0, v, 0xFF, 0xf, 0xf, false);   // [3,3,3,3]
        case 4: return __builtin_amdgcn_update_dpp(0, v, 0x50, 0xf, 0xf, false);   // [0,0,1,1]
        default: return __builtin_amdgcn_update_dpp(0, v, 0xFA, 0xf, 0xf, false);  // [2,2,3,3]
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_dbg_lut_fetch(const px3* __restrict__ in, float* __restrict__ out, int64_t pixels,
                                                         const float* __restrict__ cells, int n) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pc = p < pixels ? p : pixels - 1;
    const px3 v = in[pc];
    const float top = (float)(n - 1);
    const LutAxis R = lut_axis(v.r, 0.f, 1.f, 1, top), G = lut_axis(v.g, 0.f, 1.f, 1, top), B = lut_axis(v.b, 0.f, 1.f, 1, top);
    const int nc = n - 1;
    // MODE 4: cell-major table, one 128-byte aligned record per (b0, g0, r0) cell = all 24 corner values of the cell
    const int cell = MODE == 4 ? (B.cell * nc + G.cell) * nc + R.cell : (B.cell * nc + G.cell) * n + R.cell;
    constexpr int STRIDE = MODE == 4 ? 32 : (MODE == 3 ? 16 : 12);   // floats per record
    float acc = 0.0f;
    if (MODE == 0 || MODE == 3 || MODE == 1 || MODE == 4) {
        const f32x4* q = reinterpret_cast<const f32x4*>(cells + (size_t)cell * STRIDE);
        constexpr int NQ = MODE == 1 ? 3 : 6;
#pragma unroll
        for (int i = 0; i < NQ; ++i) { const f32x4 t = q[i]; acc += (t.x + t.y) + (t.z + t.w); }
    } else {
        const int ql = threadIdx.x & 3;
#pragma unroll
        for (int rnd = 0; rnd < 6; ++rnd) {
            const int served = dbg_quad_bcast(cell, rnd);                       // cell of the pixel this round serves
            const int chunk = rnd < 4 ? ql : 4 + (ql & 1);
            const f32x4 t = *reinterpret_cast<const f32x4*>(cells + (size_t)served * STRIDE + chunk * 4);
            acc += (t.x + t.y) + (t.z + t.w);
        }
    }
    if (p < pixels) out[p] = acc;
}

// Round 4 probes of the same record run (96 B per pixel, 48-byte records):
//  mode 9 : ONE 16-byte request per lane (piece 0)            -- one line per lane-instruction, 1 lane-request per pixel
//  mode 10: TWO requests per lane (pieces 0 and 5)            -- the L2 -> L1 line traffic of mode 0 with a third of its lane-requests
//  mode 11: the six pieces of a lane's own run through LDS-DMA (global_load_lds_dwordx4: no VGPR destination), read back with ds_read_b128
//  mode 12: quad-cooperative LDS-DMA: in round p = 0..3 the four lanes of a quad fetch the first 64 bytes of the run of the quad's
//           pixel p (one 64-byte segment per quad and instruction), rounds 4 / 5 fetch the last 32 bytes of two pixels each; the LDS-DMA
//           lays every round out lane-linear, so pixel 4q+j finds its pieces at [round j][4q + c] and [4 + j/2][4q + 2 (j&1) + c]
//           -- the transposition costs no VALU and no VGPR.  Round stride 1040 B keeps the ds_read_b128 groups conflict-free.
typedef const __attribute__((address_space(1))) void* dbg_gptr;
typedef __attribute__((address_space(3))) void* dbg_lptr;
constexpr int DBG_ROUND_BYTES = 1040;

template <int MODE>
__global__ __launch_bounds__(256) void k_dbg_lut_fetch_r4(const px3* __restrict__ in, float* __restrict__ out, int64_t pixels,
                                                            const float* __restrict__ cells, int n) {
    __shared__ __attribute__((aligned(16))) char slots[4][6][DBG_ROUND_BYTES];
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pc = p < pixels ? p : pixels - 1;
    const px3 v = in[pc];
    const float top = (float)(n - 1);
    const LutAxis R = lut_axis(v.r, 0.f, 1.f, 1, top), G = lut_axis(v.g, 0.f, 1.f, 1, top), B = lut_axis(v.b, 0.f, 1.f, 1, top);
    const int nc = n - 1;
    // MODE 19: mode 12's quad-cooperative LDS-DMA over the CELL-MAJOR table (one 128-byte aligned record per cell: one line per pixel)
    constexpr int QSTRIDE = MODE == 19 ? 32 : 12;
    const int cell = MODE == 19 ? (B.cell * nc + G.cell) * nc + R.cell : (B.cell * nc + G.cell) * n + R.cell;
    float acc = 0.0f;
    if (MODE >= 13 && MODE <= 18) {
        // the six 16-byte pieces per lane (mode 0's requests) with cache-policy bits on the loads: 13 none (the inline-assembly baseline),
        // 14 sc0, 15 sc1, 16 nt, 17 sc0 sc1, 18 sc0 sc1 nt -- does any of them make the L1 ask the L2 for less than a whole 128-byte line?
        const float* q = cells + (size_t)cell * 12;
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f t[6];
#define VRG_DBG_LD(BITS)                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 6; ++i)                                                                \
            asm volatile("global_load_dwordx4 %0, %1, off " BITS : "=&v"(t[i]) : "v"(q + 4 * i) : "memory");
        if (MODE == 13) { VRG_DBG_LD("") }
        else if (MODE == 14) { VRG_DBG_LD("sc0") }
        else if (MODE == 15) { VRG_DBG_LD("sc1") }
        else if (MODE == 16) { VRG_DBG_LD("nt") }
        else if (MODE == 17) { VRG_DBG_LD("sc0 sc1") }
        else { VRG_DBG_LD("sc0 sc1 nt") }
#undef VRG_DBG_LD
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]) :: "memory");
#pragma unroll
        for (int i = 0; i < 6; ++i) acc += (t[i].x + t[i].y) + (t[i].z + t[i].w);
    } else if (MODE == 9 || MODE == 10) {
        const f32x4* q = reinterpret_cast<const f32x4*>(cells + (size_t)cell * 12);
        const f32x4 t = q[0];
        acc += (t.x + t.y) + (t.z + t.w);
        if (MODE == 10) { const f32x4 u = q[5]; acc += (u.x + u.y) + (u.z + u.w); }
    } else {
        const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int lane = threadIdx.x & 63, ql = lane & 3;
        char* my = &slots[w][0][0];
        if (MODE == 11) {
            const float* q = cells + (size_t)cell * 12;
#pragma unroll
            for (int i = 0; i < 6; ++i)
                __builtin_amdgcn_global_load_lds((dbg_gptr)(q + 4 * i), (dbg_lptr)(my + i * DBG_ROUND_BYTES), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(my + i * DBG_ROUND_BYTES + lane * 16);
                acc += (t.x + t.y) + (t.z + t.w);
            }
        } else {
#pragma unroll
            for (int rnd = 0; rnd < 6; ++rnd) {
                const int served = dbg_quad_bcast(cell, rnd);
                const int chunk = rnd < 4 ? ql : 4 + (ql & 1);
                __builtin_amdgcn_global_load_lds((dbg_gptr)(cells + (size_t)served * QSTRIDE + chunk * 4), (dbg_lptr)(my + rnd * DBG_ROUND_BYTES), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int q4 = lane & ~3;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(my + ql * DBG_ROUND_BYTES + (q4 + c) * 16);
                acc += (t.x + t.y) + (t.z + t.w);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(my + (4 + (ql >> 1)) * DBG_ROUND_BYTES + (q4 + 2 * (ql & 1) + c) * 16);
                acc += (t.x + t.y) + (t.z + t.w);
            }
        }
    }
    if (p < pixels) out[p] = acc;
}

// Timing probe for the channel-split LUT: CH_LDS channels of the node table live in LDS (one float / float2 per node), the
// other 3 - CH_LDS channels are gathered from a global record table with 4 * (3 - CH_LDS) floats per (b0, g0, r) record.
// Persistent 1024-thread workgroups (one per CU next to the LDS table).  Values are checksums, not pixels.
template <int CH_LDS>
__global__ __launch_bounds__(1024) void k_dbg_lut_split(const px3* __restrict__ in, float* __restrict__ out, int64_t pixels,
                                                          const float* __restrict__ cells, int n) {
    extern __shared__ __attribute__((aligned(16))) float dbg_nodes[];
    const int total = n * n * n * CH_LDS;
    for (int i = threadIdx.x; i < total; i += 1024) dbg_nodes[i] = cells[i];
    __syncthreads();
    const float top = (float)(n - 1);
    const int nc = n - 1;
    constexpr int REC = 4 * (3 - CH_LDS);
    for (int64_t p = (int64_t)blockIdx.x * 1024 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 1024) {
        const px3 v = in[p];
        const LutAxis R = lut_axis(v.r, 0.f, 1.f, 1, top), G = lut_axis(v.g, 0.f, 1.f, 1, top), B = lut_axis(v.b, 0.f, 1.f, 1, top);
        const int base = (B.cell * n + G.cell) * n + R.cell;
        float acc = 0.0f;
        if (CH_LDS == 1) {
            const float* T = dbg_nodes;
            acc += (T[base] + T[base + 1]) + (T[base + n] + T[base + n + 1]);
            acc += (T[base + n * n] + T[base + n * n + 1]) + (T[base + n * n + n] + T[base + n * n + n + 1]);
        } else {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2* T = reinterpret_cast<const f2*>(dbg_nodes);
            const f2 a = T[base], b = T[base + 1], c = T[base + n], d = T[base + n + 1];
            const f2 e = T[base + n * n], f = T[base + n * n + 1], g = T[base + n * n + n], h = T[base + n * n + n + 1];
            acc += (a.x + a.y) + (b.x + b.y) + (c.x + c.y) + (d.x + d.y) + (e.x + e.y) + (f.x + f.y) + (g.x + g.y) + (h.x + h.y);
        }
        const f32x4* q = reinterpret_cast<const f32x4*>(cells + (size_t)((B.cell * nc + G.cell) * n + R.cell) * REC);
#pragma unroll
        for (int i = 0; i < 2 * (3 - CH_LDS); ++i) { const f32x4 t = q[i]; acc += (t.x + t.y) + (t.z + t.w); }
        out[p] = acc;
    }
}

// Timing probe: the whole table through LDS in three channel passes.  A persistent 1024-thread workgroup takes a batch of
// 1024 * PX pixels, keeps their cell indices in registers and, per output channel, (re)fills the LDS with that channel's
// node table (n^3 floats) and reads the 8 corners of each of its pixels: no global gathers at all, LDS refill traffic
// 12 n^3 / (1024 PX) bytes per pixel.
template <int PX>
__global__ __launch_bounds__(1024) void k_dbg_lut_passes(const px3* __restrict__ in, float* __restrict__ out, int64_t pixels,
                                                           const float* __restrict__ cells, int n) {
    extern __shared__ __attribute__((aligned(16))) float dbg_nodes[];
    const int total = n * n * n;
    const float top = (float)(n - 1);
    const int64_t batch = 1024 * PX;
    for (int64_t b0 = (int64_t)blockIdx.x * batch; b0 < pixels; b0 += (int64_t)gridDim.x * batch) {
        int base[PX];
        float acc[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            int64_t p = b0 + j * 1024 + threadIdx.x;
            p = p < pixels ? p : pixels - 1;
            const px3 v = in[p];
            const LutAxis R = lut_axis(v.r, 0.f, 1.f, 1, top), G = lut_axis(v.g, 0.f, 1.f, 1, top), B = lut_axis(v.b, 0.f, 1.f, 1, top);
            base[j] = (B.cell * n + G.cell) * n + R.cell;
            acc[j] = 0.0f;
        }
        for (int pass = 0; pass < 3; ++pass) {
            __syncthreads();
            const f32x4* src = reinterpret_cast<const f32x4*>(cells + (size_t)pass * total);
            f32x4* dst = reinterpret_cast<f32x4*>(dbg_nodes);
            for (int i = threadIdx.x; i < (total + 3) / 4; i += 1024) dst[i] = src[i];
            __syncthreads();
            const float* T = dbg_nodes;
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const int bs = base[j];
                acc[j] += (T[bs] + T[bs + 1]) + (T[bs + n] + T[bs + n + 1]);
                acc[j] += (T[bs + n * n] + T[bs + n * n + 1]) + (T[bs + n * n + n] + T[bs + n * n + n + 1]);
            }
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int64_t p = b0 + j * 1024 + threadIdx.x;
            if (p < pixels) out[p] = acc[j];
        }
    }
}

// Element-wise pieces of the colour-match arithmetic, one fp32 in -> one fp32 out, so that tests can compare each with
// the torch op the reference executes on this GPU (tests/test_gpu_parity.py::test_device_math_pieces_equal_torch):
//  op 0: __ocml_pow_f32(x, y) with y a kernel argument      (torch.pow(x, y))
//  op 1: x * fl(1/y)                                         (x / python_scalar on the GPU)
//  op 2: x / y, IEEE                                         (x / tensor)
//  op 3: pow_pos(x, y)      op 4: cbrt_pow(x)                (the fast policy's powers)
//  op 9 / 10 / 11: dev_pow_t<DEV_POW_ANY / _OVF / _UNIT>(x, y), the scaffolding-free transcription of ocml powf the device
//        policy uses (_OVF in srgb -> linear, _UNIT in linear -> srgb and the Lab cube root)
//  op 5: Lab of an RGB triple / op 6: RGB of a Lab triple, device policy; op 7 / 8: the same with the fast policy
//  op 20: {lab, mean, std} -> {div_sigma_unscaled, (lab - mean) / std, 1.0 where sigma_recip's and the per-pixel condition hold}
__global__ __launch_bounds__(256) void k_dbg_cm_math(const float* __restrict__ in, float* __restrict__ out, int64_t n, int op, float y,
                                                      DevMath dm) {
    VRG_CM_MATH(PT, true, true, dm);
    __shared__ __attribute__((aligned(16))) float zivt[ZIV_TABLE_WORDS];
    ziv_table_fill(zivt, (int)threadIdx.x, 256);
    __syncthreads();
    dm.logt = zivt;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (op == 20) {          // {lab, mean, std}: the unscaled form of (lab - mean) / std, the division itself, and whether the conditions admit the former
        const float lab = in[3 * i], mu = in[3 * i + 1], sd = in[3 * i + 2];
        const float ms[6] = {mu, sd, mu, sd, mu, sd};
        const SigmaRecip R = sigma_recip(ms);
        const float d = lab - mu;
        out[3 * i] = div_sigma_unscaled(d, sd, R.y1[0]);
        out[3 * i + 1] = d / sd;
        out[3 * i + 2] = (R.usable && (__builtin_fabsf(d) + __builtin_fabsf(d)) + __builtin_fabsf(d) < 0x1p40f) ? 1.0f : 0.0f;
    } else if (op >= 16) {          // 16 / 17: ocml's ln x = hi + lo (epln, as transcribed); 18 / 19: dev_pow_ziv's table log
        float a, b;
        float eh, aj;
        if (op <= 17) dev_epln<DEV_POW_UNIT>(in[i], a, b); else ziv_log(in[i], zivt, a, b, eh, aj);
        out[i] = (op & 1) ? b : a;
    } else if (op >= 12) {
        // the Lab transforms' powers as they are called there (dev_pow_ziv with each call site's domain): op 12 sRGB -> linear
        // (y = 2.4), 13 linear -> sRGB (1/2.4), 14 Lab cube root (1/3); 15: 1.0 where the rounding test fails (the lane would run
        // the transcription), else 0.0 (statistics of the fallback rate)
        const float x = in[i];
        if (op == 12) out[i] = dev_pow_ziv<DEV_POW_OVF>(x, y, zivt, 0x3d800000u, 0x40000000u);
        else if (op == 13) out[i] = dev_pow_ziv<DEV_POW_UNIT>(x, y, zivt, 0x3b4d2e1cu, 0x40800000u);
        else if (op == 14) out[i] = dev_pow_ziv<DEV_POW_UNIT>(x, y, zivt, 0x3c1118c2u, 0x40800000u);
        else {
            float r;
            out[i] = ziv_try(x, y, zivt, 0x00800000u, 0x7f7fffffu, r) ? 0.0f : 1.0f;       // any normal positive x: the rounding test alone
        }
    } else if (op >= 9) {
        out[i] = op == 9 ? dev_pow_t<DEV_POW_ANY>(in[i], y) : (op == 10 ? dev_pow_t<DEV_POW_OVF>(in[i], y) : dev_pow_t<DEV_POW_UNIT>(in[i], y));
    } else if (op <= 4) {
        const float x = in[i];
        float r;
        if (op == 0) r = VRG_LIB_POWF(x, y);
        else if (op == 1) r = x * (float)(1.0 / (double)y);      // caller passes constants that are exact in fp32, or checks 1.055 via op 5
        else if (op == 2) r = x / y;
        else if (op == 3) r = pow_pos(x, y, PT);
        else r = cbrt_pow(x);
        out[i] = r;
    } else {
        const float x[3] = {in[3 * i], in[3 * i + 1], in[3 * i + 2]};
        float o[3];
        if (op == 5) rgb_to_lab(x, o, dm);
        else if (op == 6) lab_to_rgb(x, o, dm);
        else if (op == 7) rgb_to_lab(x, o, PT);
        else lab_to_rgb(x, o, PT);
        out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
    }
}

}  // namespace vrg

namespace vrg {

// Streaming-copy ceiling: out[i] = in[i], 16 B per lane.  The practical HBM roofline every streaming kernel of this library is
// priced against (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).
//  mode 0: one float4 per thread, plain loads / stores        mode 1: the same with the non-temporal hint on both sides
//  mode 2: four float4 per thread (one 64-B run per lane would break coalescing: the four are a workgroup-stride apart), non-temporal
//  mode 3: read only (sum into one float per workgroup)          mode 4: write only
template <int MODE>
__global__ __launch_bounds__(256) void k_dbg_copy(const f32x4* __restrict__ in, f32x4* __restrict__ out, int64_t n4) {
    if (MODE == 0 || MODE == 1) {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= n4) return;
        if (MODE == 0) out[i] = in[i];
        else {
            typedef float v4 __attribute__((ext_vector_type(4)));
            const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(in) + i);
            __builtin_nontemporal_store(v, reinterpret_cast<v4*>(out) + i);
        }
    } else if (MODE == 2) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
        v4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j * 256 < n4) v[j] = __builtin_nontemporal_load(reinterpret_cast<const v4*>(in) + base + j * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j * 256 < n4) __builtin_nontemporal_store(v[j], reinterpret_cast<v4*>(out) + base + j * 256);
    } else if (MODE == 3) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j * 256 < n4) {
                const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(in) + base + j * 256);
                acc += (v.x + v.y) + (v.z + v.w);
            }
        if (acc == 12345.678f) reinterpret_cast<float*>(out)[blockIdx.x] = acc;      // keeps the loads alive, writes (almost) never
    } else {
        typedef float v4 __attribute__((ext_vector_type(4)));
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i < n4) __builtin_nontemporal_store(v4{1.0f, 2.0f, 3.0f, 4.0f}, reinterpret_cast<v4*>(out) + i);
    }
}

// Issue-rate probe: every lane runs `iters` passes over 64 independent-enough instructions of one kind (8 chains x 8),
// nothing else in the loop but the counter.  The measured lane-instructions per second are the VALU roofline the fused
// chains are priced against in DESIGN.md (they are issue bound, not HBM bound).
#define VRG_REP8(X) X X X X X X X X
#define VRG_REP4(X) X X X X
template <int MODE>
__global__ __launch_bounds__(256) void k_dbg_valu_rate(float* __restrict__ out, int32_t iters) {
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f,
          a7 = a0 + 7.0f;
    const float b = 0.999f, c = 1e-3f;
    uint32_t u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 * 11, u5 = u0 * 13, u6 = u0 * 17, u7 = u0 * 19;
    uint64_t w0 = u0, w1 = u1, w2 = u2, w3 = u3, w4 = u4, w5 = u5, w6 = u6, w7 = u7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const f2 pb = {b, b}, pc = {c, c};
    const uint32_t k = 0xD2511F53u, k3 = 0x9E3779B9u;
    const float sk = __builtin_amdgcn_readfirstlane((int)iters) > 2 ? 0.9997f : 1.0f;
    const uint64_t mask = __builtin_amdgcn_readfirstlane((int)iters) > 3 ? 0x5555aaaa0f0ff0f0ull : 0x1ull;
    for (int32_t i = 0; i < iters; ++i) {
        if (MODE == 0) {
            VRG_REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (MODE == 1) {
            VRG_REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %10, %1\n v_mad_u64_u32 %2, vcc, %8, %11, %2\n"
                                  "v_mad_u64_u32 %3, vcc, %8, %12, %3\n v_mad_u64_u32 %4, vcc, %8, %13, %4\n v_mad_u64_u32 %5, vcc, %8, %14, %5\n"
                                  "v_mad_u64_u32 %6, vcc, %8, %15, %6\n v_mad_u64_u32 %7, vcc, %8, %16, %7"
                                  : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                                  : "v"(k), "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7) : "vcc");)
        } else if (MODE == 2) {
            VRG_REP8(asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                                  "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 3) {
            VRG_REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                                  "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        } else if (MODE == 4) {
            VRG_REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                                  "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k));)
        } else if (MODE == 5) {
            VRG_REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sin_f32 %1, %1\n v_cos_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                                  "v_sqrt_f32 %4, %4\n v_sin_f32 %5, %5\n v_cos_f32 %6, %6\n v_rcp_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 6) {
            VRG_REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
                                  "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 7) {
            VRG_REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                                  "v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3"
                                  : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(w4));)
        } else if (MODE == 8) {
            VRG_REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 9) {
            VRG_REP8(asm volatile("v_add_f32_dpp %0, %8, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %8, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %2, %8, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %8, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %4, %8, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %5, %8, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %6, %8, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %7, %8, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 10) {
            VRG_REP8(asm volatile("v_add_f32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 11) {
            VRG_REP8(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 12) {
            VRG_REP8(asm volatile("v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 13) {
            VRG_REP8(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 14) {
            VRG_REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 15) {
            VRG_REP8(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k) : "vcc");)
        } else if (MODE == 16) {
            VRG_REP8(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k) : "vcc");)
        } else if (MODE == 17) {
            VRG_REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k) : "vcc");)
        } else if (MODE == 18) {
            VRG_REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_nop 1\n"
                                  "v_add_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                                  "v_add_f32_dpp %2, %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
        } else if (MODE == 19) {
            VRG_REP8(asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n v_sin_f32 %4, %4\n v_sin_f32 %5, %5\n v_sin_f32 %6, %6\n v_sin_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 20) {
            VRG_REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                                  "v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9"
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        } else if (MODE == 21) {
            VRG_REP4(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %9\n v_mul_f32 %1, %1, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %2, %2, %9\n v_mul_f32 %3, %3, %8\n v_add_f32 %3, %3, %9\n v_mul_f32 %4, %4, %8\n v_add_f32 %4, %4, %9\n v_mul_f32 %5, %5, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %6, %6, %9\n v_mul_f32 %7, %7, %8\n v_add_f32 %7, %7, %9"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 30) {
            VRG_REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 31) {
            VRG_REP8(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 32) {
            VRG_REP8(asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 33) {
            VRG_REP8(asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 34) {
            VRG_REP8(asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %9\n v_fmamk_f32 %1, %1, 0x3f7fbe77, %9\n v_fmamk_f32 %2, %2, 0x3f7fbe77, %9\n v_fmamk_f32 %3, %3, 0x3f7fbe77, %9\n v_fmamk_f32 %4, %4, 0x3f7fbe77, %9\n v_fmamk_f32 %5, %5, 0x3f7fbe77, %9\n v_fmamk_f32 %6, %6, 0x3f7fbe77, %9\n v_fmamk_f32 %7, %7, 0x3f7fbe77, %9"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 35) {
            VRG_REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 36) {
            VRG_REP8(asm volatile("v_add_f32_e64 %0, %0, %9 clamp\n v_add_f32_e64 %1, %1, %9 clamp\n v_add_f32_e64 %2, %2, %9 clamp\n v_add_f32_e64 %3, %3, %9 clamp\n v_add_f32_e64 %4, %4, %9 clamp\n v_add_f32_e64 %5, %5, %9 clamp\n v_add_f32_e64 %6, %6, %9 clamp\n v_add_f32_e64 %7, %7, %9 clamp"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 37) {
            VRG_REP8(asm volatile("v_cmp_u_f32 vcc, %0, %8\n v_cmp_u_f32 vcc, %1, %8\n v_cmp_u_f32 vcc, %2, %8\n v_cmp_u_f32 vcc, %3, %8\n v_cmp_u_f32 vcc, %4, %8\n v_cmp_u_f32 vcc, %5, %8\n v_cmp_u_f32 vcc, %6, %8\n v_cmp_u_f32 vcc, %7, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 38) {
            VRG_REP8(asm volatile("v_cmp_class_f32 vcc, %0, %8\n v_cmp_class_f32 vcc, %1, %8\n v_cmp_class_f32 vcc, %2, %8\n v_cmp_class_f32 vcc, %3, %8\n v_cmp_class_f32 vcc, %4, %8\n v_cmp_class_f32 vcc, %5, %8\n v_cmp_class_f32 vcc, %6, %8\n v_cmp_class_f32 vcc, %7, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 40) {
            VRG_REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %9, %10\n v_cndmask_b32_e64 %1, %1, %9, %10\n v_cndmask_b32_e64 %2, %2, %9, %10\n v_cndmask_b32_e64 %3, %3, %9, %10\n v_cndmask_b32_e64 %4, %4, %9, %10\n v_cndmask_b32_e64 %5, %5, %9, %10\n v_cndmask_b32_e64 %6, %6, %9, %10\n v_cndmask_b32_e64 %7, %7, %9, %10"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 41) {
            VRG_REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 42) {
            VRG_REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 43) {
            VRG_REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 44) {
            VRG_REP8(asm volatile("v_cos_f32 %0, %0\n v_cos_f32 %1, %1\n v_cos_f32 %2, %2\n v_cos_f32 %3, %3\n v_cos_f32 %4, %4\n v_cos_f32 %5, %5\n v_cos_f32 %6, %6\n v_cos_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 45) {
            VRG_REP8(asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 46) {
            VRG_REP8(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 47) {
            VRG_REP8(asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 48) {
            VRG_REP8(asm volatile("v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %1, %1, %8\n v_ldexp_f32 %2, %2, %8\n v_ldexp_f32 %3, %3, %8\n v_ldexp_f32 %4, %4, %8\n v_ldexp_f32 %5, %5, %8\n v_ldexp_f32 %6, %6, %8\n v_ldexp_f32 %7, %7, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 49) {
            VRG_REP8(asm volatile("v_frexp_mant_f32 %0, %0\n v_frexp_mant_f32 %1, %1\n v_frexp_mant_f32 %2, %2\n v_frexp_mant_f32 %3, %3\n v_frexp_mant_f32 %4, %4\n v_frexp_mant_f32 %5, %5\n v_frexp_mant_f32 %6, %6\n v_frexp_mant_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 50) {
            VRG_REP8(asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 51) {
            VRG_REP8(asm volatile("v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3\n v_cvt_u32_f32 %4, %4\n v_cvt_u32_f32 %5, %5\n v_cvt_u32_f32 %6, %6\n v_cvt_u32_f32 %7, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 52) {
            VRG_REP8(asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 53) {
            VRG_REP8(asm volatile("v_add_f32 %0, %0, %8\n s_nop 1\n v_add_f32 %1, %1, %8\n s_nop 1\n v_add_f32 %2, %2, %8\n s_nop 1\n v_add_f32 %3, %3, %8\n s_nop 1\n v_add_f32 %4, %4, %8\n s_nop 1\n v_add_f32 %5, %5, %8\n s_nop 1\n v_add_f32 %6, %6, %8\n s_nop 1\n v_add_f32 %7, %7, %8\n s_nop 1"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "vcc");)
        } else if (MODE == 60) {
            VRG_REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k3), "s"(mask) : "vcc");)
        } else if (MODE == 61) {
            VRG_REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k3), "s"(mask) : "vcc");)
        } else if (MODE == 62) {
            VRG_REP8(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k3), "s"(mask) : "vcc");)
        } else if (MODE == 63) {
            VRG_REP8(asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k3), "s"(mask) : "vcc");)
        } else if (MODE == 64) {
            VRG_REP8(asm volatile("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k3), "s"(mask) : "vcc");)
        } else if (MODE == 65) {
            VRG_REP8(asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k3), "s"(mask) : "vcc");)
        } else if (MODE == 66) {
            VRG_REP8(asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k3), "s"(mask) : "vcc");)
        } else if (MODE == 67) {
            VRG_REP8(asm volatile("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8\n v_bfe_u32 %4, %4, 3, 8\n v_bfe_u32 %5, %5, 3, 8\n v_bfe_u32 %6, %6, 3, 8\n v_bfe_u32 %7, %7, 3, 8"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k3), "s"(mask) : "vcc");)
        } else if (MODE == 68) {
            VRG_REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k3), "s"(mask) : "vcc");)
        } else if (MODE == 69) {
            VRG_REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc"
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(k), "v"(k3), "s"(mask) : "vcc");)
        } else if (MODE == 72) {
            VRG_REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        } else if (MODE == 73) {
            VRG_REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        } else if (MODE == 55) {
            VRG_REP8(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                                  "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sk));)
        } else if (MODE == 56) {
            VRG_REP8(asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0\n v_mul_f32 %1, 0x3f7fbe77, %1\n v_mul_f32 %2, 0x3f7fbe77, %2\n v_mul_f32 %3, 0x3f7fbe77, %3\n"
                                  "v_mul_f32 %4, 0x3f7fbe77, %4\n v_mul_f32 %5, 0x3f7fbe77, %5\n v_mul_f32 %6, 0x3f7fbe77, %6\n v_mul_f32 %7, 0x3f7fbe77, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 57) {
            VRG_REP8(asm volatile("v_mul_f32 %0, 0.5, %0\n v_mul_f32 %1, 0.5, %1\n v_mul_f32 %2, 0.5, %2\n v_mul_f32 %3, 0.5, %3\n"
                                  "v_mul_f32 %4, 0.5, %4\n v_mul_f32 %5, 0.5, %5\n v_mul_f32 %6, 0.5, %6\n v_mul_f32 %7, 0.5, %7"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 39) {
            VRG_REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n"
                                  "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 54) {
            VRG_REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n"
                                  "v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 5\n v_readlane_b32 s22, %6, 7\n v_readlane_b32 s23, %7, 9"
                                  : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21", "s22", "s23");)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) +
                                          (float)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7) + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

// lane_prev / lane_next self-test (csrc/vrg_lanes.hpp): out[lane] = lane_prev(lane), out[64 + lane] = lane_next(lane)
__global__ void k_selftest_lanes(float* out) {
    const float v = (float)threadIdx.x;
    out[threadIdx.x] = lane_prev(v);
    out[64 + threadIdx.x] = lane_next(v);
}

// Exhaustive check of the Welford update's division (vrg_tstats_body.hpp: q = delta * rn; e = fma(-n, q, delta); q' = fma(e, rn, q) with
// rn = 1.0f / n) against the IEEE quotient: one thread per (count n, fp32 significand of delta).  The sequence commutes with scaling
// delta by a power of two as long as nothing leaves the normal range -- which the kernels' range flag (2^-100 <= |delta| <= 2^100)
// guarantees -- and with its sign, so 2^23 significands per count are ALL inputs: equality is established by enumeration for every
// count the kernels can reach (the launcher falls back to the IEEE division beyond TS_MARKSTEIN_MAX_COUNT).
__global__ void __launch_bounds__(256) k_selftest_welford_division(unsigned long long* mismatches, uint32_t n_first) {
    const uint32_t n = n_first + blockIdx.y;
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;            // 0 .. 2^23 - 1
    const float delta = f32_from_bits(0x3f800000u | m);
    const float nf = (float)n;
    const float rn = 1.0f / nf;
    const float q0 = delta * rn;
    const float e = __builtin_fmaf(-nf, q0, delta);
    const float q = __builtin_fmaf(e, rn, q0);
    if (!(q == delta / nf)) atomicAdd(mismatches, 1ull);
}
}  // namespace vrg

extern "C" int vrg_selftest_welford_division(unsigned long long* mismatches1, uint32_t n_first, uint32_t n_count, void* stream) {
    if (!mismatches1 || n_first == 0 || n_count == 0 || (uint64_t)n_first + n_count > (1ull << 24)) return VRG_ERR_BAD_ARG;
    for (uint32_t done = 0; done < n_count; done += 32768u) {
        const uint32_t now = n_count - done < 32768u ? n_count - done : 32768u;
        hipLaunchKernelGGL(vrg::k_selftest_welford_division, dim3(1u << 15, now), dim3(256), 0, (hipStream_t)stream, mismatches1, n_first + done);
        if (hipGetLastError() != hipSuccess) return VRG_ERR_LAUNCH;
    }
    return VRG_OK;
}

// Host-only: the geometry ts_config derives for (num_outputs, reduction length, vector width): cfg4 = {block_width, block_height,
// split across warps, vectorised}; VRG_ERR_UNSUPPORTED where torch would split the reduction across workgroups.  No GPU needed
// (tests/test_torch_reduce_oracle.py compares it with what rocprofv3 recorded for torch's own launches).
extern "C" int vrg_debug_torch_reduce_config(int64_t num_outputs, int64_t reduce_len, int32_t vec, int32_t* cfg4) {
    if (!cfg4 || num_outputs < 1 || reduce_len < 1 || (vec != 2 && vec != 4)) return VRG_ERR_BAD_ARG;
    vrg::TsCfg c;
    if (!vrg::ts_config(num_outputs, reduce_len, vec, c)) return VRG_ERR_UNSUPPORTED;
    cfg4[0] = c.bw; cfg4[1] = c.bh; cfg4[2] = c.split; cfg4[3] = c.vectorize;
    return VRG_OK;
}

extern "C" {

int vrg_selftest_lanes(float* out128, void* stream) {
    if (!out128) return VRG_ERR_BAD_ARG;
    hipLaunchKernelGGL(vrg::k_selftest_lanes, dim3(1), dim3(64), 0, (hipStream_t)stream, out128);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}


int vrg_debug_valu_rate(float* out, int32_t blocks, int32_t iters, int32_t mode, void* stream) {
    if (!out || blocks <= 0 || iters <= 0 || mode < 0) return VRG_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
#define VRG_VALU_CASE(M) case M: hipLaunchKernelGGL(vrg::k_dbg_valu_rate<M>, dim3(blocks), dim3(256), 0, st, out, iters); break;
    switch (mode) {
        VRG_VALU_CASE(0) VRG_VALU_CASE(1) VRG_VALU_CASE(2) VRG_VALU_CASE(3) VRG_VALU_CASE(4) VRG_VALU_CASE(5) VRG_VALU_CASE(6) VRG_VALU_CASE(7) VRG_VALU_CASE(8) VRG_VALU_CASE(9) VRG_VALU_CASE(10) VRG_VALU_CASE(11) VRG_VALU_CASE(12) VRG_VALU_CASE(13) VRG_VALU_CASE(14) VRG_VALU_CASE(15) VRG_VALU_CASE(16) VRG_VALU_CASE(17) VRG_VALU_CASE(18) VRG_VALU_CASE(19) VRG_VALU_CASE(20) VRG_VALU_CASE(21) VRG_VALU_CASE(30) VRG_VALU_CASE(31) VRG_VALU_CASE(32) VRG_VALU_CASE(33) VRG_VALU_CASE(34) VRG_VALU_CASE(35) VRG_VALU_CASE(36) VRG_VALU_CASE(37) VRG_VALU_CASE(38) VRG_VALU_CASE(39) VRG_VALU_CASE(40) VRG_VALU_CASE(41) VRG_VALU_CASE(42) VRG_VALU_CASE(43) VRG_VALU_CASE(44) VRG_VALU_CASE(45) VRG_VALU_CASE(46) VRG_VALU_CASE(47) VRG_VALU_CASE(48) VRG_VALU_CASE(49) VRG_VALU_CASE(50) VRG_VALU_CASE(51) VRG_VALU_CASE(52) VRG_VALU_CASE(53) VRG_VALU_CASE(54) VRG_VALU_CASE(55) VRG_VALU_CASE(56) VRG_VALU_CASE(57) VRG_VALU_CASE(60) VRG_VALU_CASE(61) VRG_VALU_CASE(62) VRG_VALU_CASE(63) VRG_VALU_CASE(64) VRG_VALU_CASE(65) VRG_VALU_CASE(66) VRG_VALU_CASE(67) VRG_VALU_CASE(68) VRG_VALU_CASE(69) VRG_VALU_CASE(72) VRG_VALU_CASE(73)
        default: return VRG_ERR_BAD_ARG;
    }
#undef VRG_VALU_CASE
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

int vrg_debug_copy_f32(const float* in, float* out, int64_t n_floats, int32_t mode, void* stream) {
    if (!in || !out || n_floats <= 0 || (n_floats & 3) || mode < 0 || mode > 4) return VRG_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) return VRG_ERR_BAD_ARG;
    const int64_t n4 = n_floats / 4;
    const int64_t per_block = (mode == 2 || mode == 3) ? 1024 : 256;
    const uint64_t blocks = (uint64_t)((n4 + per_block - 1) / per_block);
    if (blocks > 0x7fffffffull) return VRG_ERR_UNSUPPORTED;
    const vrg::f32x4* src = reinterpret_cast<const vrg::f32x4*>(in);
    vrg::f32x4* dst = reinterpret_cast<vrg::f32x4*>(out);
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case 0: hipLaunchKernelGGL(vrg::k_dbg_copy<0>, dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, n4); break;
        case 1: hipLaunchKernelGGL(vrg::k_dbg_copy<1>, dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, n4); break;
        case 2: hipLaunchKernelGGL(vrg::k_dbg_copy<2>, dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, n4); break;
        case 3: hipLaunchKernelGGL(vrg::k_dbg_copy<3>, dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, n4); break;
        default: hipLaunchKernelGGL(vrg::k_dbg_copy<4>, dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, n4); break;
    }
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

int vrg_debug_cm_math(const float* in, float* out, int64_t n, int32_t op, float y, void* stream) {
    if (!in || !out || n <= 0 || op < 0 || op > 20) return VRG_ERR_BAD_ARG;
    const uint64_t blocks = (uint64_t)(n + 255) / 256;
    if (blocks > 0x7fffffffull) return VRG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(vrg::k_dbg_cm_math, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, in, out, n, op, y, vrg::host_dev_math());
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

int vrg_debug_lut_fetch(const float* in, float* out, int64_t pixels, const float* cells, int32_t lut_size, int32_t mode, void* stream) {
    if (!in || !out || !cells || pixels <= 0 || lut_size < 2 || mode < 0 || mode > 19) return VRG_ERR_BAD_ARG;
    const uint32_t blocks = (uint32_t)((pixels + 255) / 256);
    const vrg::px3* src = reinterpret_cast<const vrg::px3*>(in);
    if (mode >= 9) {
        switch (mode) {
            case 9: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<9>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
            case 10: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<10>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
            case 11: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<11>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
            case 12: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<12>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
            case 13: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<13>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
            case 14: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<14>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
            case 15: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<15>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
            case 16: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
            case 17: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<17>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
            case 18: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<18>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
            default: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch_r4<19>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
        }
        VRG_CHECK_LAUNCH();
        return VRG_OK;
    }
    if (mode >= 7) {      // three channel passes through LDS, 8 (mode 7) or 16 (mode 8) pixels per thread
        const size_t lds = ((size_t)lut_size * lut_size * lut_size * 4 + 15) / 16 * 16;
        if (lds > 160 * 1024) return VRG_ERR_UNSUPPORTED;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        const void* fn = mode == 7 ? reinterpret_cast<const void*>(vrg::k_dbg_lut_passes<8>) : reinterpret_cast<const void*>(vrg::k_dbg_lut_passes<16>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VRG_ERR_LAUNCH;
        if (mode == 7) hipLaunchKernelGGL(vrg::k_dbg_lut_passes<8>, dim3(cus), dim3(1024), lds, (hipStream_t)stream, src, out, pixels, cells, lut_size);
        else hipLaunchKernelGGL(vrg::k_dbg_lut_passes<16>, dim3(cus), dim3(1024), lds, (hipStream_t)stream, src, out, pixels, cells, lut_size);
        VRG_CHECK_LAUNCH();
        return VRG_OK;
    }
    if (mode >= 5) {      // channel-split probes: mode 5 = one channel in LDS, 6 = two
        const int ch = mode - 4;
        const size_t lds = (size_t)lut_size * lut_size * lut_size * 4 * ch;
        if (lds > 160 * 1024) return VRG_ERR_UNSUPPORTED;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        const void* fn = ch == 1 ? reinterpret_cast<const void*>(vrg::k_dbg_lut_split<1>) : reinterpret_cast<const void*>(vrg::k_dbg_lut_split<2>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VRG_ERR_LAUNCH;
        if (ch == 1) hipLaunchKernelGGL(vrg::k_dbg_lut_split<1>, dim3(cus), dim3(1024), lds, (hipStream_t)stream, src, out, pixels, cells, lut_size);
        else hipLaunchKernelGGL(vrg::k_dbg_lut_split<2>, dim3(cus), dim3(1024), lds, (hipStream_t)stream, src, out, pixels, cells, lut_size);
        VRG_CHECK_LAUNCH();
        return VRG_OK;
    }
    switch (mode) {
        case 0: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
        case 1: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
        case 2: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
        case 3: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
        default: hipLaunchKernelGGL(vrg::k_dbg_lut_fetch<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, out, pixels, cells, lut_size); break;
    }
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

int vrg_selftest_bm_radius(unsigned long long* counts1, void* stream) {
    if (!counts1) return VRG_ERR_BAD_ARG;
    if (hipMemsetAsync(counts1, 0, sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess) return VRG_ERR_LAUNCH;
    for (uint32_t part = 0; part < 4; ++part) {
        hipLaunchKernelGGL(vrg::k_selftest_bm_radius, dim3(1u << 22), dim3(256), 0, (hipStream_t)stream, counts1, part << 30);
        VRG_CHECK_LAUNCH();
    }
    return VRG_OK;
}

int vrg_selftest_divconst(unsigned long long* counts18, void* stream) {
    if (!counts18) return VRG_ERR_BAD_ARG;
    if (hipMemsetAsync(counts18, 0, 18 * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess) return VRG_ERR_LAUNCH;
    // the dispatch packet counts work-items in 32 bits: sweep the 2^32 patterns in four launches
    for (uint32_t part = 0; part < 4; ++part) {
        hipLaunchKernelGGL(vrg::k_selftest_divconst, dim3(1u << 22), dim3(256), 0, (hipStream_t)stream, counts18, part << 30);
        VRG_CHECK_LAUNCH();
    }
    return VRG_OK;
}

}  // extern "C"
