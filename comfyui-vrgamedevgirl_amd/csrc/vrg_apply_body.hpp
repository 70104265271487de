// vrg_apply_body.hpp -- the workgroup body of the apply march (see vrg_apply_march.hip for the design), as a device function shared by
// k_apply_march (round 3's fused stage kernel, tools/experiments/vrg_stage.hip.txt, ran it as a workgroup role).
#pragma once
#include "vrg_chain_stages.hpp"
#include "vrg_lanes.hpp"

namespace vrg {

#define VRG_APPLY_ROWS 60     /* rows per strip segment, a multiple of 3 (the row registers rotate by name) */
constexpr int APPLY_ROWS = VRG_APPLY_ROWS;
constexpr int APPLY_COLS = 62;                    // output columns per wave
static_assert(APPLY_ROWS % 3 == 0, "APPLY_ROWS must be a multiple of 3");

// value held by lane-1 / lane+1 (lane 0 / lane 63 read 0.0: halo lanes, their results are never stored).  The bound_ctrl form without an
// `old` operand (vrg_lanes.hpp): the update_dpp(0, ...) form cost one v_mov_b32 v, 0 per shift on top of the v_mov_b32_dpp
__device__ __forceinline__ float am_prev(float v) { return tap_prev(v); }
__device__ __forceinline__ float am_next(float v) { return tap_next(v); }

struct AmRow { float l[3], c[3], r[3]; };               // one processed row: left tap, own value, right tap per channel

// The workgroup body of k_apply_march, callable from other kernels: `vblock` / `vgrid` = the
// workgroup's index in / the size of the apply grid, `PT` = the colour-match arithmetic object (tables staged in LDS by the caller).
// GENERAL = false (frames of at least APPLY_COLS x APPLY_ROWS pixels and less than 2 GiB): the loop has NO conditional block -- the
// two halo lanes of a wave "store" through a buffer descriptor at an offset past the frame, which the hardware drops -- so the
// compiler's memory counter stays exact (a store inside an exec-masked block merges to `s_waitcnt vmcnt(0)` at the join), and the
// next row is requested BEFORE the current row's ~450 instructions of colour transfer instead of after them: the request lands under
// that arithmetic where it used to be waited for ~100 instructions after its issue (one exposed L2 / HBM latency per row and wave).
#define VRG_APPLY_EARLY_LOAD 1
template <int STAGES, class MATH, bool GENERAL = true>
__device__ __forceinline__ void apply_march_body(uint32_t vblock, uint32_t vgrid, const px3* __restrict__ in, px3* __restrict__ out, int32_t H, int32_t W,
                                                 int32_t strips_x, int32_t segs_y, uint32_t total_waves, const ChainK& D, const MATH& PT) {
    // XCD-aware placement as in k_chain_tile: workgroup b runs on XCD b % 8; give every XCD one contiguous run of work
    // (a launch with fewer workgroups than wave groups walks them with a stride of vgrid / 8 per XCD: the persistent form, used
    // by an experiment that ran pass 2 beside pass 1 -- slower, LABNOTES.md -- and kept because it costs nothing)
    const uint32_t groups = (total_waves + 3u) / 4u;
    const uint32_t per_xcd = (groups + 7u) / 8u;
    const int lane = threadIdx.x & 63;
    const bool zero = D.zero_border != 0;
    const int64_t ppf = (int64_t)H * W;
    const float n0[3] = {0.0f, 0.0f, 0.0f};
    for (uint32_t slot = vblock >> 3; slot < per_xcd; slot += (vgrid >> 3)) {
    const uint32_t grp = (vblock & 7u) * per_xcd + slot;
    if (grp >= groups) break;
    // the wave's index is the same in all its lanes: say so (readfirstlane), and the strip / segment / frame arithmetic, the row
    // loop's bounds and the row addresses live in SGPRs -- a row loop the compiler takes for divergent runs under an exec mask and
    // merges its memory counters to `vmcnt(0)` at the loop header
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(grp * 4u + (threadIdx.x >> 6)));
    if (wv >= total_waves) break;
    const uint32_t strip = wv % (uint32_t)strips_x;
    const uint32_t rest = wv / (uint32_t)strips_x;
    const uint32_t seg = rest % (uint32_t)segs_y;
    const int64_t f = rest / (uint32_t)segs_y;
    int32_t x0 = (int32_t)strip * APPLY_COLS, y0 = (int32_t)seg * APPLY_ROWS;
    if (W >= APPLY_COLS) x0 = x0 < W - APPLY_COLS ? x0 : W - APPLY_COLS;          // last strip: overlap instead of a ragged end
    if (H >= APPLY_ROWS) y0 = y0 < H - APPLY_ROWS ? y0 : H - APPLY_ROWS;          // last segment: likewise
    const int32_t rows = H >= APPLY_ROWS ? APPLY_ROWS : H;
    const int32_t x = x0 - 1 + lane;                                             // the column this lane evaluates
    const bool x_in = x >= 0 && x < W;
    const int32_t xc = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    const bool stores = lane >= 1 && lane <= APPLY_COLS && x_in;                 // (x_in: frames narrower than a strip)
    const px3* fin = in + f * ppf;
    px3* fout = out + f * ppf;
    __amdgpu_buffer_rsrc_t frame_rsrc, in_rsrc;
    if (!GENERAL) {                                                              // wave-uniform frame bases for the descriptors (SGPRs)
        const uint64_t fb = reinterpret_cast<uint64_t>(fout);
        const uint64_t fbu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(fb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)fb);
        frame_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(fbu), 0, (int)(ppf * 12), 0x00020000);
        const uint64_t ib = reinterpret_cast<uint64_t>(fin);
        const uint64_t ibu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ib >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ib);
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(ibu), 0, (int)(ppf * 12), 0x00020000);
    }
    const FrameCtx FC = frame_ctx<STAGES>(D, f);
    const uint32_t x_bytes = (uint32_t)xc * 12u;                                 // this lane's column inside a row: the loop-invariant vector offset
    const uint32_t store_bytes = stores ? (uint32_t)x * 12u : 0x80000000u;       // the same for the store; halo lanes point past the frame

    auto load = [&](int32_t y) {                                                  // raw pixel of row y (clamped), this lane's column
        const int32_t yc = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        if (!GENERAL) {
            // the row's byte offset is wave-uniform (the descriptor's scalar offset), the column's is loop-invariant: no vector arithmetic
            // per row, where the 64-bit address of a global load cost two v_mad_u64_u32 and two moves
            typedef unsigned u3 __attribute__((ext_vector_type(3)));
            const u3 v = __builtin_amdgcn_raw_buffer_load_b96(in_rsrc, (int)x_bytes, (int)((uint32_t)(yc * W) * 12u), 0);
            return px3{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z)};
        }
        return fin[(int64_t)yc * W + xc];
    };
    auto process = [&](const px3& v, int32_t y) {                                 // pre stages + the row's left / right taps
        AmRow r;
        const float xi[3] = {v.r, v.g, v.b};
        float o[3];
        chain_apply_stages<STAGES>(D, FC, xi, n0, o, PT);
        r.c[0] = o[0]; r.c[1] = o[1]; r.c[2] = o[2];                              // replicate border = the clamped coordinate's own value
        if (zero) {
            // zero border (the reference's use_gpu flag): pixels outside the frame count as 0.  A wave-uniform branch that the optimiser must
            // keep (the empty volatile asm cannot be speculated): as three selects in the straight line it cost every row of the default,
            // replicated, border three v_cndmask_b32
            asm volatile("" ::: "memory");
            const bool inside = x_in && y >= 0 && y < H;
#pragma unroll
            for (int c = 0; c < 3; ++c) r.c[c] = inside ? r.c[c] : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r.l[c] = am_prev(r.c[c]);
            r.r[c] = am_next(r.c[c]);
        }
        return r;
    };
    auto emit = [&](int32_t y, const AmRow& a, const AmRow& b, const AmRow& c) {
        float res[3];
        if (STAGES & VRG_STAGE_COLORMATCH) {       // the taps left lab_to_rgb's clip (or are border zeros): [0, 1] or NaN
            const float p[3][3][3] = {{{a.l[0], a.c[0], a.r[0]}, {b.l[0], b.c[0], b.r[0]}, {c.l[0], c.c[0], c.r[0]}},
                                      {{a.l[1], a.c[1], a.r[1]}, {b.l[1], b.c[1], b.r[1]}, {c.l[1], c.c[1], c.r[1]}},
                                      {{a.l[2], a.c[2], a.r[2]}, {b.l[2], b.c[2], b.r[2]}, {c.l[2], c.c[2], c.r[2]}}};
            stencil_value3_unit(D.stencil_op, p, D.strength, D.zero_border, res);
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float p[3][3] = {{a.l[ch], a.c[ch], a.r[ch]}, {b.l[ch], b.c[ch], b.r[ch]}, {c.l[ch], c.c[ch], c.r[ch]}};
                res[ch] = stencil_value(D.stencil_op, p, D.strength, D.zero_border);
            }
        }
        if (GENERAL) {
            if (stores && y < y0 + rows) fout[(int64_t)y * W + x] = px3{res[0], res[1], res[2]};
        } else {
            typedef unsigned u3 __attribute__((ext_vector_type(3)));
            // halo lanes: a vector offset out of range, dropped by the hardware; the row's offset is the scalar one (y < H: the sum stays below 2^31 + 2^31)
            __builtin_amdgcn_raw_buffer_store_b96(u3{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2])}, frame_rsrc, (int)store_bytes,
                                                  (int)((uint32_t)(y * W) * 12u), 0);
        }
    };

    AmRow r0 = process(load(y0 - 1), y0 - 1);
    AmRow r1 = process(load(y0), y0);
    AmRow r2;
    const int32_t y1 = y0 + rows;
    px3 q = load(y0 + 1);
    if (!GENERAL && VRG_APPLY_EARLY_LOAD) {
        for (int32_t y = y0; y < y1; y += 3) {                                    // three steps per trip: r0, r1, r2 rotate by name
            const px3 a = load(y + 2);                                            // lands under process(q)
            r2 = process(q, y + 1);
            emit(y, r0, r1, r2);
            const px3 b = load(y + 3);
            r0 = process(a, y + 2);
            emit(y + 1, r1, r2, r0);
            q = load(y + 4);
            r1 = process(b, y + 3);
            emit(y + 2, r2, r0, r1);
        }
    } else {
        for (int32_t y = y0; y < y1; y += 3) {
            r2 = process(q, y + 1);
            q = load(y + 2);
            emit(y, r0, r1, r2);
            r0 = process(q, y + 2);
            q = load(y + 3);
            emit(y + 1, r1, r2, r0);
            r1 = process(q, y + 3);
            q = load(y + 4);
            emit(y + 2, r2, r0, r1);
        }
    }
    }       // persistent walk
}


}  // namespace vrg
