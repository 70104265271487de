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

// ----------------------------------------------------------------------------------------------
// Stand-alone stencil, streaming form ("flat march").  A 3x3 stencil on interleaved frames is the same formula for every
// FLOAT of a row, with horizontal taps C floats away -- pixels only matter at the two row ends.  So a frame row is walked as
// W*C/4 float4 vectors: one wave64 owns a strip of 64 consecutive vectors (1 KB per row, loaded and stored as global_*_dwordx4,
// fully coalesced) and marches down VRG_FLAT_ROWS rows keeping three rows in registers.  A vector's left / right taps are the last
// C floats of the previous lane's vector and the first C of the next lane's: 2 C DPP wave shifts per row; lanes 0 and 63 take
// them from one extra 16-byte load each (one wave-level load with two active lanes).  No LDS, no barrier, nothing recomputed; the
// two priming rows of a strip segment are re-read from L2.  Same arithmetic as every other stencil kernel of the library
// (stencil_value), so results are bit-identical.  Needs W*C % 4 == 0 and 16-byte aligned frames; instantiated for C = 3 and 4.
// ----------------------------------------------------------------------------------------------
#define VRG_FLAT_ROWS 36      /* rows per strip segment: a multiple of 3 (the row registers rotate by name three steps per trip) */
constexpr int FLAT_ROWS = VRG_FLAT_ROWS;
static_assert(FLAT_ROWS % 3 == 0, "FLAT_ROWS must be a multiple of 3");

typedef float fv4 __attribute__((ext_vector_type(4)));

template <int C>
struct FlatRow { float o[4], p[C], n[C]; };            // own vector, the C floats left of it, the C floats right of it
struct FlatRaw { fv4 own, halo; };                      // as loaded: halo = neighbour vector for lane 0 (left) / lane 63 (right)

__device__ __forceinline__ float flat_shr(float old, float v) {   // value of lane-1; lane 0 keeps `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float flat_shl(float old, float v) {   // value of lane+1; lane 63 keeps `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// One strip segment: rows [y0, y0 + rows) of the 64 vectors starting at column col - lane.
// GENERAL = false (every launch on frames of at least 64 vectors x FLAT_ROWS rows): all 64 lanes hold a vector and the segment has
// exactly FLAT_ROWS rows -- the last strip of a row and the last segment of a frame START EARLIER instead of ending short (they
// overlap their neighbours and store the same values twice) -- so the loop has no conditional store or load: a store inside a
// conditional block, or a load whose use is behind a branch, makes the compiler wait for vmcnt(0) / sink the load next to its use,
// and the next row's loads would no longer fly behind this row's arithmetic.
// GENERAL = true: frames smaller than that (any size): ragged strip, ragged segment, same arithmetic.
template <int C, int OP, bool ZERO, bool GENERAL>
__device__ __forceinline__ void flat_march(const fv4* __restrict__ fin, fv4* __restrict__ fout, int32_t H, int32_t n4, int32_t col, int lane,
                                           bool edge_strip, int32_t y0, int32_t rows, float strength) {
    const bool valid = !GENERAL || col < n4;
    const int32_t colc = valid ? col : n4 - 1;
    const bool first = col == 0, last = col == n4 - 1;
    const bool halo_lane = valid && ((lane == 0 && col > 0) || (lane == 63 && col + 1 < n4));
    const int32_t hcol = halo_lane ? (lane == 0 ? col - 1 : col + 1) : colc;
    const int32_t y1 = y0 + rows;

    auto load = [&](int32_t y) {                                     // row y of the frame with the border rule applied in y
        FlatRaw q;
        q.own = fv4{0.0f, 0.0f, 0.0f, 0.0f};
        q.halo = q.own;
        const bool inside = y >= 0 && y < H;
        if (ZERO && !inside) return q;                               // uniform
        const int32_t yc = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        const fv4* row = fin + (int64_t)yc * n4;
        // branch-free, so that both requests stay in flight behind the previous row's arithmetic: lanes past the row end re-read its
        // last vector, and every lane but 0 and 63 re-reads its own vector as "halo" (an L1 hit; the value is unused).  Plain loads:
        // the priming rows are re-read by the neighbouring segments.
        q.own = row[colc];
        q.halo = row[hcol];
        return q;
    };
    auto process = [&](const FlatRaw& q) {
        FlatRow<C> r;
        const float o[4] = {q.own.x, q.own.y, q.own.z, q.own.w};
        const float h[4] = {q.halo.x, q.halo.y, q.halo.z, q.halo.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) r.o[i] = o[i];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            r.p[i] = flat_shr(h[4 - C + i], o[4 - C + i]);           // floats -C+i .. of this vector = the previous vector's tail
            r.n[i] = flat_shl(h[i], o[i]);                           // floats 4+i = the next vector's head
        }
        if (edge_strip) {                                            // the two ends of a frame row: replicate the end pixel, or zero
#pragma unroll
            for (int i = 0; i < C; ++i) {
                if (first) r.p[i] = ZERO ? 0.0f : o[i];
                if (last) r.n[i] = ZERO ? 0.0f : o[4 - C + i];
            }
        }
        return r;
    };
    auto emit = [&](int32_t y, const FlatRow<C>& a, const FlatRow<C>& b, const FlatRow<C>& c) {
        const FlatRow<C>* rw[3] = {&a, &b, &c};
        float res[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float p[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                p[r][0] = (k - C >= 0) ? rw[r]->o[k - C >= 0 ? k - C : 0] : rw[r]->p[k < C ? k : 0];
                p[r][1] = rw[r]->o[k];
                p[r][2] = (k + C < 4) ? rw[r]->o[k + C < 4 ? k + C : 0] : rw[r]->n[k + C - 4 >= 0 ? k + C - 4 : 0];
            }
            res[k] = stencil_value(OP, p, strength, ZERO ? 1 : 0);
        }
        if (!GENERAL || (valid && y < y1)) __builtin_nontemporal_store(fv4{res[0], res[1], res[2], res[3]}, fout + (int64_t)y * n4 + col);
    };

    FlatRow<C> r0 = process(load(y0 - 1));
    FlatRow<C> r1 = process(load(y0));
    FlatRow<C> r2;
    FlatRaw q = load(y0 + 1);
    for (int32_t y = y0; y < y1; y += 3) {                           // three steps per trip: the rows rotate through r0, r1, r2 by name
        // (the row requested ahead is capped at y1, the last one this segment needs: past the end it re-reads that row from L1)
        r2 = process(q);
        q = load(y + 2 < y1 ? y + 2 : y1);
        emit(y, r0, r1, r2);
        r0 = process(q);
        q = load(y + 3 < y1 ? y + 3 : y1);
        emit(y + 1, r1, r2, r0);
        r1 = process(q);
        q = load(y + 4 < y1 ? y + 4 : y1);
        emit(y + 2, r2, r0, r1);
    }
}

template <int C, int OP, bool ZERO, bool GENERAL>
__global__ __launch_bounds__(256) void k_stencil_flat(const float* __restrict__ in, float* __restrict__ out, int32_t H, int32_t n4 /* W*C/4 */,
                                                       int32_t strips_x, int32_t segs_y, uint32_t total_waves, float strength) {
    // XCD-aware placement as in k_chain_tile: workgroup b runs on XCD b % 8; give every XCD one contiguous run of work
    const uint32_t groups = (total_waves + 3u) / 4u;
    const uint32_t per_xcd = (groups + 7u) / 8u;
    const uint32_t grp = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || grp >= groups) return;
    const uint32_t wv = grp * 4u + (threadIdx.x >> 6);      // (through readfirstlane -- strip / segment arithmetic in SGPRs -- measured 1-3 % slower here: profiles/r03_flat_stencil_scalar_index_ab.log)
    if (wv >= total_waves) return;
    const int lane = threadIdx.x & 63;
    const uint32_t strip = wv % (uint32_t)strips_x;
    const uint32_t rest = wv / (uint32_t)strips_x;
    const uint32_t seg = rest % (uint32_t)segs_y;
    const int64_t f = rest / (uint32_t)segs_y;
    int32_t c0 = (int32_t)strip * 64, y0 = (int32_t)seg * FLAT_ROWS, rows = FLAT_ROWS;
    if (!GENERAL) {                                                  // the last strip / segment overlap their neighbours instead of ending short
        c0 = c0 < n4 - 64 ? c0 : n4 - 64;
        y0 = y0 < H - FLAT_ROWS ? y0 : H - FLAT_ROWS;
    } else {
        rows = y0 + FLAT_ROWS < H ? FLAT_ROWS : H - y0;
    }
    const bool edge_strip = strip == 0 || (int32_t)strip == strips_x - 1;      // wave-uniform
    const fv4* fin = reinterpret_cast<const fv4*>(in) + f * (int64_t)H * n4;
    fv4* fout = reinterpret_cast<fv4*>(out) + f * (int64_t)H * n4;
    flat_march<C, OP, ZERO, GENERAL>(fin, fout, H, n4, c0 + lane, lane, edge_strip, y0, rows, strength);
}

template <int C>
static int launch_stencil_flat(const float* in, float* out, int64_t frames, int32_t H, int32_t W, int32_t op, int32_t border, float strength,
                               hipStream_t st) {
    const int32_t n4 = W * C / 4;
    const int32_t strips_x = (n4 + 63) / 64, segs_y = (H + FLAT_ROWS - 1) / FLAT_ROWS;
    const bool general = n4 < 64 || H < FLAT_ROWS;
    const int64_t per_frame = (int64_t)strips_x * segs_y;
    const int64_t step = ((int64_t)1 << 30) / per_frame;            // frames per launch: the wave count must fit 32 bits
    if (step < 1) return VRG_ERR_UNSUPPORTED;
    for (int64_t f0 = 0; f0 < frames; f0 += step) {
        const int64_t nf = frames - f0 < step ? frames - f0 : step;
        const uint32_t total = (uint32_t)(per_frame * nf);
        const uint32_t groups = (total + 3u) / 4u;
        const uint32_t blocks = ((groups + 7u) / 8u) * 8u;
        const float* src = in + f0 * (int64_t)H * W * C;
        float* dst = out + f0 * (int64_t)H * W * C;
#define VRG_FLAT(OPV, Z, G) hipLaunchKernelGGL((k_stencil_flat<C, OPV, Z, G>), dim3(blocks), dim3(256), 0, st, src, dst, H, n4, strips_x, segs_y, total, strength)
#define VRG_FLAT2(OPV, Z) do { if (general) VRG_FLAT(OPV, Z, true); else VRG_FLAT(OPV, Z, false); } while (0)
        const bool zero = border == VRG_BORDER_ZERO;
        if (op == 0) { if (zero) VRG_FLAT2(0, true); else VRG_FLAT2(0, false); }
        else if (op == 1) { if (zero) VRG_FLAT2(1, true); else VRG_FLAT2(1, false); }
        else { if (zero) VRG_FLAT2(2, true); else VRG_FLAT2(2, false); }
#undef VRG_FLAT2
#undef VRG_FLAT
        if (hipGetLastError() != hipSuccess) return VRG_ERR_LAUNCH;
    }
    return VRG_OK;
}

}  // namespace vrg

using namespace vrg;

extern "C" int vrg_stencil3x3_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width, int32_t channels,
                                  int32_t op, int32_t border, float strength, void* stream) {
    if (!in || !out || frames < 0 || height <= 0 || width <= 0 || channels <= 0 || op < 0 || op > 2 || border < 0 || border > 1)
        return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    // RGB / RGBA frames with whole float4 vectors per row and 16-byte aligned bases: the streaming flat march
    const bool flat_ok = (channels == 3 || channels == 4) && ((int64_t)width * channels) % 4 == 0 && (int64_t)width * channels / 4 <= 0x7fffffff / 64 &&
                         ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (flat_ok) {
        hipStream_t st = (hipStream_t)stream;
        if (channels == 3) return launch_stencil_flat<3>(in, out, frames, height, width, op, border, strength, st);
        return launch_stencil_flat<4>(in, out, frames, height, width, op, border, strength, st);
    }
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
