// vrg_pointwise.hip -- stand-alone per-pixel kernels: noise stream, film grain, 3D LUT, colour-match
// apply.  gfx950 only.  Every kernel is HBM-streaming: consecutive lanes touch consecutive addresses
// (12 B/lane pixel records or 16 B/lane element quads), nothing is re-read from HBM.
#include "vrg_common.hpp"

namespace vrg {

// ----------------------------------------------------------------------------------------------
// Raw torch.randn stream.  One thread = one (chunk, call k, subsequence idx): one Philox call, four
// normals, written to the four elements idx + G*(4k+ii) -- the same thread/element relation as
// ATen's distribution_elementwise_grid_stride_kernel, so the Philox work is the minimum possible.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_noise(float* __restrict__ out, NoiseK nk, int64_t chunk_numel,
                                                uint32_t groups_per_chunk /* K = ceil(numel/(4G)) */) {
    const uint32_t G = nk.G;
    const uint32_t blocks_per_group = (G + 255u) / 256u;
    const uint32_t bpc = blocks_per_group * groups_per_chunk;
    const int64_t chunk = blockIdx.x / bpc;
    const uint32_t rem = blockIdx.x - (uint32_t)chunk * bpc;
    const uint32_t k = rem / blocks_per_group;
    const uint32_t idx = (rem - k * blocks_per_group) * 256u + threadIdx.x;
    if (idx >= G) return;
    const uint64_t seed = chunk_seed(nk, chunk);
    const uint64_t ctr = (chunk_offset(nk, chunk) >> 2) + k;
    const u32x4 r = philox_for(seed, idx, ctr);
    const f32x2 a = box_muller(r.x, r.y);
    const f32x2 b = box_muller(r.z, r.w);
    const float n[4] = {a.x, a.y, b.x, b.y};
    float* base = out + chunk * chunk_numel;
    const int64_t li0 = (int64_t)4 * G * k + idx;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int64_t li = li0 + (int64_t)G * ii;
        if (li < chunk_numel) base[li] = n[ii];
    }
}

// ----------------------------------------------------------------------------------------------
// Film grain with in-register noise.  Block = 256 threads = 1024 consecutive Philox subsequences
// (4 per thread) of one call k of one chunk: 1024 Philox calls feed the 4096 elements
// {idx + G*(4k+ii)}.  Each element also needs the raw normal of its pixel's green element, which is
// the element itself or its left/right neighbour: the normals are exchanged through LDS, and the two
// neighbours outside the block's range (per sibling range) are produced by the general per-element
// routine on 8 lanes.
// ----------------------------------------------------------------------------------------------
#define VRG_GRAIN_NT 0
constexpr int GRAIN_IPT = 4;                    // subsequences per thread
constexpr int GRAIN_N = 256 * GRAIN_IPT;        // subsequences per block

// U8: decoded video frames, uint8 B,G,R per pixel (element li = 3 p + c of the reference's fp32 R,G,B tensor lives in byte
// 3 p + 2 - c); the / 255 and the * 255-clip-truncate of the frame converters happen at the load and the store.
template <bool VEC, bool U8 = false>
__global__ __launch_bounds__(256) void k_grain(const void* __restrict__ in_, void* __restrict__ out_, NoiseK nk,
                                                int64_t chunk_numel, uint32_t groups_per_chunk, float I, float S, float T) {
    const float* in = reinterpret_cast<const float*>(in_);
    float* out = reinterpret_cast<float*>(out_);
    __shared__ float sn[4][GRAIN_N + 8];
    const uint32_t G = nk.G;
    const uint32_t blocks_per_group = (G + GRAIN_N - 1) / GRAIN_N;
    const uint32_t bpc = blocks_per_group * groups_per_chunk;
    const int64_t chunk = blockIdx.x / bpc;
    const uint32_t rem = blockIdx.x - (uint32_t)chunk * bpc;
    const uint32_t k = rem / blocks_per_group;
    const uint32_t idx_base = (rem - k * blocks_per_group) * GRAIN_N;
    const uint32_t valid_n = (G - idx_base) < (uint32_t)GRAIN_N ? (G - idx_base) : (uint32_t)GRAIN_N;
    const uint32_t t4 = threadIdx.x * GRAIN_IPT;
    const uint64_t seed = chunk_seed(nk, chunk);
    const uint64_t off = chunk_offset(nk, chunk);
    const uint64_t ctr = (off >> 2) + k;

    // the frame data of this thread's four element quads is requested BEFORE the Philox rounds: the ~700 instructions of noise
    // synthesis then run under the HBM latency instead of in front of it (VEC form; the ragged forms load below)
    const int64_t group_base0 = (int64_t)4 * G * k + idx_base;
    float4 pre[4];
    bool pre_ok[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int64_t li0 = group_base0 + (int64_t)G * ii + t4;
        pre_ok[ii] = VEC && !U8 && t4 < valid_n && li0 + 3 < chunk_numel;
        if (VEC && !U8) {       // branch-free: lanes with nothing to load re-read the chunk's first quad (the result is not used)
            typedef float v4 __attribute__((ext_vector_type(4)));
            const v4* src = reinterpret_cast<const v4*>(in + chunk * chunk_numel + (pre_ok[ii] ? li0 : 0));
#if VRG_GRAIN_NT
            const v4 v = __builtin_nontemporal_load(src);
#else
            const v4 v = *src;
#endif
            pre[ii] = make_float4(v.x, v.y, v.z, v.w);
        } else {
            pre[ii] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }

    float nz[GRAIN_IPT][4];
#pragma unroll
    for (int j = 0; j < GRAIN_IPT; ++j) {
        const u32x4 r = philox_for(seed, idx_base + t4 + j, ctr);
        const f32x2 a = box_muller(r.x, r.y);
        const f32x2 b = box_muller(r.z, r.w);
        nz[j][0] = a.x; nz[j][1] = a.y; nz[j][2] = b.x; nz[j][3] = b.y;
    }
    if (t4 < valid_n) {   // valid_n is a multiple of 4 whenever G is (G = grid*256)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            float4 v = make_float4(nz[0][ii], nz[1][ii], nz[2][ii], nz[3][ii]);
            *reinterpret_cast<float4*>(&sn[ii][4 + t4]) = v;
        }
    }
    const int64_t group_base = (int64_t)4 * G * k + idx_base;   // chunk-local element of (ii=0, first idx)
    if (threadIdx.x < 8) {
        const int ii = threadIdx.x >> 1;
        const int right = threadIdx.x & 1;
        const int64_t li = group_base + (int64_t)G * ii + (right ? (int64_t)valid_n : -1);
        float v = 0.0f;
        if (li >= 0 && li < chunk_numel) v = torch_randn_element(seed, off, G, (uint64_t)li);
        sn[ii][right ? 4 + valid_n : 3] = v;
    }
    __syncthreads();
    if (t4 >= valid_n) return;

    const float* cin = in + chunk * chunk_numel;
    float* cout = out + chunk * chunk_numel;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int64_t li0 = group_base + (int64_t)G * ii + t4;
        if (li0 >= chunk_numel) continue;
        float x[4], o[4];
        const bool full = li0 + 3 < chunk_numel;
        int c = (int)((uint32_t)li0 % 3u);   // chunk_numel < 2^31 (checked by the entry point)
        const uint8_t* cin8 = reinterpret_cast<const uint8_t*>(in_) + chunk * chunk_numel;
        uint8_t* cout8 = reinterpret_cast<uint8_t*>(out_) + chunk * chunk_numel;
        if (U8) {
            int cc = c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[j] = (li0 + j < chunk_numel) ? unit_from_u8(cin8[li0 + j + 2 - 2 * cc]) : 0.0f;      // byte 3p + 2 - c = li + 2 - 2c
                cc = (cc == 2) ? 0 : cc + 1;
            }
        } else if (VEC && full) {
            const float4 v = pre[ii];          // == pre_ok[ii]: requested before the Philox rounds
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = (li0 + j < chunk_numel) ? cin[li0 + j] : 0.0f;
        }
        const int c_first = c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ng = sn[ii][4 + t4 + j + 1 - c];   // c==1: itself
            o[j] = grain_element(x[j], nz[j][ii], ng, c, I, S, T);
            c = (c == 2) ? 0 : c + 1;
        }
        if (U8) {
            int cc = c_first;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (li0 + j < chunk_numel) cout8[li0 + j + 2 - 2 * cc] = u8_from_unit(o[j]);
                cc = (cc == 2) ? 0 : cc + 1;
            }
        } else if (VEC && full) {
#if VRG_GRAIN_NT
            typedef float v4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(v4{o[0], o[1], o[2], o[3]}, reinterpret_cast<v4*>(cout + li0));
#else
            *reinterpret_cast<float4*>(cout + li0) = make_float4(o[0], o[1], o[2], o[3]);
#endif
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (li0 + j < chunk_numel) cout[li0 + j] = o[j];
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Unsharp -> per-frame-seeded grain in one pass (the stand-alone enhancer's order, VRGDG_StandaloneVideoEnhancerNodes.py:278-294):
// 24 B/px of HBM traffic instead of the 48 of the two kernels.  The GRAIN geometry leads -- a block is the same 1024 Philox
// subsequences x 4 sibling element runs (G apart) as in k_grain, so the noise still costs one Philox call per four elements for
// every frame size -- and the stencil follows it: a thread's four consecutive floats of a run are one float4 of the row-major
// frame, so each wave holds 64 consecutive vectors of each run and builds the 3x3 neighbourhoods exactly like the flat march of
// vrg_stencil.hip (left / right taps by DPP wave shifts from the neighbouring lanes, lanes 0 and 63 from one extra load), only
// without the vertical register reuse: the rows above and below are other blocks' centre rows and come from L2 (the workgroups of
// one XCD walk consecutive segments, so the three rows of a segment are resident in that XCD's L2 when it needs them).  A wave may
// straddle a row end (any W with W*3 % 4 == 0): row / column are per lane, the row-end rule is a per-lane select.
// Needs W % 4 == 0, W*3/4 >= 256, one frame per noise chunk, 16-byte aligned frames; otherwise the entry point reports
// VRG_ERR_UNSUPPORTED and the caller runs the two kernels.  Same arithmetic as both (stencil_value, grain_element): bit-identical.
// ----------------------------------------------------------------------------------------------
typedef float sg4 __attribute__((ext_vector_type(4)));
struct SgRaw { sg4 own[3], halo[3]; };

__device__ __forceinline__ float sg_shr(float old, float v) {    // value of lane-1; lane 0 keeps `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float sg_shl(float old, float v) {    // value of lane+1; lane 63 keeps `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

template <bool ZERO>
__global__ __launch_bounds__(256) void k_sharpen_grain(const float* __restrict__ in, float* __restrict__ out, NoiseK nk, int32_t H, int32_t n4,
                                                        uint32_t groups_per_frame, uint32_t total_blocks, float strength, float I, float S,
                                                        float T) {
    __shared__ float sn[4][GRAIN_N + 8];
    // workgroup b runs on XCD b % 8: every XCD gets one contiguous run of (frame, call, segment) blocks
    const uint32_t per_xcd = (total_blocks + 7u) >> 3;
    const uint32_t b = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || b >= total_blocks) return;
    const uint32_t G = nk.G;
    const uint32_t segs = (G + GRAIN_N - 1) / GRAIN_N;
    const uint32_t bpf = segs * groups_per_frame;
    const uint32_t frame = b / bpf;
    const uint32_t rem = b - frame * bpf;
    const uint32_t k = rem / segs;
    const uint32_t idx_base = (rem - k * segs) * GRAIN_N;
    const uint32_t valid_n = (G - idx_base) < (uint32_t)GRAIN_N ? (G - idx_base) : (uint32_t)GRAIN_N;      // a multiple of 256: whole waves
    const uint32_t tid = threadIdx.x, t4 = tid * GRAIN_IPT;
    const int lane = (int)(tid & 63u);
    const uint64_t seed = chunk_seed(nk, frame);
    const uint64_t off = chunk_offset(nk, frame);
    const uint64_t ctr = (off >> 2) + k;
    const uint32_t nvec = (uint32_t)H * (uint32_t)n4;                 // float4 vectors per frame (< 2^29)
    const sg4* fin = reinterpret_cast<const sg4*>(in) + (int64_t)frame * nvec;
    sg4* fout = reinterpret_cast<sg4*>(out) + (int64_t)frame * nvec;
    const uint32_t vec0 = (4u * G * k + idx_base) >> 2;               // first vector of sibling run 0 (4 G k < frame elements < 2^31)

    // the three rows around this thread's vector of sibling run ii, requested in one go (branch-free: lanes that need no halo vector
    // re-read their own, lanes past the frame end read its last row)
    auto request = [&](int ii, SgRaw& q, uint32_t& v_out, int32_t& y_out, int32_t& col_out) {
        const uint32_t vb = vec0 + (G >> 2) * (uint32_t)ii;           // block-uniform
        const uint32_t yb = vb / (uint32_t)n4;
        int32_t col = (int32_t)(vb - yb * (uint32_t)n4 + tid);
        int32_t y = (int32_t)yb;
        if (col >= n4) { col -= n4; y += 1; }                          // n4 >= 256: at most one row end inside a block
        v_out = vb + tid;
        y = y < H ? y : H - 1;
        const int32_t yu = y > 0 ? y - 1 : 0, yd = y < H - 1 ? y + 1 : H - 1;
        const int32_t hc = (lane == 0 && col > 0) ? col - 1 : ((lane == 63 && col + 1 < n4) ? col + 1 : col);
        const int32_t ys[3] = {yu, y, yd};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const sg4* row = fin + (int64_t)ys[r] * n4;
            q.own[r] = row[col];
            q.halo[r] = row[hc];
        }
        y_out = y;
        col_out = col;
    };

#define VRG_SG_PIPE 0         /* 1: request run ii + 1's rows before run ii is computed (81 instead of 58 VGPRs); measured equal (6.71 / 6.82 against 6.75 / 6.80 ms per 128 4K frames, profiles/r03_sharpen_grain_fused_issue.log): the kernel does not wait on its loads */
    SgRaw qq[2];
    uint32_t vv[2];
    int32_t yy[2], cc[2];
    request(0, qq[0], vv[0], yy[0], cc[0]);                           // in flight under the Philox rounds

    float nz[GRAIN_IPT][4];
#pragma unroll
    for (int j = 0; j < GRAIN_IPT; ++j) {
        const u32x4 r = philox_for(seed, idx_base + t4 + j, ctr);
        const f32x2 a = box_muller(r.x, r.y);
        const f32x2 bb = box_muller(r.z, r.w);
        nz[j][0] = a.x; nz[j][1] = a.y; nz[j][2] = bb.x; nz[j][3] = bb.y;
    }
    if (t4 < valid_n) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) *reinterpret_cast<float4*>(&sn[ii][4 + t4]) = make_float4(nz[0][ii], nz[1][ii], nz[2][ii], nz[3][ii]);
    }
    const int64_t fe = (int64_t)nvec * 4;
    const int64_t group_base = (int64_t)4 * G * k + idx_base;
    if (tid < 8) {                                                    // the green normals just outside the block's four runs
        const int ii = (int)(tid >> 1);
        const int right = (int)(tid & 1);
        const int64_t li = group_base + (int64_t)G * ii + (right ? (int64_t)valid_n : -1);
        float nv = 0.0f;
        if (li >= 0 && li < fe) nv = torch_randn_element(seed, off, G, (uint64_t)li);
        sn[ii][right ? 4 + valid_n : 3] = nv;
    }
    __syncthreads();
    if (t4 >= valid_n) return;                                        // whole waves

#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int cur = VRG_SG_PIPE ? (ii & 1) : 0;
        if (!VRG_SG_PIPE && ii > 0) request(ii, qq[0], vv[0], yy[0], cc[0]);
        if (VRG_SG_PIPE && ii < 3) request(ii + 1, qq[(ii + 1) & 1], vv[(ii + 1) & 1], yy[(ii + 1) & 1], cc[(ii + 1) & 1]);   // clamped: in bounds even past the frame
        if (vec0 + (G >> 2) * (uint32_t)ii >= nvec) continue;         // block-uniform: this run starts past the frame
        const SgRaw& q = qq[cur];
        const uint32_t v = vv[cur];
        const int32_t y = yy[cur], col = cc[cur];
        const bool first = col == 0, last = col == n4 - 1;
        // wave-uniform: does any lane of this wave sit at a row end / (zero border) on the frame's first or last row?  2 waves in 45 at 4K
        const bool row_end_here = __builtin_amdgcn_ballot_w64(first || last) != 0;
        const bool frame_edge_here = ZERO && __builtin_amdgcn_ballot_w64(y == 0 || y == H - 1) != 0;
        float o[3][4], pl[3][3], nr[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float ow[4] = {q.own[r].x, q.own[r].y, q.own[r].z, q.own[r].w};
            float hw[4] = {q.halo[r].x, q.halo[r].y, q.halo[r].z, q.halo[r].w};
            if (ZERO && r != 1 && frame_edge_here) {                  // avg_pool2d(padding=1): the rows outside the frame are zeros
                const bool outside = (r == 0 && y == 0) || (r == 2 && y == H - 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) { ow[i] = outside ? 0.0f : ow[i]; hw[i] = outside ? 0.0f : hw[i]; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) o[r][i] = ow[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pl[r][i] = sg_shr(hw[1 + i], ow[1 + i]);              // floats -3 + i of this vector = the previous vector's tail
                nr[r][i] = sg_shl(hw[i], ow[i]);                      // floats 4 + i = the next vector's head
            }
            if (row_end_here) {                                       // row ends: replicate the end pixel, or zero
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    pl[r][i] = first ? (ZERO ? 0.0f : ow[i]) : pl[r][i];
                    nr[r][i] = last ? (ZERO ? 0.0f : ow[1 + i]) : nr[r][i];
                }
            }
        }
        float x[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float p[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                p[r][0] = kk >= 3 ? o[r][kk >= 3 ? kk - 3 : 0] : pl[r][kk < 3 ? kk : 0];
                p[r][1] = o[r][kk];
                p[r][2] = kk < 1 ? o[r][kk < 1 ? kk + 3 : 0] : nr[r][kk >= 1 ? kk - 1 : 0];
            }
            x[kk] = stencil_value(0, p, strength, ZERO ? 1 : 0);
        }
        int c = (int)((4u * v) % 3u);                                 // channel of the vector's first float (frame-local element 4 v)
        float res[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ng = sn[ii][4 + t4 + j + 1 - c];              // c == 1: itself
            res[j] = grain_element(x[j], nz[j][ii], ng, c, I, S, T);
            c = (c == 2) ? 0 : c + 1;
        }
        if (v < nvec) fout[v] = sg4{res[0], res[1], res[2], res[3]};
    }
}

// ----------------------------------------------------------------------------------------------
// The same pass on DECODED frames: uint8 B,G,R in -> / 255 -> unsharp -> per-frame-seeded grain -> * 255 clip truncate -> uint8 B,G,R
// out, i.e. _tensor_to_frames(_apply_effects_batch(_frames_to_tensor(frames))) of the stand-alone enhancer's render loop
// (VRGDG_StandaloneVideoEnhancerNodes.py:311-324, 278-294, 417-421) in ONE kernel: 3 + 3 B/px of HBM traffic where the converter ->
// fused fp32 kernel -> converter route moves 3 + 12 | 12 + 12 | 12 + 3.  Same geometry as k_sharpen_grain with BYTES in the place of
// floats: element li = 3 p + c of the reference's fp32 R,G,B tensor is byte 3 p + 2 - c of the frame, so element and byte indices
// cover the same ranges and a block's four sibling runs of 1024 elements are four runs of 1024 bytes = 256 dwords; a thread owns
// one dword (four bytes) of each run.  The 3x3 window is built per channel in byte space (horizontal taps 3 bytes away: the
// previous / next dword of the row come from the neighbouring lanes with ONE DPP wave shift of the raw dword each, lanes 0 / 63 from
// one extra load), every tap is converted with the reference's own v / 255 (unit_from_u8); the grain of byte 3 p + j uses the
// normal of element 3 p + 2 - j and the green normal of element 3 p + 1, both read from the block's staged normals (two halo
// elements per side and run instead of one).  Arithmetic: unit_from_u8, stencil_value, grain_element, u8_from_unit -- the functions
// the three-kernel route evaluates, in its order: byte-identical to it.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sg_shr_u(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t sg_shl_u(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
struct SgRawU8 { uint32_t own[3], halo[3]; };

template <bool ZERO>
__global__ __launch_bounds__(256) void k_sharpen_grain_u8(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, NoiseK nk, int32_t H,
                                                           int32_t n4, uint32_t groups_per_frame, uint32_t total_blocks, float strength,
                                                           float I, float S, float T) {
    __shared__ float sn[4][GRAIN_N + 8];                              // [run][4 + element of the run]; halo elements at 2, 3 and 4 + valid_n, 5 + valid_n
    const uint32_t per_xcd = (total_blocks + 7u) >> 3;
    const uint32_t b = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || b >= total_blocks) return;
    const uint32_t G = nk.G;
    const uint32_t segs = (G + GRAIN_N - 1) / GRAIN_N;
    const uint32_t bpf = segs * groups_per_frame;
    const uint32_t frame = b / bpf;
    const uint32_t rem = b - frame * bpf;
    const uint32_t k = rem / segs;
    const uint32_t idx_base = (rem - k * segs) * GRAIN_N;
    const uint32_t valid_n = (G - idx_base) < (uint32_t)GRAIN_N ? (G - idx_base) : (uint32_t)GRAIN_N;      // a multiple of 256: whole waves
    const uint32_t tid = threadIdx.x, t4 = tid * GRAIN_IPT;
    const int lane = (int)(tid & 63u);
    const uint64_t seed = chunk_seed(nk, frame);
    const uint64_t off = chunk_offset(nk, frame);
    const uint64_t ctr = (off >> 2) + k;
    const uint32_t nvec = (uint32_t)H * (uint32_t)n4;                 // dwords per frame
    const uint32_t* fin = in + (int64_t)frame * nvec;
    uint32_t* fout = out + (int64_t)frame * nvec;
    const uint32_t vec0 = (4u * G * k + idx_base) >> 2;

    auto request = [&](int ii, SgRawU8& q, uint32_t& v_out, int32_t& y_out, int32_t& col_out) {
        const uint32_t vb = vec0 + (G >> 2) * (uint32_t)ii;           // block-uniform
        const uint32_t yb = vb / (uint32_t)n4;
        int32_t col = (int32_t)(vb - yb * (uint32_t)n4 + tid);
        int32_t y = (int32_t)yb;
        if (col >= n4) { col -= n4; y += 1; }                          // n4 >= 256: at most one row end inside a block
        v_out = vb + tid;
        y = y < H ? y : H - 1;
        const int32_t yu = y > 0 ? y - 1 : 0, yd = y < H - 1 ? y + 1 : H - 1;
        const int32_t hc = (lane == 0 && col > 0) ? col - 1 : ((lane == 63 && col + 1 < n4) ? col + 1 : col);
        const int32_t ys[3] = {yu, y, yd};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint32_t* row = fin + (int64_t)ys[r] * n4;
            q.own[r] = row[col];
            q.halo[r] = row[hc];
        }
        y_out = y;
        col_out = col;
    };

    SgRawU8 q;
    uint32_t v;
    int32_t y, col;
    request(0, q, v, y, col);                                         // in flight under the Philox rounds

    float nz[GRAIN_IPT][4];
#pragma unroll
    for (int j = 0; j < GRAIN_IPT; ++j) {
        const u32x4 r = philox_for(seed, idx_base + t4 + j, ctr);
        const f32x2 a = box_muller(r.x, r.y);
        const f32x2 bb = box_muller(r.z, r.w);
        nz[j][0] = a.x; nz[j][1] = a.y; nz[j][2] = bb.x; nz[j][3] = bb.y;
    }
    if (t4 < valid_n) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) *reinterpret_cast<float4*>(&sn[ii][4 + t4]) = make_float4(nz[0][ii], nz[1][ii], nz[2][ii], nz[3][ii]);
    }
    const int64_t fe = (int64_t)nvec * 4;
    const int64_t group_base = (int64_t)4 * G * k + idx_base;
    if (tid < 16) {                                                   // the two normals on either side of the block's four runs
        const int ii = (int)(tid >> 2);
        const int right = (int)((tid >> 1) & 1), d = (int)(tid & 1);
        const int64_t li = group_base + (int64_t)G * ii + (right ? (int64_t)valid_n + d : -1 - d);
        float nv = 0.0f;
        if (li >= 0 && li < fe) nv = torch_randn_element(seed, off, G, (uint64_t)li);
        sn[ii][right ? 4 + valid_n + d : 3 - d] = nv;
    }
    __syncthreads();
    if (t4 >= valid_n) return;                                        // whole waves

#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        if (ii > 0) request(ii, q, v, y, col);
        if (vec0 + (G >> 2) * (uint32_t)ii >= nvec) continue;         // block-uniform: this run starts past the frame
        const bool first = col == 0, last = col == n4 - 1;
        const bool row_end_here = __builtin_amdgcn_ballot_w64(first || last) != 0;
        const bool frame_edge_here = ZERO && __builtin_amdgcn_ballot_w64(y == 0 || y == H - 1) != 0;
        float o[3][4], pl[3][3], nr[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            uint32_t ow = q.own[r], hw = q.halo[r];
            if (ZERO && r != 1 && frame_edge_here) {                  // avg_pool2d(padding=1): the rows outside the frame are zeros (byte 0 -> 0.0f)
                const bool outside = (r == 0 && y == 0) || (r == 2 && y == H - 1);
                ow = outside ? 0u : ow;
                hw = outside ? 0u : hw;
            }
            const uint32_t prev = sg_shr_u(hw, ow), next = sg_shl_u(hw, ow);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[r][i] = unit_from_u8((uint8_t)(ow >> (8 * i)));
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pl[r][i] = unit_from_u8((uint8_t)(prev >> (8 * (1 + i))));   // bytes -3 + i of this dword = the previous dword's tail
                nr[r][i] = unit_from_u8((uint8_t)(next >> (8 * i)));         // bytes 4 + i = the next dword's head
            }
            if (row_end_here) {                                       // row ends: replicate the end pixel, or zero
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    pl[r][i] = first ? (ZERO ? 0.0f : o[r][i]) : pl[r][i];
                    nr[r][i] = last ? (ZERO ? 0.0f : o[r][1 + i]) : nr[r][i];
                }
            }
        }
        int jj = (int)((4u * v) % 3u);                                // position of the dword's first byte in its pixel: 0 = B, 1 = G, 2 = R
        uint32_t packed = 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float p[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                p[r][0] = kk >= 3 ? o[r][kk >= 3 ? kk - 3 : 0] : pl[r][kk < 3 ? kk : 0];
                p[r][1] = o[r][kk];
                p[r][2] = kk < 1 ? o[r][kk < 1 ? kk + 3 : 0] : nr[r][kk >= 1 ? kk - 1 : 0];
            }
            const float x = stencil_value(0, p, strength, ZERO ? 1 : 0);
            const float n_own = sn[ii][4 + t4 + kk + 2 - 2 * jj];     // element 3 p + 2 - jj of byte 3 p + jj
            const float n_green = sn[ii][4 + t4 + kk + 1 - jj];       // element 3 p + 1
            const float res = grain_element(x, n_own, n_green, 2 - jj, I, S, T);
            packed |= (uint32_t)u8_from_unit(res) << (8 * kk);
            jj = (jj == 2) ? 0 : jj + 1;
        }
        if (v < nvec) fout[v] = packed;
    }
}

// ----------------------------------------------------------------------------------------------
// The same pass for ANY frame size and alignment (round 5): widths that are not a multiple of 4 (854, 1366), rows shorter than a block's
// 1024 bytes, frames whose byte count is not a multiple of 4 (so that later frames start off the dword grid), down to 1 x 1 -- everything
// k_sharpen_grain_u8 above leaves to the three-kernel route.  Same geometry (a thread owns four consecutive bytes of each of the block's
// four sibling runs; the previous / next four bytes come from the neighbouring lanes by one DPP shift), but in FLAT BYTE space of the
// whole batch instead of on the frame's dword grid:
//   * the bytes above / below byte a are bytes a - E / a + E (E = 3 W bytes per row) whatever the row alignment: three UNALIGNED dword loads
//     per run and thread (the hardware takes them; windows that would leave the batch are fetched from a clamped address and shifted into
//     place -- the bytes that do not exist are never used, see below);
//   * row and column are per BYTE (one 32-bit division per thread and run, a conditional step per byte): a thread's four bytes may
//     straddle a row end, several of them when rows are shorter than four bytes;
//   * the border rules are applied per byte on the assembled 3 x 3 window -- left / right first, then top / bottom, which is
//     np.pad(mode="edge") (the corner tap is the pixel itself) resp. avg_pool2d's zero padding -- in wave-uniform branches only the waves
//     that touch a border take;
//   * a frame's last bytes (a byte count that is no multiple of 4) are stored byte by byte: the next frame's first bytes are its own.
// Every tap that an edge rule replaces may hold garbage (another frame's bytes, zeros of a clamped window): it never reaches the
// arithmetic.  Same device functions in the same order as the fast kernel: byte-identical to it and to the three-kernel route.
// ----------------------------------------------------------------------------------------------
template <bool ZERO>
__global__ __launch_bounds__(256) void k_sharpen_grain_u8_any(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, NoiseK nk, int32_t H,
                                                               int32_t E, uint32_t fe, int64_t total_bytes, uint32_t groups_per_frame,
                                                               uint32_t total_blocks, float strength, float I, float S, float T) {
    __shared__ float sn[4][GRAIN_N + 8];
    const uint32_t per_xcd = (total_blocks + 7u) >> 3;
    const uint32_t b = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || b >= total_blocks) return;
    const uint32_t G = nk.G;
    const uint32_t segs = (G + GRAIN_N - 1) / GRAIN_N;
    const uint32_t bpf = segs * groups_per_frame;
    const uint32_t frame = b / bpf;
    const uint32_t rem = b - frame * bpf;
    const uint32_t k = rem / segs;
    const uint32_t idx_base = (rem - k * segs) * GRAIN_N;
    const uint32_t valid_n = (G - idx_base) < (uint32_t)GRAIN_N ? (G - idx_base) : (uint32_t)GRAIN_N;
    const uint32_t tid = threadIdx.x, t4 = tid * GRAIN_IPT;
    const int lane = (int)(tid & 63u);
    const uint64_t seed = chunk_seed(nk, frame);
    const uint64_t off = chunk_offset(nk, frame);
    const uint64_t ctr = (off >> 2) + k;
    const int64_t fbase = (int64_t)frame * (int64_t)fe;               // the frame's first byte in the batch
    const uint32_t a_first = 4u * G * k + idx_base;                    // frame-relative byte (= element) of run 0's first element in this block

    // bytes [g, g + 4) of the batch; positions outside it read as unspecified values
    auto load4 = [&](int64_t g) -> uint32_t {
        const int64_t hi = total_bytes - 4;
        const int64_t lo = g < 0 ? 0 : (g > hi ? hi : g);
        uint32_t d;
        __builtin_memcpy(&d, in + lo, 4);
        const int sh = (int)(g - lo);                                  // > 0: the wanted window starts above the loaded one
        if (sh > 0) d = sh >= 4 ? 0u : d >> (8 * sh);
        else if (sh < 0) d = sh <= -4 ? 0u : d << (8 * (-sh));
        return d;
    };
    struct Raw { uint32_t own[3], halo[3]; uint32_t a0; int32_t y0, c0; };
    auto request = [&](int ii, Raw& q) {
        q.a0 = a_first + G * (uint32_t)ii + 4u * tid;
        const uint32_t a = q.a0 < fe ? q.a0 : (fe - 1u);               // (threads past the frame compute nothing that is stored)
        q.y0 = (int32_t)(a / (uint32_t)E);
        q.c0 = (int32_t)(a - (uint32_t)q.y0 * (uint32_t)E);
        const int64_t g = fbase + (int64_t)q.a0;
        const int64_t hoff = lane == 0 ? -4 : (lane == 63 ? 4 : 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int64_t gr = g + (int64_t)(r - 1) * E;
            q.own[r] = load4(gr);
            q.halo[r] = load4(gr + hoff);
        }
    };

    Raw q;
    request(0, q);                                                    // in flight under the Philox rounds

    float nz[GRAIN_IPT][4];
#pragma unroll
    for (int j = 0; j < GRAIN_IPT; ++j) {
        const u32x4 r = philox_for(seed, idx_base + t4 + j, ctr);
        const f32x2 a = box_muller(r.x, r.y);
        const f32x2 bb = box_muller(r.z, r.w);
        nz[j][0] = a.x; nz[j][1] = a.y; nz[j][2] = bb.x; nz[j][3] = bb.y;
    }
    if (t4 < valid_n) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) *reinterpret_cast<float4*>(&sn[ii][4 + t4]) = make_float4(nz[0][ii], nz[1][ii], nz[2][ii], nz[3][ii]);
    }
    const int64_t group_base = (int64_t)4 * G * k + idx_base;
    if (tid < 16) {                                                   // the two normals on either side of the block's four runs
        const int ii = (int)(tid >> 2);
        const int right = (int)((tid >> 1) & 1), d = (int)(tid & 1);
        const int64_t li = group_base + (int64_t)G * ii + (right ? (int64_t)valid_n + d : -1 - d);
        float nv = 0.0f;
        if (li >= 0 && li < (int64_t)fe) nv = torch_randn_element(seed, off, G, (uint64_t)li);
        sn[ii][right ? 4 + valid_n + d : 3 - d] = nv;
    }
    __syncthreads();
    if (t4 >= valid_n) return;                                        // whole waves

#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        if (ii > 0) request(ii, q);
        if (a_first + G * (uint32_t)ii >= fe) continue;               // block-uniform: this run starts past the frame
        // per byte: column (in bytes) and row, and which border rules apply
        int32_t colk[4], yk[4];
        bool lft[4], rgt[4], top[4], bot[4];
        bool any_lr = false, any_tb = false;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            int32_t c = q.c0 + kk, y = q.y0;
            if (c >= E) { c -= E; y += 1; }                            // E >= 3: at most one row end inside c0 + 3
            if (c >= E) { c -= E; y += 1; }                            // (E == 3: two)
            colk[kk] = c; yk[kk] = y;
            lft[kk] = c < 3; rgt[kk] = c >= E - 3; top[kk] = y == 0; bot[kk] = y >= H - 1;
            any_lr = any_lr || lft[kk] || rgt[kk];
            any_tb = any_tb || top[kk] || bot[kk];
        }
        const bool row_end_here = __builtin_amdgcn_ballot_w64(any_lr) != 0;
        const bool frame_edge_here = __builtin_amdgcn_ballot_w64(any_tb) != 0;
        float o[3][4], pl[3][3], nr[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint32_t ow = q.own[r], hw = q.halo[r];
            const uint32_t prev = sg_shr_u(hw, ow), next = sg_shl_u(hw, ow);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[r][i] = unit_from_u8((uint8_t)(ow >> (8 * i)));
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pl[r][i] = unit_from_u8((uint8_t)(prev >> (8 * (1 + i))));
                nr[r][i] = unit_from_u8((uint8_t)(next >> (8 * i)));
            }
        }
        int jj = (int)(q.a0 % 3u);                                     // position of the first byte in its pixel: 0 = B, 1 = G, 2 = R
        uint32_t packed = 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float p[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                p[r][0] = kk >= 3 ? o[r][kk >= 3 ? kk - 3 : 0] : pl[r][kk < 3 ? kk : 0];
                p[r][1] = o[r][kk];
                p[r][2] = kk < 1 ? o[r][kk < 1 ? kk + 3 : 0] : nr[r][kk >= 1 ? kk - 1 : 0];
            }
            if (row_end_here) {                                       // left / right first ...
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    p[r][0] = lft[kk] ? (ZERO ? 0.0f : p[r][1]) : p[r][0];
                    p[r][2] = rgt[kk] ? (ZERO ? 0.0f : p[r][1]) : p[r][2];
                }
            }
            if (frame_edge_here) {                                    // ... then top / bottom: the corner tap of the replicate border is the pixel itself
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    p[0][j] = top[kk] ? (ZERO ? 0.0f : p[1][j]) : p[0][j];
                    p[2][j] = bot[kk] ? (ZERO ? 0.0f : p[1][j]) : p[2][j];
                }
            }
            const float x = stencil_value(0, p, strength, ZERO ? 1 : 0);
            const float n_own = sn[ii][4 + t4 + kk + 2 - 2 * jj];     // element 3 p + 2 - jj of byte 3 p + jj
            const float n_green = sn[ii][4 + t4 + kk + 1 - jj];       // element 3 p + 1
            const float res = grain_element(x, n_own, n_green, 2 - jj, I, S, T);
            packed |= (uint32_t)u8_from_unit(res) << (8 * kk);
            jj = (jj == 2) ? 0 : jj + 1;
        }
        uint8_t* dst = out + fbase + (int64_t)q.a0;
        if (q.a0 + 4u <= fe) {
            __builtin_memcpy(dst, &packed, 4);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (q.a0 + (uint32_t)kk < fe) dst[kk] = (uint8_t)(packed >> (8 * kk));
        }
        (void)colk; (void)yk;
    }
}

// Noise-injection form: out = grain(x, noise) with caller-supplied normals (one pixel per thread).
__global__ __launch_bounds__(256) void k_grain_injected(const px3* __restrict__ in, const px3* __restrict__ nz,
                                                         px3* __restrict__ out, int64_t pixels, float I, float S, float T) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= pixels) return;
    const px3 v = in[p];
    const px3 n = nz[p];
    const float x[3] = {v.r, v.g, v.b};
    const float nn[3] = {n.r, n.g, n.b};
    float o[3];
    grain_pixel(x, nn, I, S, T, o);
    out[p] = px3{o[0], o[1], o[2]};
}

// ----------------------------------------------------------------------------------------------
// 3D LUT.  One pixel per thread.  The table does not fit the 160 KB LDS in fp32 (33^3*12 B = 431 KB,
// SURVEY.md section 7), so it is served by the vector L1 / per-XCD L2 -- in cell-major form (see
// vrg_pixel_math.hpp): one contiguous 96-byte run per pixel instead of eight scattered corners.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lut_build_cells(const float* __restrict__ table, int n, float* __restrict__ cells) {
    const int nc = n - 1;
    const int rec = blockIdx.x * 256 + threadIdx.x;
    if (rec >= nc * nc * n) return;
    const int r = rec % n, g0 = (rec / n) % nc, b0 = rec / (n * nc);
    float v[LUT_REC_FLOATS];
    lut_build_record(table, n, b0, g0, r, v);
    f32x4* dst = reinterpret_cast<f32x4*>(cells + (size_t)rec * LUT_REC_FLOATS);
#pragma unroll
    for (int i = 0; i < 3; ++i) dst[i] = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
}

template <bool RGB_ONLY>
__global__ __launch_bounds__(256) void k_lut3d(const float* __restrict__ in, float* __restrict__ out, int64_t pixels,
                                                int channels, LutParams P) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= pixels) return;
    float x[3], o[3];
    if (RGB_ONLY) {
        const px3 v = load_px_stream(reinterpret_cast<const px3*>(in) + p);
        x[0] = v.r; x[1] = v.g; x[2] = v.b;
    } else {
        const float* s = in + p * channels;
        x[0] = s[0]; x[1] = s[1]; x[2] = s[2];
    }
    lut_pixel(P, x, o);
    if (RGB_ONLY) {
        store_px_stream(reinterpret_cast<px3*>(out) + p, px3{o[0], o[1], o[2]});
    } else {
        float* d = out + p * channels;
        d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
        const float* s = in + p * channels;
        // channels beyond RGB: image.clone() pass-through in _apply_cube_lut (IV_Adjustments.py:341-343); the strength blend of
        // apply_lut then runs over ALL channels, a*(1-B) + a*B (:355-359), which is not always a
        for (int c = 3; c < channels; ++c) d[c] = P.blend_mode == 2 ? lerp2(s[c], P.one_minus_blend, s[c], P.blend) : s[c];
    }
}

// ----------------------------------------------------------------------------------------------
// Small cubes (N^3 * 16 B fits the LDS: N <= 21, e.g. the common 17^3): the whole LUT lives in LDS, one float4 per
// grid node, and a pixel's eight corners are eight ds_read_b128 -- no trip through the L1 / L2 gather path that bounds
// the global-memory kernel.  Persistent workgroups (as many as fit next to the table) walk the pixels grid-stride, so
// the table is staged once per workgroup.  Same corner order and arithmetic as lut_pixel (bit-identical results).
// The node values are read back out of the record table (vrg_lut_prepare_f32), so the ABI needs no second layout.
// ----------------------------------------------------------------------------------------------
template <class IO>
__global__ __launch_bounds__(1024) void k_lut3d_lds(const typename IO::elem* __restrict__ in, typename IO::elem* __restrict__ out,
                                                     int64_t pixels, LutParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds_nodes[];          // [b][g][r] x {R, G, B, pad}
    lut_nodes_to_lds(P, reinterpret_cast<f32x4*>(lds_nodes), (int)threadIdx.x, 1024);
    __syncthreads();
    const f32x4* T = reinterpret_cast<const f32x4*>(lds_nodes);
    for (int64_t p = (int64_t)blockIdx.x * 1024 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 1024) {
        const px3 v = IO::load_stream(in + p);
        const float x[3] = {v.r, v.g, v.b};
        float o[3];
        lut_pixel_nodes(P, T, x, o);
        IO::store_stream(out + p, px3{o[0], o[1], o[2]});
    }
}

// the table must fit the LDS next to nothing else, and there must be enough pixels to pay for staging it
bool lut_lds_applicable(int lut_size, int64_t pixels) {
    return (size_t)lut_size * lut_size * lut_size * 16 <= 152 * 1024 && pixels >= (1 << 16);
}

template <class IO>
static int launch_lut_lds_t(const void* in, void* out, int64_t pixels, const LutParams& P, hipStream_t st) {
    const size_t lds_bytes = (size_t)P.n * P.n * P.n * 16;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return VRG_ERR_NO_DEVICE;
    const int per_cu = lds_bytes <= 78 * 1024 ? 2 : 1;      // persistent 1024-thread workgroups, as many per CU as fit
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_lut3d_lds<IO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) !=
        hipSuccess)
        return VRG_ERR_LAUNCH;
    hipLaunchKernelGGL(k_lut3d_lds<IO>, dim3((uint32_t)(cus * per_cu)), dim3(1024), lds_bytes, st,
                       reinterpret_cast<const typename IO::elem*>(in), reinterpret_cast<typename IO::elem*>(out), pixels, P);
    return hipGetLastError() == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH;
}

int launch_lut_lds(const void* in, void* out, int64_t pixels, const LutParams& P, bool u8, hipStream_t st) {
    return u8 ? launch_lut_lds_t<IoU8>(in, out, pixels, P, st) : launch_lut_lds_t<IoF32>(in, out, pixels, P, st);
}

// ----------------------------------------------------------------------------------------------
// Colour-match apply (pass 2).  ms arrays: [frame][3][2] = {mean, std + 1e-5}.
// ----------------------------------------------------------------------------------------------
template <bool FAST>
__global__ __launch_bounds__(256) void k_colormatch_apply(const px3* __restrict__ in, px3* __restrict__ out,
                                                           int32_t pixels_per_frame, CmK cm, DevMath dm) {
    VRG_CM_MATH(PT, true, FAST, dm);
    const int32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= pixels_per_frame) return;
    const int64_t f = blockIdx.y;
    const int64_t at = f * pixels_per_frame + p;
    const px3 v = load_px_stream(in + at);
    const float x[3] = {v.r, v.g, v.b};
    const float* ims = cm.img_ms + f * 6;
    const float* rms = cm.ref_ms + (cm.ref_frames == 1 ? 0 : (f % cm.ref_frames)) * 6;
    float o[3];
    const SigmaRecip SR = sigma_recip(ims);
    colormatch_pixel(x, ims, rms, cm.K, cm.T, o, PT, &SR);
    store_px_stream(out + at, px3{o[0], o[1], o[2]});
}

// Frames below 2 GiB: four pixels per thread, as k_chain_pointwise4 (vrg_chain.hip) -- the four requests ahead of the table staging,
// the staging and its barrier once per 1024 pixels, no exec-masked block (the tail's stores are dropped by the buffer range check).
template <bool FAST>
__global__ __launch_bounds__(256) void k_colormatch_apply4(const px3* __restrict__ in, px3* __restrict__ out, int32_t pixels_per_frame, CmK cm,
                                                            DevMath dm) {
    const int64_t f = blockIdx.y;
    const px3* fin = in + f * pixels_per_frame;
    const int32_t p0 = (int32_t)blockIdx.x * 1024 + (int32_t)threadIdx.x;
    px3 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int32_t p = p0 + 256 * j;
        v[j] = load_px_stream(fin + (p < pixels_per_frame ? p : 0));
    }
    VRG_CM_MATH(PT, true, FAST, dm);
    typedef unsigned u3 __attribute__((ext_vector_type(3)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(out + f * pixels_per_frame), 0, pixels_per_frame * 12, 0x00020000);
    const float* ims = cm.img_ms + f * 6;
    const float* rms = cm.ref_ms + (cm.ref_frames == 1 ? 0 : (f % cm.ref_frames)) * 6;
    const SigmaRecip SR = sigma_recip(ims);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int32_t p = p0 + 256 * j;
        const float x[3] = {v[j].r, v[j].g, v[j].b};
        float o[3];
        colormatch_pixel(x, ims, rms, cm.K, cm.T, o, PT, &SR);
        __builtin_amdgcn_raw_buffer_store_b96(u3{__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2])}, rs,
                                              (int)(p < pixels_per_frame ? (uint32_t)p * 12u : 0x80000000u), 0, 0);
    }
}

}  // namespace vrg

using namespace vrg;

extern "C" {

int vrg_noise_f32(float* out, int64_t frames, int64_t frame_elems, const vrg_noise_desc* nd, void* stream) {
    if (!out || !nd || frames < 0 || frame_elems <= 0 || nd->chunk_frames < 1 || nd->grid_threads == 0 ||
        (nd->grid_threads % 256u) != 0)
        return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    if (frames % nd->chunk_frames != 0) return VRG_ERR_BAD_ARG;
    const int64_t chunks = frames / nd->chunk_frames;
    const int64_t numel = (int64_t)nd->chunk_frames * frame_elems;
    const NoiseK nk = make_noise(nd, frame_elems);
    const uint64_t groups = (uint64_t)((numel + 4 * (int64_t)nk.G - 1) / (4 * (int64_t)nk.G));
    const uint64_t blocks = (uint64_t)chunks * groups * ((nk.G + 255u) / 256u);
    if (blocks > 0x7fffffffull) return VRG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_noise, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, out, nk, numel, (uint32_t)groups);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

static int launch_grain_any(const void* in, void* out, int64_t frames, int32_t height, int32_t width, float intensity, float sat,
                            float one_minus_sat, const vrg_noise_desc* nd, bool u8, void* stream) {
    if (!in || !out || !nd || frames < 0 || height <= 0 || width <= 0 || nd->chunk_frames < 1 || nd->grid_threads == 0 ||
        (nd->grid_threads % 256u) != 0)
        return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    if (frames % nd->chunk_frames != 0) return VRG_ERR_BAD_ARG;
    const int64_t frame_elems = (int64_t)height * width * 3;
    const int64_t chunks = frames / nd->chunk_frames;
    const int64_t numel = (int64_t)nd->chunk_frames * frame_elems;
    if (numel > 0x7fffffffll) return VRG_ERR_UNSUPPORTED;   // torch itself splits randn above INT32 indexing
    const NoiseK nk = make_noise(nd, frame_elems);
    const uint64_t groups = (uint64_t)((numel + 4 * (int64_t)nk.G - 1) / (4 * (int64_t)nk.G));
    const uint64_t blocks = (uint64_t)chunks * groups * ((nk.G + GRAIN_N - 1) / GRAIN_N);
    if (blocks > 0x7fffffffull) return VRG_ERR_UNSUPPORTED;
    // float4 path needs every chunk base and G*ii offsets 16-byte aligned
    const bool vec = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16 == 0) && (numel % 4 == 0);
    if (u8)
        hipLaunchKernelGGL((k_grain<false, true>), dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, in, out, nk, numel,
                           (uint32_t)groups, intensity, sat, one_minus_sat);
    else if (vec)
        hipLaunchKernelGGL((k_grain<true, false>), dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, in, out, nk, numel,
                           (uint32_t)groups, intensity, sat, one_minus_sat);
    else
        hipLaunchKernelGGL((k_grain<false, false>), dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, in, out, nk, numel,
                           (uint32_t)groups, intensity, sat, one_minus_sat);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

int vrg_grain_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width, float intensity, float sat,
                  float one_minus_sat, const vrg_noise_desc* nd, void* stream) {
    return launch_grain_any(in, out, frames, height, width, intensity, sat, one_minus_sat, nd, false, stream);
}

int vrg_sharpen_grain_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width, float strength, int32_t border,
                          float intensity, float sat, float one_minus_sat, const vrg_noise_desc* nd, void* stream) {
    if (!in || !out || in == out || !nd || frames < 0 || height <= 0 || width <= 0 || border < 0 || border > 1 || nd->chunk_frames < 1 ||
        nd->grid_threads == 0 || (nd->grid_threads % 256u) != 0)
        return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    const int64_t frame_elems = (int64_t)height * width * 3;
    const bool aligned = (reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
    if (nd->chunk_frames != 1 || width % 4 != 0 || (int64_t)width * 3 / 4 < 256 || frame_elems > 0x7fffffffll || !aligned)
        return VRG_ERR_UNSUPPORTED;                                   // the caller runs vrg_stencil3x3_f32 + vrg_grain_f32
    const NoiseK nk = make_noise(nd, frame_elems);
    const uint64_t groups = (uint64_t)((frame_elems + 4 * (int64_t)nk.G - 1) / (4 * (int64_t)nk.G));
    const uint64_t per_frame = groups * ((nk.G + GRAIN_N - 1) / GRAIN_N);
    const int64_t step = (int64_t)(0x7fffffffull / per_frame) - 1;   // frames per launch
    if (step < 1) return VRG_ERR_UNSUPPORTED;
    for (int64_t f0 = 0; f0 < frames; f0 += step) {
        const int64_t nf = frames - f0 < step ? frames - f0 : step;
        NoiseK nkk = nk;
        nkk.chunk0 += f0;
        const uint32_t total = (uint32_t)(per_frame * (uint64_t)nf);
        const uint32_t blocks = ((total + 7u) / 8u) * 8u;
        const float* src = in + f0 * frame_elems;
        float* dst = out + f0 * frame_elems;
        if (border == VRG_BORDER_ZERO)
            hipLaunchKernelGGL((k_sharpen_grain<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, nkk, height, width * 3 / 4,
                               (uint32_t)groups, total, strength, intensity, sat, one_minus_sat);
        else
            hipLaunchKernelGGL((k_sharpen_grain<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, nkk, height, width * 3 / 4,
                               (uint32_t)groups, total, strength, intensity, sat, one_minus_sat);
        VRG_CHECK_LAUNCH();
    }
    return VRG_OK;
}

int vrg_sharpen_grain_u8(const uint8_t* in, uint8_t* out, int64_t frames, int32_t height, int32_t width, float strength, int32_t border,
                         float intensity, float sat, float one_minus_sat, const vrg_noise_desc* nd, void* stream) {
    if (!in || !out || in == out || !nd || frames < 0 || height <= 0 || width <= 0 || border < 0 || border > 1 || nd->chunk_frames < 1 ||
        nd->grid_threads == 0 || (nd->grid_threads % 256u) != 0)
        return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    const int64_t frame_elems = (int64_t)height * width * 3;
    const bool aligned = (reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 4 == 0;
    // several frames per noise chunk: the enhancer seeds every frame on its own (reference :233-276), nothing calls that; a batch of less than
    // four bytes (one 1 x 1 frame) has no dword to load: the caller converts and runs the fp32 entry points
    if (nd->chunk_frames != 1 || frame_elems > 0x7fffffffll || frames * frame_elems < 4) return VRG_ERR_UNSUPPORTED;
    const bool fast = width % 4 == 0 && (int64_t)width * 3 / 4 >= 256 && aligned;      // rows on the frame's dword grid, at most one row end per block
    const NoiseK nk = make_noise(nd, frame_elems);
    const uint64_t groups = (uint64_t)((frame_elems + 4 * (int64_t)nk.G - 1) / (4 * (int64_t)nk.G));
    const uint64_t per_frame = groups * ((nk.G + GRAIN_N - 1) / GRAIN_N);
    const int64_t step = (int64_t)(0x7fffffffull / per_frame) - 1;   // frames per launch
    if (step < 1) return VRG_ERR_UNSUPPORTED;
    for (int64_t f0 = 0; f0 < frames; f0 += step) {
        const int64_t nf = frames - f0 < step ? frames - f0 : step;
        NoiseK nkk = nk;
        nkk.chunk0 += f0;
        const uint32_t total = (uint32_t)(per_frame * (uint64_t)nf);
        const uint32_t blocks = ((total + 7u) / 8u) * 8u;
        if (!fast) {
            // any width, any alignment: flat byte space of the batch (k_sharpen_grain_u8_any); the windows may reach into the neighbouring
            // frames of the WHOLE batch, so the kernel gets the batch's base and the launch's first frame through the chunk index
            NoiseK nka = nk;
            nka.chunk0 += f0;
            const uint8_t* base = in + f0 * frame_elems;
            uint8_t* obase = out + f0 * frame_elems;
            const int64_t reach = (frames - f0) * frame_elems;       // bytes addressable from `base` upwards (windows below it are clamped)
            if (border == VRG_BORDER_ZERO)
                hipLaunchKernelGGL((k_sharpen_grain_u8_any<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, base, obase, nka, height, width * 3,
                                   (uint32_t)frame_elems, reach, (uint32_t)groups, total, strength, intensity, sat, one_minus_sat);
            else
                hipLaunchKernelGGL((k_sharpen_grain_u8_any<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, base, obase, nka, height, width * 3,
                                   (uint32_t)frame_elems, reach, (uint32_t)groups, total, strength, intensity, sat, one_minus_sat);
            VRG_CHECK_LAUNCH();
            continue;
        }
        const uint32_t* src = reinterpret_cast<const uint32_t*>(in + f0 * frame_elems);
        uint32_t* dst = reinterpret_cast<uint32_t*>(out + f0 * frame_elems);
        if (border == VRG_BORDER_ZERO)
            hipLaunchKernelGGL((k_sharpen_grain_u8<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, nkk, height, width * 3 / 4,
                               (uint32_t)groups, total, strength, intensity, sat, one_minus_sat);
        else
            hipLaunchKernelGGL((k_sharpen_grain_u8<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, nkk, height, width * 3 / 4,
                               (uint32_t)groups, total, strength, intensity, sat, one_minus_sat);
        VRG_CHECK_LAUNCH();
    }
    return VRG_OK;
}

// uint8 B,G,R frames, grain only: the shared-Philox kernel (one Philox call per four elements instead of the point-wise
// chain kernel's one per element).  Used by vrg_fused_chain_u8 for a chain that is just the grain stage.
}  // extern "C"
namespace vrg {
int launch_grain_u8(const uint8_t* in, uint8_t* out, int64_t frames, int32_t height, int32_t width, float intensity, float sat,
                    float one_minus_sat, const vrg_noise_desc* nd, void* stream) {
    return launch_grain_any(in, out, frames, height, width, intensity, sat, one_minus_sat, nd, true, stream);
}
}  // namespace vrg
extern "C" {

int vrg_grain_injected_f32(const float* in, const float* noise, float* out, int64_t pixels, float intensity, float sat,
                           float one_minus_sat, void* stream) {
    if (!in || !noise || !out || pixels < 0) return VRG_ERR_BAD_ARG;
    if (pixels == 0) return VRG_OK;
    const uint64_t blocks = (uint64_t)(pixels + 255) / 256;
    if (blocks > 0x7fffffffull) return VRG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_grain_injected, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const px3*>(in), reinterpret_cast<const px3*>(noise), reinterpret_cast<px3*>(out), pixels,
                       intensity, sat, one_minus_sat);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

int64_t vrg_lut_cells_floats(int32_t lut_size) {
    if (lut_size < 2 || lut_size > 256) return 0;
    const int64_t nc = lut_size - 1;
    return nc * nc * lut_size * LUT_REC_FLOATS;
}

int vrg_lut_prepare_f32(const float* lut, int32_t lut_size, float* cells, void* stream) {
    if (!lut || !cells || lut_size < 2 || lut_size > 256) return VRG_ERR_BAD_ARG;
    const int64_t nc = lut_size - 1;
    const uint32_t blocks = (uint32_t)((nc * nc * lut_size + 255) / 256);
    hipLaunchKernelGGL(k_lut_build_cells, dim3(blocks), dim3(256), 0, (hipStream_t)stream, lut, lut_size, cells);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

int vrg_lut3d_f32(const float* in, float* out, int64_t pixels, int32_t channels, const float* lut, int32_t lut_size,
                  const float domain_min[3], const float domain_max[3], int32_t blend_mode, float blend, float one_minus_blend,
                  void* stream) {
    if (!in || !out || !lut || !domain_min || !domain_max || pixels < 0 || channels < 3 || lut_size < 2 || lut_size > 256 ||
        (blend_mode != 1 && blend_mode != 2))
        return VRG_ERR_BAD_ARG;
    if (pixels == 0) return VRG_OK;
    const LutParams P = make_lut(lut, lut_size, domain_min, domain_max, blend_mode, blend, one_minus_blend);
    const uint64_t blocks = (uint64_t)(pixels + 255) / 256;
    if (blocks > 0x7fffffffull) return VRG_ERR_UNSUPPORTED;
    if (channels == 3 && lut_lds_applicable(lut_size, pixels)) {
        return launch_lut_lds(in, out, pixels, P, false, (hipStream_t)stream);      // LDS-resident table
    } else if (channels == 3)
        hipLaunchKernelGGL(k_lut3d<true>, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, in, out, pixels, channels, P);
    else
        hipLaunchKernelGGL(k_lut3d<false>, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, in, out, pixels, channels, P);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

int vrg_colormatch_apply_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width, const float* img_ms,
                             const float* ref_ms, int32_t ref_frames, float k, float one_minus_k, int32_t cm_math, void* stream) {
    if (!in || !out || !img_ms || !ref_ms || frames < 0 || height <= 0 || width <= 0 || ref_frames < 1 ||
        (cm_math != VRG_CM_MATH_DEVICE && cm_math != VRG_CM_MATH_FAST))
        return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    const int64_t ppf = (int64_t)height * width;
    if (ppf > 0x7fffffff) return VRG_ERR_UNSUPPORTED;
    CmK cm{img_ms, ref_ms, ref_frames, k, one_minus_k};
    const uint32_t bx = (uint32_t)((ppf + 255) / 256);
    for (int64_t f0 = 0; f0 < frames; f0 += 32768) {
        const int64_t nf = frames - f0 < 32768 ? frames - f0 : 32768;
        CmK c = cm;
        c.img_ms = img_ms + f0 * 6;
        // ref frame index uses f % ref_frames relative to the call start; 32768 is a multiple of any
        // chunk size only if ref_frames divides it -- keep the mapping exact by offsetting explicitly.
        if (ref_frames != 1 && (f0 % ref_frames) != 0) return VRG_ERR_UNSUPPORTED;
        if (ppf * 12 < ((int64_t)1 << 31)) {
            const dim3 g4((uint32_t)((ppf + 1023) / 1024), (uint32_t)nf);
            if (cm_math == VRG_CM_MATH_FAST)
                hipLaunchKernelGGL(k_colormatch_apply4<true>, g4, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const px3*>(in) + f0 * ppf,
                                   reinterpret_cast<px3*>(out) + f0 * ppf, (int32_t)ppf, c, host_dev_math());
            else
                hipLaunchKernelGGL(k_colormatch_apply4<false>, g4, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const px3*>(in) + f0 * ppf,
                                   reinterpret_cast<px3*>(out) + f0 * ppf, (int32_t)ppf, c, host_dev_math());
        } else if (cm_math == VRG_CM_MATH_FAST)
            hipLaunchKernelGGL(k_colormatch_apply<true>, dim3(bx, (uint32_t)nf), dim3(256), 0, (hipStream_t)stream,
                               reinterpret_cast<const px3*>(in) + f0 * ppf, reinterpret_cast<px3*>(out) + f0 * ppf, (int32_t)ppf, c, host_dev_math());
        else
            hipLaunchKernelGGL(k_colormatch_apply<false>, dim3(bx, (uint32_t)nf), dim3(256), 0, (hipStream_t)stream,
                               reinterpret_cast<const px3*>(in) + f0 * ppf, reinterpret_cast<px3*>(out) + f0 * ppf, (int32_t)ppf, c, host_dev_math());
        VRG_CHECK_LAUNCH();
    }
    return VRG_OK;
}

}  // extern "C"
