// vrg_march.hip -- the fused grain -> LUT -> 3x3 sharpen chain (any subset, no colour-match stage) as a
// register-resident "wave march": the kernel vrg_fused_chain_f32 picks for grain -> (LUT) -> sharpen.  gfx950 only.
//
// One wave64 owns a vertical strip 64 pixels wide and walks down it one frame row per step: every lane
// keeps the last three processed rows of its column in VGPRs, the 3x3 taps of the left/right columns come
// from the neighbouring lanes with DPP wave shifts, so the grain -> LUT result of a pixel is
// computed once (no LDS tile, no barrier, no halo rows recomputed per tile; 61 of 64 lanes produce
// output, 2 rows of ~49 are priming).
//
// Noise: torch.randn gives element li of a chunk the component (li/G)%4 of Philox call (li/G)/4 of
// subsequence li%G.  The four outputs of one Philox call therefore belong to four elements G apart --
// about 45 rows of a 4K frame.  A wave marches FOUR strips at once ("siblings" m = 0..3, G*m elements
// apart): each lane makes three Philox calls per step and uses all twelve normals, one per element of
// its four pixels.  G is not a multiple of 3, so sibling m's pixel grid is shifted by s_m elements
// against the lane's subsequences; the 1-2 normals that fall into the next lane's calls are fetched with
// a DPP shift, and lane 63 only provides noise.  Rows where a strip leaves its Philox quarter (the ragged
// first/last row of a quarter, or the priming rows) fall back to the general per-element routine.
// Everything is expressed in flat element space, so sibling strips that wrap around a row end or cross
// a frame boundary need no special cases: borders are decided per pixel from its true (x, y).
//
// Without a grain stage the same kernel runs with a synthetic G = 48 rows (siblings = four row bands).
#include "vrg_chain_stages.hpp"

namespace vrg {

struct MarchK {
    int32_t H, W, E;        // E = 3*W
    int32_t chunk_frames;
    int32_t rows_chunk;     // chunk_frames * H
    int32_t numel;          // chunk elements (< 2^31)
    uint32_t G;             // Philox subsequences per randn call (or the synthetic band size)
    uint32_t K, T;          // super-groups (4G elements) per chunk, column tiles per row
    uint32_t chunks;
    int32_t s[4];           // s_m = (3 - (G*m)%3)%3 : element shift that re-aligns sibling m to pixels
    int32_t delta[4];       // (G*m + s_m) / 3 : sibling m's pixel offset
    int64_t elems_before;   // addressable elements of the caller's buffer before chunk 0 / after the last chunk
    int64_t elems_after;
};

__device__ __forceinline__ float lane_prev(float v) {   // value held by lane-1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v) {   // value held by lane+1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

__device__ __forceinline__ int64_t floor_div64(int64_t a, int64_t b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
__device__ __forceinline__ int floor_div32(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// lane_prev/lane_next self-test: out[lane] = lane_prev(lane), out[64+lane] = lane_next(lane)
__global__ void k_selftest_lanes(float* out) {
    const float v = (float)threadIdx.x;
    out[threadIdx.x] = lane_prev(v);
    out[64 + threadIdx.x] = lane_next(v);
}

// WAVES = 4: one job per wave, nothing shared.  WAVES = 12 (LUT stage with a cube of at most 21^3): the workgroup first
// stages the cube's node table in LDS (dynamic shared memory, one float4 per node) and the gathers of the LUT stage
// become ds_read_b128 -- the L1 / L2 gather path that holds the global-table form at ~65 Gpix/s is not used at all.
template <int STAGES, bool SHARPEN, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES) void k_chain_march(const float* __restrict__ in, float* __restrict__ out, MarchK M, ChainK D) {
    static_assert(!(STAGES & VRG_STAGE_COLORMATCH), "colour-match chains run on the tile / point-wise kernels");
    extern __shared__ __attribute__((aligned(16))) float march_lut_nodes[];
    const f32x4* lut_nodes = nullptr;
    if (WAVES != 4) {
        lut_nodes_to_lds(D.lut, reinterpret_cast<f32x4*>(march_lut_nodes), (int)threadIdx.x, 64 * WAVES);
        lut_nodes = reinterpret_cast<const f32x4*>(march_lut_nodes);
    }
    if (WAVES != 4) __syncthreads();

    constexpr int CLO = SHARPEN ? 1 : 0;     // first lane that produces output
    constexpr int CW = SHARPEN ? 61 : 63;    // output lanes per wave (lane 63 only provides noise)
    const int lane = threadIdx.x & 63;
    const uint32_t job = __builtin_amdgcn_readfirstlane(blockIdx.x * (uint32_t)WAVES + (threadIdx.x >> 6));
    const uint32_t jobs_per_chunk = M.K * M.T;
    const uint32_t chunk = job / jobs_per_chunk;
    if (chunk >= M.chunks) return;
    const uint32_t rem = job - chunk * jobs_per_chunk;
    const uint32_t k = rem / M.T;
    const uint32_t t = rem - k * M.T;
    const int H = M.H, W = M.W, E = M.E;
    const int x0 = (int)t * CW - CLO;
    const int xp = x0 + lane;
    const bool lane_out = (lane >= CLO) && (lane < CLO + CW) && (xp >= 0) && (xp < W);
    const int64_t q0 = (int64_t)4 * M.G * k;                       // first element of Philox quarter 0 of this group
    const int r_lo = (int)floor_div64(q0 - 4, E);
    int r_hi = (int)floor_div64(q0 + (int64_t)M.G - 1, E);
    if (r_hi > M.rows_chunk - 1) r_hi = M.rows_chunk - 1;
    const int r_first = r_lo - (SHARPEN ? 1 : 0);
    const int r_last = r_hi + (SHARPEN ? 1 : 0);

    // All element arithmetic below is 32-bit and relative to the chunk base (chunk elements < 2^31 - margin).
    const float* cin = in + (int64_t)chunk * M.numel;
    float* cout = out + (int64_t)chunk * M.numel;
    const uint64_t seed = chunk_seed(D.noise, chunk);
    const uint64_t off = chunk_offset(D.noise, chunk);
    const uint64_t ctr = (off >> 2) + k;
    const bool zero = D.zero_border != 0;
    const uint32_t G = M.G;
    const uint32_t px_limit = (uint32_t)(M.numel - 2);              // li is a whole pixel of the chunk iff (u32)li < px_limit
    // loads are issued for every lane; addresses are clamped to memory that exists (the neighbouring chunks of
    // this launch are addressable, the outside of the caller's buffer is not)
    const int64_t before = (int64_t)chunk * M.numel + M.elems_before;
    const int64_t after = (int64_t)(M.chunks - 1 - chunk) * M.numel + M.elems_after;
    const int li_min = -(int)(before < 0x30000000ll ? before : 0x30000000ll);
    const int li_max = M.numel - 3 + (int)(after < 0x08000000ll ? after : 0x08000000ll);

    // true coordinates of the pixel this lane computes for sibling m at the current step
    int xm[4], yc[4], fc[4], offm[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int t0 = xp + M.delta[m];
        const int a = floor_div32(t0, W);
        xm[m] = t0 - a * W;
        const int rho_m = r_first + a;
        fc[m] = floor_div32(rho_m, H);
        yc[m] = rho_m - fc[m] * H;
        offm[m] = (int)(G * (uint32_t)m) + M.s[m] + 3 * lane;      // element offset of the lane's sibling-m pixel in a row step
    }
    float U[4][3], Mi[4][3];
    int yM[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        yM[m] = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) { U[m][c] = 0.0f; Mi[m][c] = 0.0f; }
    }

    // Straight-line schedule: the four siblings are independent, so all four input loads are issued together
    // and the loads of the NEXT row are issued before the current row is processed -- the LUT gathers and the
    // HBM stream then overlap across siblings and steps.
    int rowbase = r_first * E + 3 * x0;                                // element of lane 0's primary pixel
    const int q0s = (int)q0;
    px3 xin[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int li = rowbase + offm[m];
        li = li < li_min ? li_min : (li > li_max ? li_max : li);
        xin[m] = *reinterpret_cast<const px3*>(cin + li);
    }

    for (int rho = r_first; rho <= r_last; ++rho, rowbase += E) {
        px3 xnext[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            int li = rowbase + E + offm[m];
            li = li < li_min ? li_min : (li > li_max ? li_max : li);
            xnext[m] = *reinterpret_cast<const px3*>(cin + li);
        }
        // ---------------- noise: three Philox calls per lane feed all four siblings
        float nz[3][4], nx[2][4];
        bool fast = false;
        const int b0 = rowbase - q0s;
        if (STAGES & VRG_STAGE_GRAIN) {
            fast = (b0 >= 0) && ((uint32_t)(b0 + 3 * 63 + 4) < G);
            if (fast) {
                const uint32_t idx0 = (uint32_t)b0 + 3u * (uint32_t)lane;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const u32x4 r = philox_for(seed, idx0 + j, ctr);
                    const f32x2 a = box_muller(r.x, r.y);
                    const f32x2 b = box_muller(r.z, r.w);
                    nz[j][0] = a.x; nz[j][1] = a.y; nz[j][2] = b.x; nz[j][3] = b.y;
                }
#pragma unroll
                for (int m = 1; m < 4; ++m) {
                    nx[0][m] = lane_next(nz[0][m]);
                    nx[1][m] = lane_next(nz[1][m]);
                }
            }
        }
        // ---------------- per sibling: the pixel's normals, grain
        const int b0o = b0 - E;                                            // middle row, relative to the quarter (sharpen output)
        float V[4][3];                                                     // the pixel after the grain stage
        bool valid[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int li = rowbase + offm[m];
            valid[m] = (uint32_t)li < px_limit;                            // whole pixel inside the chunk
            const float x[3] = {xin[m].r, xin[m].g, xin[m].b};
            float n[3] = {0.0f, 0.0f, 0.0f};
            if (STAGES & VRG_STAGE_GRAIN) {
                if (fast) {
                    if (m == 0 || M.s[m] == 0) {
                        n[0] = nz[0][m]; n[1] = nz[1][m]; n[2] = nz[2][m];
                    } else if (M.s[m] == 1) {
                        n[0] = nz[1][m]; n[1] = nz[2][m]; n[2] = nx[0][m];
                    } else {
                        n[0] = nz[2][m]; n[1] = nx[0][m]; n[2] = nx[1][m];
                    }
                } else if (valid[m]) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) n[c] = torch_randn_element(seed, off, G, (uint64_t)(uint32_t)(li + c));
                }
                grain_pixel(x, n, D.I, D.S, D.T, V[m]);
            } else {
                V[m][0] = x[0]; V[m][1] = x[1]; V[m][2] = x[2];
            }
            xin[m] = xnext[m];
        }
        // ---------------- the rest of the pre stages and the output of sibling m
        float Dn[4][3];
        auto set_row = [&](int m, const float o[3]) {
            Dn[m][0] = valid[m] ? o[0] : 0.0f; Dn[m][1] = valid[m] ? o[1] : 0.0f; Dn[m][2] = valid[m] ? o[2] : 0.0f;
        };
        auto emit = [&](int m) {
            if (SHARPEN) {
                if (rho >= r_first + 2) {
                    const bool top = yM[m] == 0, bottom = yM[m] == H - 1;
                    const bool left = xm[m] == 0, right = xm[m] == W - 1;
                    const bool any_edge = __builtin_amdgcn_ballot_w64(top || bottom || left || right) != 0;   // wave-uniform
                    float res[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float p[3][3];
                        p[0][0] = lane_prev(U[m][c]);  p[0][1] = U[m][c];  p[0][2] = lane_next(U[m][c]);
                        p[1][0] = lane_prev(Mi[m][c]); p[1][1] = Mi[m][c]; p[1][2] = lane_next(Mi[m][c]);
                        p[2][0] = lane_prev(Dn[m][c]); p[2][1] = Dn[m][c]; p[2][2] = lane_next(Dn[m][c]);
                        if (any_edge) {                                        // rare: some lane of the wave sits on a frame border
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                if (top) p[0][j] = zero ? 0.0f : p[1][j];
                                if (bottom) p[2][j] = zero ? 0.0f : p[1][j];
                            }
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                if (left) p[i][0] = zero ? 0.0f : p[i][1];
                                if (right) p[i][2] = zero ? 0.0f : p[i][1];
                            }
                        }
                        res[c] = stencil_value(D.stencil_op, p, D.strength, D.zero_border);
                    }
                    const int li = rowbase - E + offm[m];
                    const uint32_t idx = (uint32_t)(b0o + M.s[m] + 3 * lane);   // subsequence of channel 0 (wraps to huge if negative)
                    const bool act = lane_out && (uint32_t)li < px_limit;
                    const bool a0 = idx < G, a1 = idx + 1u < G, a2 = idx + 2u < G;
                    // wave-uniform split: in all but the rows that touch the end of the Philox quarter every active lane
                    // stores a whole pixel -> one global_store_dwordx3 (the merged form costs a dword + a dwordx2 per pixel)
                    if (__builtin_amdgcn_ballot_w64(act && !(a0 && a1 && a2)) == 0) {
                        if (act) *reinterpret_cast<px3*>(cout + li) = px3{res[0], res[1], res[2]};
                    } else if (act) {
                        if (a0) cout[li] = res[0];
                        if (a1) cout[li + 1] = res[1];
                        if (a2) cout[li + 2] = res[2];
                    }
                }
                yM[m] = yc[m];
#pragma unroll
                for (int c = 0; c < 3; ++c) { U[m][c] = Mi[m][c]; Mi[m][c] = Dn[m][c]; }
            } else {
                const int li = rowbase + offm[m];
                const uint32_t idx = (uint32_t)(b0 + M.s[m] + 3 * lane);
                const bool act = lane_out && (uint32_t)li < px_limit;
                const bool a0 = idx < G, a1 = idx + 1u < G, a2 = idx + 2u < G;
                if (__builtin_amdgcn_ballot_w64(act && !(a0 && a1 && a2)) == 0) {
                    if (act) *reinterpret_cast<px3*>(cout + li) = px3{Dn[m][0], Dn[m][1], Dn[m][2]};
                } else if (act) {
                    if (a0) cout[li] = Dn[m][0];
                    if (a1) cout[li + 1] = Dn[m][1];
                    if (a2) cout[li + 2] = Dn[m][2];
                }
            }
        };
        {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float o[3] = {V[m][0], V[m][1], V[m][2]};
                if (STAGES & VRG_STAGE_LUT) {
                    float g[3];
                    if (lut_nodes) lut_pixel_nodes(D.lut, lut_nodes, V[m], g);      // small cube staged in LDS by the workgroup
                    else lut_pixel(D.lut, V[m], g);
                    o[0] = g[0]; o[1] = g[1]; o[2] = g[2];
                }
                set_row(m, o);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) emit(m);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (++yc[m] == H) { yc[m] = 0; ++fc[m]; }
        }
    }
}

template <int STAGES, bool SHARPEN>
static int launch_march_t(const float* in, float* out, const MarchK& M, const ChainK& D, hipStream_t st) {
    const uint64_t jobs = (uint64_t)M.chunks * M.K * M.T;
    const size_t lut_bytes = (STAGES & VRG_STAGE_LUT) ? (size_t)D.lut.n * D.lut.n * D.lut.n * 16 : 0;
    if ((STAGES & VRG_STAGE_LUT) && lut_bytes <= 152 * 1024 && jobs >= 1536) {
        // small cube: node table in LDS, 12-wave workgroups (one per CU next to the table)
        constexpr int WV = 12;
        const uint64_t blocks = (jobs + WV - 1) / WV;
        if (blocks >= (1ull << 22)) return VRG_ERR_UNSUPPORTED;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_march<STAGES, SHARPEN, WV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lut_bytes) != hipSuccess)
            return VRG_ERR_LAUNCH;
        hipLaunchKernelGGL((k_chain_march<STAGES, SHARPEN, WV>), dim3((uint32_t)blocks), dim3(64 * WV), lut_bytes, st, in, out, M, D);
        return hipGetLastError() == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH;
    }
    const uint64_t blocks = (jobs + 3) / 4;
    if (blocks >= (1ull << 24)) return VRG_ERR_UNSUPPORTED;      // work-items per launch are counted in 32 bits
    hipLaunchKernelGGL((k_chain_march<STAGES, SHARPEN>), dim3((uint32_t)blocks), dim3(256), 0, st, in, out, M, D);
    return hipGetLastError() == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH;
}

template <int STAGES>
static int launch_march_s(const float* in, float* out, const MarchK& M, const ChainK& D, bool sharpen, hipStream_t st) {
    return sharpen ? launch_march_t<STAGES, true>(in, out, M, D, st) : launch_march_t<STAGES, false>(in, out, M, D, st);
}

// One launch per run of equal chunks.  With a grain stage the chunk is the RNG chunk; without, frames are
// grouped so that a chunk stays below 2^31 elements and G is a synthetic 48-row band.
int launch_march(const float* in, float* out, int64_t frames, int32_t H, int32_t W, const ChainK& D0, int stages, hipStream_t st) {
    if (stages & (VRG_STAGE_COLORMATCH | VRG_STAGE_FROM_LAB)) return VRG_ERR_UNSUPPORTED;
    const bool sharpen = (stages & VRG_STAGE_SHARPEN) != 0;
    const bool grain = (stages & VRG_STAGE_GRAIN) != 0;
    const int64_t fe = (int64_t)H * W * 3;
    int64_t cf;
    uint32_t G;
    if (grain) {
        cf = D0.noise.chunk_frames;
        G = D0.noise.G;
        if (frames % cf) return VRG_ERR_BAD_ARG;
    } else {
        cf = 0x60000000ll / fe;
        if (cf < 1) return VRG_ERR_UNSUPPORTED;
        if (cf > frames) cf = frames;
        const int64_t band = 48ll * W * 3;
        G = (uint32_t)(band < 0x08000000ll ? band : 0x08000000ll);
    }
    if (cf * fe > 0x60000000ll) return VRG_ERR_UNSUPPORTED;
    constexpr int CWs = 61, CWp = 63;
    int64_t done = 0;
    while (done < frames) {
        int64_t cfr = cf, nchunks = (frames - done) / cf;
        if (nchunks == 0) { cfr = frames - done; nchunks = 1; }      // ragged tail (only without grain)
        MarchK M;
        M.H = H; M.W = W; M.E = 3 * W; M.chunk_frames = (int32_t)cfr; M.rows_chunk = (int32_t)(cfr * H);
        M.numel = (int32_t)(cfr * fe); M.G = G;
        M.K = (uint32_t)((cfr * fe + 4ll * G - 1) / (4ll * G));
        M.T = (uint32_t)((W + (sharpen ? CWs : CWp) - 1) / (sharpen ? CWs : CWp));
        M.chunks = (uint32_t)nchunks;
        for (int m = 0; m < 4; ++m) {
            const int64_t gm = (int64_t)G * m;
            M.s[m] = (int32_t)((3 - gm % 3) % 3);
            M.delta[m] = (int32_t)((gm + M.s[m]) / 3);
        }
        M.elems_before = done * fe;
        M.elems_after = (frames - done - nchunks * cfr) * fe;
        ChainK D = D0;
        const int64_t chunk_index0 = done / cf;
        if (grain) D.noise.chunk0 += chunk_index0;
        const float* src = in + done * fe;
        float* dst = out + done * fe;
        int rc;
        switch (stages & 3) {
            case 0: rc = launch_march_s<0>(src, dst, M, D, sharpen, st); break;
            case 1: rc = launch_march_s<1>(src, dst, M, D, sharpen, st); break;
            case 2: rc = launch_march_s<2>(src, dst, M, D, sharpen, st); break;
            default: rc = launch_march_s<3>(src, dst, M, D, sharpen, st); break;
        }
        if (rc) return rc;
        done += nchunks * cfr;
    }
    return VRG_OK;
}

}  // namespace vrg

extern "C" int vrg_selftest_lanes(float* out128, void* stream) {
    if (!out128) return VRG_ERR_BAD_ARG;
    hipLaunchKernelGGL(vrg::k_selftest_lanes, dim3(1), dim3(64), 0, (hipStream_t)stream, out128);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}
