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

// The variants this kernel was chosen from (round 3's general row step for every row, per-lane gathers, loads / stores at the head of the
// row, rotated noise synthesis, timing ablations with wrong pixels, ...) are NOT compile-time switches of the product source: the
// round-4 source with all of them is tools/ab/r04/vrg_march.hip (build: tools/build_variant.py <name> --unit vrg_march.hip --source
// tools/ab/r04/vrg_march.hip -DVRG_MARCH_...=...), their measurements LABNOTES.md I.2 / I.9 / I.11.
#ifndef LAB_WGW
#define LAB_WGW 4
#endif
#ifndef LAB_EXTRA_LDS
#define LAB_EXTRA_LDS 0
#endif
#ifndef LAB_NOFENCE
#define LAB_NOFENCE 0
#endif
#ifndef LAB_PERSIST
#define LAB_PERSIST 0
#endif
#ifndef LAB_MINW
#define LAB_MINW 3
#endif
#ifndef LAB_SLOTS
#define LAB_SLOTS 2            // LDS landing slots per wave of the quad-cooperative gathers: 3 = three siblings' gathers in flight (with LAB_MINW=2: two workgroups per CU, 256 VGPRs)
#endif
#define LAB_SL(m) ((m) % LAB_SLOTS)
#ifndef LAB_ROTATE
#define LAB_ROTATE 0           // 1: the NEXT row's noise (three Philox calls, six Box-Muller pairs: ~240 VALU instructions) is computed right behind the gathers' issue, in front of the first wait
#endif
#ifndef LAB_BALLAST
#define LAB_BALLAST 0          // extra independent v_fma_f32 per row step (4 chains), a quarter behind each sibling: is the loop VALU-issue bound?
#endif
#if LAB_NOFENCE
#define LAB_FENCE() do {} while (0)
#else
#define LAB_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
namespace vrg {

constexpr int MARCH_MIN_WAVES = LAB_MINW;      // waves per SIMD the register allocation must leave room for (launch bound of the 4-wave form)

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

// the same shifts for operands of an add (steady rows): no `old` value and bound_ctrl, so that the backend can fold the shift into
// the add's first operand (v_add_f32_dpp) -- the lane at the wave's end reads 0.0, and it is a halo lane whose result is dropped
__device__ __forceinline__ float tap_prev(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float tap_next(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
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
__global__ __launch_bounds__(64 * (WAVES == 4 ? LAB_WGW : WAVES), WAVES == 4 ? MARCH_MIN_WAVES : 1) void k_chain_march(const float* __restrict__ in, float* __restrict__ out, MarchK M, ChainK D) {
    constexpr int WGW = WAVES == 4 ? LAB_WGW : WAVES;
    static_assert(!(STAGES & VRG_STAGE_COLORMATCH), "colour-match chains run on the tile / point-wise kernels");
    extern __shared__ __attribute__((aligned(16))) float march_lut_nodes[];
    const f32x4* lut_nodes = nullptr;
    if (WAVES != 4) {
        lut_nodes_to_lds(D.lut, reinterpret_cast<f32x4*>(march_lut_nodes), (int)threadIdx.x, 64 * WAVES);
        lut_nodes = reinterpret_cast<const f32x4*>(march_lut_nodes);
    }
    if (WAVES != 4) __syncthreads();
    // LDS landing zone of the quad-cooperative gathers (steady rows of grain -> LUT chains over a global table): per wave two slots of six
    // 1040-byte rounds
    constexpr bool QUADP = (STAGES & VRG_STAGE_GRAIN) && (STAGES & VRG_STAGE_LUT) && WAVES == 4;
    constexpr int Q_ROUND = 1040, Q_SLOT = 6 * Q_ROUND;
    __shared__ __attribute__((aligned(16))) char quad_slots[QUADP ? WGW * LAB_SLOTS * Q_SLOT : 16];

    constexpr int CLO = SHARPEN ? 1 : 0;     // first lane that produces output
    constexpr int CW = SHARPEN ? 61 : 63;    // output lanes per wave (lane 63 only provides noise)
    const int lane = threadIdx.x & 63;
    const uint32_t jobs_per_chunk = M.K * M.T;
#if LAB_PERSIST
    for (uint32_t job = __builtin_amdgcn_readfirstlane(blockIdx.x * (uint32_t)WGW + (threadIdx.x >> 6)); job < jobs_per_chunk * M.chunks; job += gridDim.x * (uint32_t)WGW) {
#else
    const uint32_t job = __builtin_amdgcn_readfirstlane(blockIdx.x * (uint32_t)WGW + (threadIdx.x >> 6));
    {
#endif
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

    auto general_row = [&](const int rho, const int rowbase) {
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
        // Rows that are not wholly inside the job's Philox quarter.  Element `rel` (relative to sibling m's quarter) below 0 lies in the
        // PREVIOUS quarter -- component m - 1 of the same call for m >= 1, component 3 of call k - 1 for m = 0, subsequence rel + G --, at or
        // above G in the NEXT one (component m + 1, or component 0 of call k + 1 for m = 3, subsequence rel - G): the same sharing as in
        // the quarter, so three calls per lane at the shifted subsequences and three at the neighbouring call index feed all four siblings
        // (a ragged row adds the quarter's own three) instead of twelve per-element calls, each behind a 64-bit division.  A row never
        // reaches both neighbours (G >> 193).  Pixels outside the chunk (k = 0 / the last call) get garbage that `valid` discards.
        bool shared_edge = false;
        float nzb[3][4];
        if (STAGES & VRG_STAGE_GRAIN) {
            fast = (b0 >= 0) && ((uint32_t)(b0 + 3 * 63 + 4) < G);
            const bool own = (b0 + 3 * 63 + 4 >= 0) && (b0 < (int)G);        // some element of the row lies in the quarter
            if (fast || own) {
                const uint32_t idx0 = (uint32_t)b0 + 3u * (uint32_t)lane;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const u32x4 r = philox_for(seed, idx0 + j, ctr);
                    const f32x2 a = box_muller(r.x, r.y);
                    const f32x2 b = box_muller(r.z, r.w);
                    nz[j][0] = a.x; nz[j][1] = a.y; nz[j][2] = b.x; nz[j][3] = b.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) { nz[j][0] = 0.0f; nz[j][1] = 0.0f; nz[j][2] = 0.0f; nz[j][3] = 0.0f; }
            }
            if (fast) {
#pragma unroll
                for (int m = 1; m < 4; ++m) {
                    nx[0][m] = lane_next(nz[0][m]);
                    nx[1][m] = lane_next(nz[1][m]);
                }
            } else {
                shared_edge = true;
                const bool prev = b0 < 0;                                       // wave-uniform: the row reaches into the previous quarter (else the next)
                const uint32_t idxb = (uint32_t)b0 + (prev ? G : 0u - G) + 3u * (uint32_t)lane;
                const uint64_t ctr_far = prev ? ctr - 1 : ctr + 1;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const u32x4 r = philox_for(seed, idxb + j, ctr);
                    const f32x2 a = box_muller(r.x, r.y);
                    const f32x2 b = box_muller(r.z, r.w);
                    const u32x4 rf = philox_for(seed, idxb + j, ctr_far);
                    const float far = normal_component(rf, prev ? 3 : 0);
                    nzb[j][0] = prev ? far : a.y;
                    nzb[j][1] = prev ? a.x : b.x;
                    nzb[j][2] = prev ? a.y : b.y;
                    nzb[j][3] = prev ? b.x : far;
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
                } else if (shared_edge) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        // s_m is wave-uniform but not a compile-time constant here: both alignments of the element, selected
                        float own_v[3], far_v[3];
#pragma unroll
                        for (int sft = 0; sft < 3; ++sft) {
                            const int p = sft + c, j = p % 3;
                            own_v[sft] = p >= 3 ? lane_next(nz[j][m]) : nz[j][m];
                            far_v[sft] = p >= 3 ? lane_next(nzb[j][m]) : nzb[j][m];
                        }
                        const int sm = M.s[m];
                        const float o_ = sm == 0 ? own_v[0] : (sm == 1 ? own_v[1] : own_v[2]);
                        const float f_ = sm == 0 ? far_v[0] : (sm == 1 ? far_v[1] : far_v[2]);
                        const bool inq = (uint32_t)(b0 + 3 * lane + sm + c) < G;
                        n[c] = inq ? o_ : f_;
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
    };

    // ------------------------------------------------------------------------------------------------------------------------
    // Round 4: the STEADY rows of a job -- every lane of all four siblings a whole pixel of the chunk, the row inside the job's
    // Philox quarter (three calls per lane feed all twelve elements), the 3x3 windows of the output row inside one frame and one
    // image row per sibling -- run a straight-line body with no validity / border / split-store decisions at all (the general
    // row step above spends ~110 of its ~418 VALU instructions per pixel on them, and its wave-uniform branches cut the row into
    // ~40 basic blocks that nothing can be scheduled across).  Which rows are steady is decided with scalar arithmetic only:
    //   * per job (loop invariant): s = {0, 1, 2, 0} (G mod 3 == 2: every full-grid randn call of a 256-CU device), no sibling
    //     strip wraps around a row end (its 64 columns are one image row: then y is wave-uniform per sibling and no output lane
    //     sits on the left / right border), unsharp with a finite strength, a unit-domain LUT without strength blend, a chunk
    //     below 2^29 elements (32-bit byte offsets; buffer descriptors);
    //   * per row: b0 - E >= 0 and b0 + 193 < G (noise and output ownership), row and its neighbours inside the chunk, the
    //     computed row's y in [2, H-1] for every sibling (output row in [1, H-2]).
    // Pixel loads / stores go through buffer descriptors of the chunk: per-lane byte offsets are loop invariant (VGPRs), the row
    // offset is an SGPR, lanes that produce no output store at an out-of-range offset (dropped by the range check): no address
    // arithmetic and no exec-masked block in the loop, exact memory counters.  Same device functions (philox_for, box_muller,
    // grain_pixel, lut_axis, lut_fetch_finish, unsharp arithmetic) in the same order: bit-identical to the general step.
    // ------------------------------------------------------------------------------------------------------------------------
    // (chains without a stencil run the same body minus the taps and the row history; the 12-wave form -- cube of at most 21^3 staged in
    // LDS -- runs it with eight ds_read_b128 per pixel in the place of the gathers)
    constexpr bool FASTP = (STAGES & VRG_STAGE_GRAIN) && (WAVES == 4 || (STAGES & VRG_STAGE_LUT));
    bool fast_wave = false;
    if (FASTP) {
        bool ok = M.s[0] == 0 && M.s[1] == 1 && M.s[2] == 2 && M.s[3] == 0 && M.numel < (1 << 29) && r_last - r_first >= 4;
        if (SHARPEN) ok = ok && D.stencil_op == 0 && __builtin_isfinite(D.strength);
        if (STAGES & VRG_STAGE_LUT) ok = ok && D.lut.unit_domain != 0 && D.lut.blend_mode == 1;
        if (SHARPEN) {                    // without a stencil nothing depends on (x, y): sibling strips may wrap around row ends
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int xlo = __builtin_amdgcn_readfirstlane(xm[m]);
                ok = ok && __builtin_amdgcn_ballot_w64(xm[m] != xlo + lane) == 0;
            }
        }
        fast_wave = ok;
    }
    int rho = r_first;
    while (rho <= r_last) {
        bool steady = false;
        int ysc[4] = {0, 0, 0, 0};
        if (FASTP && fast_wave) {
            const int b0 = rowbase - q0s;
            if (SHARPEN)
                steady = rho >= r_first + 2 && b0 - E >= 0 && (uint32_t)(b0 + 3 * 63 + 4) < G && rowbase - E >= 0 &&
                         (int64_t)rowbase + 3ll * (int64_t)G + 3 * 63 < (int64_t)M.numel - 2 &&
                         (int64_t)rowbase + E + 3ll * (int64_t)G + 3 * 63 <= (int64_t)li_max;
            else                          // the row itself inside the quarter and the chunk; the next row's loads addressable
                steady = b0 >= 0 && (uint32_t)(b0 + 3 * 63 + 4) < G && rowbase >= 0 &&
                         (int64_t)rowbase + 3ll * (int64_t)G + 3 * 63 < (int64_t)M.numel - 2 &&
                         (int64_t)rowbase + E + 3ll * (int64_t)G + 3 * 63 <= (int64_t)li_max;
            if (steady && SHARPEN) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    ysc[m] = __builtin_amdgcn_readfirstlane(yc[m]);
                    steady = steady && ysc[m] >= 2;                                 // (yc <= H - 1 always)
                }
            }
        }
        if (!steady) {
            general_row(rho, rowbase);
            ++rho;
            rowbase += E;
            continue;
        }
        if (FASTP) {
            typedef unsigned u3 __attribute__((ext_vector_type(3)));
            const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cin), 0, M.numel * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(cout, 0, M.numel * 4, 0x00020000);
            int ld_voff[4], st_voff[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                ld_voff[m] = offm[m] * 4;
                st_voff[m] = lane_out ? offm[m] * 4 : (int)0x80000000u;
            }
            const LutParams& P = D.lut;
            const int nc = P.n - 1;
            // the wave's landing slots: LDS byte address (for M0) and this lane's two read positions
            const int wv_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            char* const quad_my = quad_slots + (QUADP ? wv_in_wg * LAB_SLOTS * Q_SLOT : 0);
            const unsigned quad_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) char*)quad_my);
            const char* const quad_a0 = quad_my + (lane & 3) * Q_ROUND + (lane & ~3) * 16;
            const char* const quad_a1 = quad_my + (4 + ((lane & 3) >> 1)) * Q_ROUND + ((lane & ~3) + 2 * (lane & 1)) * 16;
            int rows_done = 0;
            bool more = true;
            // behind a LUT stage the stencil's inputs are finite values in [0, 1]: the mean's Inf / NaN pass-through and the final clamp's NaN
            // pass-through are dropped there (same bits for finite data)
            constexpr bool FINITE = (STAGES & VRG_STAGE_LUT) != 0;
            // one Philox call + its two Box-Muller pairs: the normals of elements idx0 + j of the four siblings
            auto noise_call = [&](uint32_t idx0, int j, float nzj[4]) {
                const u32x4 r = philox_for(seed, idx0 + (uint32_t)j, ctr);
                const f32x2 a = box_muller(r.x, r.y);
                const f32x2 b = box_muller(r.z, r.w);
                nzj[0] = a.x; nzj[1] = a.y; nzj[2] = b.x; nzj[3] = b.y;
            };
            float nz[3][4];
#if LAB_ROTATE
            {
                const uint32_t idx0 = (uint32_t)(rowbase - q0s) + 3u * (uint32_t)lane;
#pragma unroll
                for (int j = 0; j < 3; ++j) noise_call(idx0, j, nz[j]);
            }
#endif
            float bl0 = (float)lane;
            auto ballast = [&]() {
#pragma unroll
                for (int i = 0; i < LAB_BALLAST / 16; ++i)
                    asm volatile("v_fma_f32 %0, %1, %2, %1\n v_fma_f32 %0, %2, %1, %1\n v_fma_f32 %0, %1, %2, %2\n v_fma_f32 %0, %2, %1, %2"
                                 : "=v"(bl0) : "v"(D.S), "v"(D.T));
            };
            while (more) {
                u3 xraw[4];           // the next row's pixels, requested at the END of this row step (see there)
                // ---- noise: three Philox calls per lane, twelve normals
#if !LAB_ROTATE
                {
                    const uint32_t idx0 = (uint32_t)(rowbase - q0s) + 3u * (uint32_t)lane;
#pragma unroll
                    for (int j = 0; j < 3; ++j) noise_call(idx0, j, nz[j]);
                }
#else
                float nzn[3][4];
                auto next_noise = [&]() {
                    const uint32_t idx0 = (uint32_t)(rowbase + E - q0s) + 3u * (uint32_t)lane;
#pragma unroll
                    for (int j = 0; j < 3; ++j) noise_call(idx0, j, nzn[j]);
                };
#endif
                const float nrm[4][3] = {{nz[0][0], nz[1][0], nz[2][0]},
                                         {nz[1][1], nz[2][1], lane_next(nz[0][1])},
                                         {nz[2][2], lane_next(nz[0][2]), lane_next(nz[1][2])},
                                         {nz[0][3], nz[1][3], nz[2][3]}};
                // ---- grain, LUT axes + gathers
                float V[4][3];
                LutFetch F[LAB_SLOTS];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float x[3] = {xin[m].r, xin[m].g, xin[m].b};
                    grain_pixel(x, nrm[m], D.I, D.S, D.T, V[m]);
                }
                const int out_soff = (SHARPEN ? rowbase - E : rowbase) * 4;
                // Quad-cooperative LDS-DMA gather: a pixel's 96-byte record run is fetched by the FOUR lanes of its quad --
                // round p = 0..3: the quad's lanes read the first 64 bytes of the run of the quad's pixel p (one 64-byte segment per quad
                // and instruction: the texture unit looks up ONE tag for the four lanes where the per-lane form looks up four), rounds 4 / 5
                // the last 32 bytes of two pixels each -- and the LDS-DMA lays each round out lane-linear (lane * 16 bytes), so pixel 4q+j
                // finds its pieces 0..3 at [round j][lane 4q + c] and its pieces 4, 5 at [round 4 + j/2][lane 4q + 2 (j & 1) + c]: the
                // transposition costs no VALU and no VGPR, and 2.9 instead of 6.4 L1 accesses per pixel (profiles/r04_probe_gather_pmc.json).
                // The DMA instructions are inline assembly (the compiler neither counts them nor knows they write the LDS): issue and wait
                // statements carry a memory clobber, the waits are counted by hand -- between the issue of a sibling's six rounds and their
                // use only the next sibling's six rounds are issued (the row's other memory operations sit at its end).
                auto lut_issue_dma = [&](int m) {
                    F[LAB_SL(m)].R = lut_axis(V[m][0], 0.0f, 1.0f, 1, P.top);
                    F[LAB_SL(m)].G = lut_axis(V[m][1], 0.0f, 1.0f, 1, P.top);
                    F[LAB_SL(m)].B = lut_axis(V[m][2], 0.0f, 1.0f, 1, P.top);
                    const int cell = ((F[LAB_SL(m)].B.cell * nc + F[LAB_SL(m)].G.cell) * P.n + F[LAB_SL(m)].R.cell) * (LUT_REC_FLOATS * 4);
                    const int ql16 = (lane & 3) * 16, qh16 = 64 + (lane & 1) * 16;
                    const int v0 = __builtin_amdgcn_update_dpp(0, cell, 0x00, 0xf, 0xf, false) + ql16;   // quad_perm [0,0,0,0]
                    const int v1 = __builtin_amdgcn_update_dpp(0, cell, 0x55, 0xf, 0xf, false) + ql16;   // [1,1,1,1]
                    const int v2 = __builtin_amdgcn_update_dpp(0, cell, 0xAA, 0xf, 0xf, false) + ql16;   // [2,2,2,2]
                    const int v3 = __builtin_amdgcn_update_dpp(0, cell, 0xFF, 0xf, 0xf, false) + ql16;   // [3,3,3,3]
                    const int v4 = __builtin_amdgcn_update_dpp(0, cell, 0x50, 0xf, 0xf, false) + qh16;   // [0,0,1,1]
                    const int v5 = __builtin_amdgcn_update_dpp(0, cell, 0xFA, 0xf, 0xf, false) + qh16;   // [2,2,3,3]
                    const unsigned l0 = quad_lds + (unsigned)(LAB_SL(m) * Q_SLOT);
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\t"
                                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %7\n\t"
                                 "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7\n\t"
                                 "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %7\n\t"
                                 "s_mov_b32 m0, %11\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %7\n\t"
                                 "s_mov_b32 m0, %12\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %7\n\t"
                                 "s_mov_b32 m0, %13\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %7\n\t"
                                 "s_mov_b32 m0, %0"
                                 : "=&s"(keep)
                                 : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "s"(P.cells), "s"(l0), "s"(l0 + Q_ROUND), "s"(l0 + 2 * Q_ROUND),
                                   "s"(l0 + 3 * Q_ROUND), "s"(l0 + 4 * Q_ROUND), "s"(l0 + 5 * Q_ROUND)
                                 : "memory");
                };
                auto lut_read_dma = [&](int m) {          // the pixel's six pieces out of its slot (after the hand-counted wait)
                    const char* s0 = quad_a0 + LAB_SL(m) * Q_SLOT;
                    const char* s1 = quad_a1 + LAB_SL(m) * Q_SLOT;
                    F[LAB_SL(m)].lo[0] = *reinterpret_cast<const f32x4*>(s0);
                    F[LAB_SL(m)].lo[1] = *reinterpret_cast<const f32x4*>(s0 + 16);
                    F[LAB_SL(m)].lo[2] = *reinterpret_cast<const f32x4*>(s0 + 32);
                    F[LAB_SL(m)].hi[0] = *reinterpret_cast<const f32x4*>(s0 + 48);
                    F[LAB_SL(m)].hi[1] = *reinterpret_cast<const f32x4*>(s1);
                    F[LAB_SL(m)].hi[2] = *reinterpret_cast<const f32x4*>(s1 + 16);
                };
                auto lut_issue = [&](int m) {
                    if (QUADP) { lut_issue_dma(m); return; }
                    if ((STAGES & VRG_STAGE_LUT) && WAVES != 4) {       // node table in LDS: the eight corners of the cell, laid out as the record form has them
                        F[LAB_SL(m)].R = lut_axis(V[m][0], 0.0f, 1.0f, 1, P.top);
                        F[LAB_SL(m)].G = lut_axis(V[m][1], 0.0f, 1.0f, 1, P.top);
                        F[LAB_SL(m)].B = lut_axis(V[m][2], 0.0f, 1.0f, 1, P.top);
                        const int n = P.n, nn = n * n;
                        const f32x4* t = lut_nodes + ((F[LAB_SL(m)].B.cell * n + F[LAB_SL(m)].G.cell) * n + F[LAB_SL(m)].R.cell);
                        const f32x4 q000 = t[0], q001 = t[nn], q010 = t[n], q011 = t[nn + n];
                        const f32x4 q100 = t[1], q101 = t[nn + 1], q110 = t[n + 1], q111 = t[nn + n + 1];
                        F[LAB_SL(m)].lo[0] = f32x4{q000.x, q001.x, q010.x, q011.x};  F[LAB_SL(m)].hi[0] = f32x4{q100.x, q101.x, q110.x, q111.x};
                        F[LAB_SL(m)].lo[1] = f32x4{q000.y, q001.y, q010.y, q011.y};  F[LAB_SL(m)].hi[1] = f32x4{q100.y, q101.y, q110.y, q111.y};
                        F[LAB_SL(m)].lo[2] = f32x4{q000.z, q001.z, q010.z, q011.z};  F[LAB_SL(m)].hi[2] = f32x4{q100.z, q101.z, q110.z, q111.z};
                        return;
                    }
                    if (STAGES & VRG_STAGE_LUT) {
                        F[LAB_SL(m)].R = lut_axis(V[m][0], 0.0f, 1.0f, 1, P.top);
                        F[LAB_SL(m)].G = lut_axis(V[m][1], 0.0f, 1.0f, 1, P.top);
                        F[LAB_SL(m)].B = lut_axis(V[m][2], 0.0f, 1.0f, 1, P.top);
                        const uint32_t cell = (uint32_t)((F[LAB_SL(m)].B.cell * nc + F[LAB_SL(m)].G.cell) * P.n + F[LAB_SL(m)].R.cell) * (uint32_t)(LUT_REC_FLOATS * 4);
                        const f32x4* q = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(P.cells) + cell);
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            F[LAB_SL(m)].lo[ch] = q[ch];
                            F[LAB_SL(m)].hi[ch] = q[3 + ch];
                        }
                    }
                };
                u3 resq[4];
                auto finish_emit = [&](int m) {
                    float Dn[3] = {V[m][0], V[m][1], V[m][2]};
                    if (STAGES & VRG_STAGE_LUT) lut_fetch_finish(F[LAB_SL(m)], Dn);
                    float res[3];
#pragma unroll
                    for (int c = 0; c < (SHARPEN ? 0 : 3); ++c) res[c] = Dn[c];          // no stencil: the row as it is
#pragma unroll
                    for (int c = 0; c < (SHARPEN ? 3 : 0); ++c) {
                        // unsharp_value's raster-order sum with the left / right taps taken from the neighbouring lanes
                        float sum = tap_prev(U[m][c]) + U[m][c];
                        sum = tap_next(U[m][c]) + sum;
                        sum = tap_prev(Mi[m][c]) + sum;
                        sum = sum + Mi[m][c];
                        sum = tap_next(Mi[m][c]) + sum;
                        sum = tap_prev(Dn[c]) + sum;
                        sum = sum + Dn[c];
                        sum = tap_next(Dn[c]) + sum;
                        const float blur = FINITE ? VRG_DIVC(sum, 9.0f) : div9(sum);
                        const float xc = Mi[m][c];
                        const float dd = xc - blur;
                        const float ee = D.strength * dd;
                        res[c] = FINITE ? clamp01_finite(xc + ee) : clamp01(xc + ee);
                        U[m][c] = Mi[m][c];
                        Mi[m][c] = Dn[c];
                    }
                    resq[m] = u3{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2])};
                };
                // the gathers of sibling m + 1 are in flight while sibling m is interpolated, sharpened and stored.  The hand-counted waits
                // of the quad form rest on the row's other memory operations sitting at its end: nothing but the next sibling's six rounds
                // enters the (in-order) memory counter between a sibling's issue and its use.
#if LAB_SLOTS == 3
                // three siblings' gathers in flight: sibling m is read when at most the two later siblings' twelve rounds are outstanding
                lut_issue(0);
                lut_issue(1);
                lut_issue(2);
                LAB_FENCE();
#if LAB_ROTATE
                next_noise();
                LAB_FENCE();
#endif
                if (QUADP) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); lut_read_dma(0); }
                finish_emit(0);
                ballast();
                LAB_FENCE();
                lut_issue(3);
                LAB_FENCE();
                if (QUADP) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); lut_read_dma(1); }
                finish_emit(1);
                ballast();
                LAB_FENCE();
                if (QUADP) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); lut_read_dma(2); }
                finish_emit(2);
                ballast();
                LAB_FENCE();
                if (QUADP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lut_read_dma(3); }
                finish_emit(3);
                ballast();
#else
                lut_issue(0);
                lut_issue(1);
                LAB_FENCE();
#if LAB_ROTATE
                next_noise();
                LAB_FENCE();
#endif
                if (QUADP) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); lut_read_dma(0); }
                finish_emit(0);
                ballast();
                LAB_FENCE();
                lut_issue(2);
                LAB_FENCE();
                if (QUADP) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); lut_read_dma(1); }
                finish_emit(1);
                ballast();
                LAB_FENCE();
                lut_issue(3);
                LAB_FENCE();
                if (QUADP) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); lut_read_dma(2); }
                finish_emit(2);
                ballast();
                LAB_FENCE();
                if (QUADP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lut_read_dma(3); }
                finish_emit(3);
                ballast();
#endif
                {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    // (aux 2 = nt on loads and stores: the two 25 GB frame streams bypass the L1 and do not displace the LUT's 1.6 MB from the XCD's L2:
                    //  chain 3 -2.6 % uniform, -4.6 % video-like; stores alone -2.4 / -3.3 %; sc1 or sc1 nt stores +1.5...2.5 % -- profiles/r05_ab_march_store_policy.json)
                    for (int m = 0; m < 4; ++m) xraw[m] = __builtin_amdgcn_raw_buffer_load_b96(rs_in, ld_voff[m], (rowbase + E) * 4, 2);
#pragma unroll
                    for (int m = 0; m < 4; ++m) __builtin_amdgcn_raw_buffer_store_b96(resq[m], rs_out, st_voff[m], out_soff, 2);
                    // a 96-bit store reads its data registers over several cycles; the backend pads the next VALU write of those
                    // registers only for stores WITHOUT an SGPR offset (GCNHazardRecognizer::createsVALUHazard) -- with one, as here, the
                    // R channel of the upper lanes of a 16-lane row came out as the NEXT row's value in builds whose schedule put a VALU
                    // write right behind the store (profiles/r04_noslp_dpp_fold_diff.log): pad by hand
                    asm volatile("s_nop 1" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) xin[m] = px3{__uint_as_float(xraw[m].x), __uint_as_float(xraw[m].y), __uint_as_float(xraw[m].z)};
#if LAB_ROTATE
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) nz[j][q] = nzn[j][q];
#endif
                // ---- advance; is the next row steady as well?
                ++rows_done;
                ++rho;
                rowbase += E;
                const int b0n = rowbase - q0s;
                more = rho <= r_last && (uint32_t)(b0n + 3 * 63 + 4) < G &&
                       (int64_t)rowbase + 3ll * (int64_t)G + 3 * 63 < (int64_t)M.numel - 2 &&
                       (int64_t)rowbase + E + 3ll * (int64_t)G + 3 * 63 <= (int64_t)li_max;
#pragma unroll
                for (int m = 0; m < (SHARPEN ? 4 : 0); ++m) {
                    ++ysc[m];
                    more = more && ysc[m] <= H - 1;
                }
            }
            if (LAB_BALLAST && bl0 == 123.456f) cout[0] = bl0;
            // per-lane row coordinates for the general steps that follow
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                yM[m] = yc[m] + rows_done - 1;
                yc[m] += rows_done;
                if (SHARPEN) {
                    if (yc[m] >= H) { yc[m] -= H; ++fc[m]; }
                } else {
                    while (yc[m] >= H) { yc[m] -= H; ++fc[m]; }      // without a stencil steady rows run across frame boundaries
                }
            }
        }
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
    uint64_t blocks = (jobs + LAB_WGW - 1) / LAB_WGW;
    if (blocks >= (1ull << 26)) return VRG_ERR_UNSUPPORTED;      // work-items per launch are counted in 32 bits
#if LAB_PERSIST
    { const uint64_t cap = (uint64_t)256 * 4 * LAB_PERSIST / LAB_WGW; if (blocks > cap) blocks = cap; }
#endif
#if LAB_EXTRA_LDS
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_march<STAGES, SHARPEN>), hipFuncAttributeMaxDynamicSharedMemorySize, LAB_EXTRA_LDS);
#endif
    hipLaunchKernelGGL((k_chain_march<STAGES, SHARPEN>), dim3((uint32_t)blocks), dim3(64 * LAB_WGW), LAB_EXTRA_LDS, st, in, out, M, D);
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
