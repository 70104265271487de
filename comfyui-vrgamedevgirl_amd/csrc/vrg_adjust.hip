// vrg_adjust.hip -- the 13-slider "Adjust" of the video routes:
// _apply_adjust_tensor, VRGDG_LUTVideoTools.py:307-391 of the reference.  gfx950 only.
//
//   point stage : clamp, white-balance shift, exposure, contrast, saturation, highlight / shadow / white / black
//                 masks on the post-saturation luma                                       (pure per pixel)
//   clarity     : x + (x - box9_reflect(x)) * clarity * 1.55 * (0.35 + midtone(x) * 0.65)  (k x k box, k = min(9, odd H, odd W))
//   sharpen     : x + (x - box3_replicate(x)) * sharpen * 5
//   tail        : fade, vignette (torch.linspace grid), clamp
//
// Bit-exactness needs the reference's summation order: avg_pool2d adds the k*k taps in raster order, one
// rounding per add, then divides by k*k.  The stencil kernels therefore keep one sequential chain per output
// (no separable / sliding sums) and spend their effort on issue slots instead (k_adjust_box below).  The point
// stage is recomputed for the halo instead of being stored.  With clarity AND sharpen the second box needs the
// first one's result in a 1-pixel ring: two passes through a caller-supplied temporary (48 B/px); every other
// combination is a single 24 B/px pass.  All kernels are instruction-issue bound (81 dependent adds per pixel and
// channel for clarity), not HBM bound; DESIGN.md section 8 has the measured table.
#include "vrg_common.hpp"
#include "vrg_adjust_math.hpp"

namespace vrg {

// ---------------------------------------------------------------------------------------------------------
// no stencil: point stage + tail, one pixel per thread.  enabled == 0: clamp only.
// ---------------------------------------------------------------------------------------------------------
template <class IO>
__global__ __launch_bounds__(256) void k_adjust_point(const typename IO::elem* __restrict__ in, typename IO::elem* __restrict__ out,
                                                       int32_t H, int32_t W, AdjustK A) {
    const int32_t ppf = H * W;
    const int32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= ppf) return;
    const int64_t at = (int64_t)blockIdx.y * ppf + p;
    const px3 s = IO::load_stream(in + at);
    const float x[3] = {s.r, s.g, s.b};
    float v[3];
    if (!A.enabled) {
        v[0] = clamp01(x[0]); v[1] = clamp01(x[1]); v[2] = clamp01(x[2]);
    } else {
        adjust_point(A, x, v);
        adjust_tail(A, p / W, p % W, H, W, v);
    }
    IO::store_stream(out + at, px3{v[0], v[1], v[2]});
}

// ---------------------------------------------------------------------------------------------------------
// box-detail stencils, K = 9 (clarity, reflect border) and K = 3 (sharpen, replicate border).
//
// Tile 32 x 64 output pixels (+ halo) in LDS as three planes, 256 threads, each thread 4 adjacent columns x 2 adjacent
// rows.  The running sums cannot be re-associated, so the work is 81 (9) dependent adds per pixel and channel; what
// can be saved is LDS traffic and issue slots around them: a tile row is read once (ds_read_b128) and feeds the chains
// of both owned rows (row y uses tile rows ly..ly+K-1, row y+1 uses ly+1..ly+K) and of all four columns, every chain in
// raster order.  [A vertical-pair layout V[r][c] = (T[r][c], T[r+1][c]) with v_pk_add_f32 chains was built and measured:
// packed fp32 issues at 1.8x the cost of a plain add on this chip (profiles/r01_valu_issue_rate.json), so it saved
// 10 % of the add slots, while its doubled LDS footprint (68 KB) halved the workgroups per CU -- 67 vs 81 Gpix/s.]
// PRE: the input is the frame and the point stage is applied while filling LDS (recomputed for the halo instead of
// stored); otherwise the input already is the clarity result.  TAIL: fade / vignette / clamp before storing.
// ---------------------------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int AT_H = 32, AT_W = 64;

template <int K, bool PRE, bool TAIL, class IN = IoF32, class OUT = IoF32>
__global__ __launch_bounds__(256) void k_adjust_box(const typename IN::elem* __restrict__ in, typename OUT::elem* __restrict__ out,
                                                           int32_t H, int32_t W, int32_t tiles_x, AdjustK A) {
    constexpr int R = K / 2, LR = AT_H + 2 * R, LC = AT_W + 2 * R, PITCH = LC + 4;     // LC % 4 == 0 or 2: keep rows 16-B aligned
    constexpr int NCOL = ((4 + K - 1) + 3) / 4 * 4;                                      // floats per row and thread (b128 multiples)
    static_assert((LC + 4) % 2 == 0, "alignment");
    __shared__ __attribute__((aligned(16))) float T[3][LR][PITCH + (PITCH % 4 ? 4 - PITCH % 4 : 0)];
    const int32_t ty0 = (blockIdx.x / tiles_x) * AT_H;
    const int32_t tx0 = (blockIdx.x % tiles_x) * AT_W;
    const int64_t fbase = (int64_t)blockIdx.y * H * W;
    const typename IN::elem* fin = in + fbase;
    constexpr int NIT = (LR * LC + 255) / 256;
    px3 pre[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + 256 * it;
        const int hy = i / LC, hx = i - hy * LC;
        int y = ty0 + hy - R, x = tx0 + hx - R;
        if (K != 3) {
            if (y < 0) y = -y;
            if (y > H - 1) y = 2 * (H - 1) - y;
            if (x < 0) x = -x;
            if (x > W - 1) x = 2 * (W - 1) - x;
        }
        y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
        pre[it] = IN::load(fin + (y * W + x));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + 256 * it;
        if (i >= LR * LC) break;
        const int hy = i / LC, hx = i - hy * LC;
        float v[3] = {pre[it].r, pre[it].g, pre[it].b};
        if (PRE) {
            float o[3];
            adjust_point(A, v, o);
            v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
        }
        T[0][hy][hx] = v[0];
        T[1][hy][hx] = v[1];
        T[2][hy][hx] = v[2];
    }
    __syncthreads();

    const int lx = (threadIdx.x & 15) * 4, ly = (threadIdx.x >> 4) * 2;
    float blur[3][2][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc[2][4];
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[0][o] = acc[1][o] = 0.0f;
#pragma unroll
        for (int rho = 0; rho <= K; ++rho) {
            const float* row = &T[c][ly + rho][lx];
            float r[NCOL];
#pragma unroll
            for (int j = 0; j < NCOL; j += 4) {
                const f4 t = *reinterpret_cast<const f4*>(row + j);
                r[j] = t.x; r[j + 1] = t.y; r[j + 2] = t.z; r[j + 3] = t.w;
            }
            if (rho < K) {
#pragma unroll
                for (int dx = 0; dx < K; ++dx)
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[0][o] = acc[0][o] + r[o + dx];
            }
            if (rho > 0) {
#pragma unroll
                for (int dx = 0; dx < K; ++dx)
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[1][o] = acc[1][o] + r[o + dx];
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int o = 0; o < 4; ++o) blur[c][rr][o] = K == 3 ? div9(acc[rr][o]) : VRG_ADJ_DIV(acc[rr][o], (float)(K * K));
    }
    typename OUT::elem* fout = out + fbase;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int y = ty0 + ly + rr;
        if (y >= H) continue;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int x = tx0 + lx + o;
            if (x >= W) continue;
            const float ctr[3] = {T[0][ly + rr + R][lx + R + o], T[1][ly + rr + R][lx + R + o], T[2][ly + rr + R][lx + R + o]};
            const float bl[3] = {blur[0][rr][o], blur[1][rr][o], blur[2][rr][o]};
            float v[3];
            if (K == 3) adjust_sharpen_mix(A, ctr, bl, v);
            else adjust_clarity_mix(A, ctr, bl, v);
            if (TAIL) adjust_tail(A, y, x, H, W, v);
            OUT::store(fout + (y * W + x), px3{v[0], v[1], v[2]});
        }
    }
}

// Frames smaller than 9 pixels in a dimension: the clarity box shrinks to 7, 5 or 3 (:349-352).  One pixel per
// thread straight from global memory, point stage per tap -- thumbnails only, not a performance path.
template <bool TAIL, class IN = IoF32, class OUT = IoF32>
__global__ __launch_bounds__(256) void k_adjust_box_small(const typename IN::elem* __restrict__ in, typename OUT::elem* __restrict__ out,
                                                           int32_t H, int32_t W, AdjustK A) {
    const int32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int64_t fbase = (int64_t)blockIdx.y * H * W;
    const typename IN::elem* fin = in + fbase;
    const int y = p / W, x = p % W, r = A.box / 2;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            int yy = y + dy, xx = x + dx;
            if (yy < 0) yy = -yy;
            if (yy > H - 1) yy = 2 * (H - 1) - yy;
            if (xx < 0) xx = -xx;
            if (xx > W - 1) xx = 2 * (W - 1) - xx;
            const px3 s = IN::load(fin + (yy * W + xx));
            const float t[3] = {s.r, s.g, s.b};
            float o[3];
            adjust_point(A, t, o);
            acc[0] = acc[0] + o[0]; acc[1] = acc[1] + o[1]; acc[2] = acc[2] + o[2];
        }
    const float kk = (float)(A.box * A.box);
    const px3 s = IN::load(fin + p);
    const float t[3] = {s.r, s.g, s.b};
    float ctr[3], v[3];
    adjust_point(A, t, ctr);
    const float bl[3] = {acc[0] / kk, acc[1] / kk, acc[2] / kk};
    adjust_clarity_mix(A, ctr, bl, v);
    if (TAIL) adjust_tail(A, y, x, H, W, v);
    OUT::store(out + fbase + p, px3{v[0], v[1], v[2]});
}

template <class IO>
static int launch_adjust(const void* in, void* out, float* tmp, int64_t frames, int32_t height, int32_t width, const vrg_adjust_desc* d,
                         hipStream_t st) {
    typedef typename IO::elem elem;
    if (!in || !out || !d || frames < 0 || height <= 0 || width <= 0) return VRG_ERR_BAD_ARG;
    if (d->div_mode != VRG_ADJUST_DIV_IEEE && d->div_mode != VRG_ADJUST_DIV_DEVICE) return VRG_ERR_BAD_ARG;
    if (frames == 0) return VRG_OK;
    const int64_t ppf = (int64_t)height * width;
    if (ppf > 0x7fffffff / 3) return VRG_ERR_UNSUPPORTED;
    AdjustK A{};
    A.enabled = d->enabled;
    A.div_device = d->div_mode == VRG_ADJUST_DIV_DEVICE;
    for (int c = 0; c < 3; ++c) A.shift[c] = d->shift[c];
    A.exposure = d->exposure; A.contrast = d->contrast; A.saturation = d->saturation;
    A.highlights = d->highlights; A.shadows = d->shadows; A.whites = d->whites; A.blacks = d->blacks;
    A.clarity = d->clarity; A.sharpen = d->sharpen;
    A.fade_mul = d->fade_mul; A.fade_add = d->fade_add; A.vignette = d->vignette;
    A.has_fade = d->has_fade; A.has_vignette = d->has_vignette;
    const int box = adjust_box_size(height, width);
    A.box = box;
    A.step_y = linspace_step(height);
    A.step_x = linspace_step(width);
    A.has_clarity = d->enabled && d->has_clarity && box >= 3;      // kernel < 3: blur == source, detail == 0, x + 0*... == x
    A.has_sharpen = d->enabled && d->has_sharpen;
    const bool both = A.has_clarity && A.has_sharpen;
    if (both && !tmp) return VRG_ERR_BAD_ARG;
    const elem* src = reinterpret_cast<const elem*>(in);
    elem* dst = reinterpret_cast<elem*>(out);
    const int tx = (width + AT_W - 1) / AT_W, ty = (height + AT_H - 1) / AT_H;
    for (int64_t f0 = 0; f0 < frames; f0 += 32768) {
        const uint32_t nf = (uint32_t)(frames - f0 < 32768 ? frames - f0 : 32768);
        const elem* s = src + f0 * ppf;
        elem* o = dst + f0 * ppf;
        px3* mid = both ? reinterpret_cast<px3*>(tmp) + f0 * ppf : nullptr;     // fp32 ring between the two boxes
        const dim3 tg((uint32_t)(tx * ty), nf);
        const dim3 pg((uint32_t)((ppf + 255) / 256), nf);
        if (!A.has_clarity && !A.has_sharpen) hipLaunchKernelGGL(k_adjust_point<IO>, pg, dim3(256), 0, st, s, o, height, width, A);
        if (A.has_clarity) {
            if (box == 9) {
                if (both) hipLaunchKernelGGL((k_adjust_box<9, true, false, IO, IoF32>), tg, dim3(256), 0, st, s, mid, height, width, tx, A);
                else hipLaunchKernelGGL((k_adjust_box<9, true, true, IO, IO>), tg, dim3(256), 0, st, s, o, height, width, tx, A);
            } else {
                if (both) hipLaunchKernelGGL((k_adjust_box_small<false, IO, IoF32>), pg, dim3(256), 0, st, s, mid, height, width, A);
                else hipLaunchKernelGGL((k_adjust_box_small<true, IO, IO>), pg, dim3(256), 0, st, s, o, height, width, A);
            }
        }
        if (A.has_sharpen) {
            if (both) hipLaunchKernelGGL((k_adjust_box<3, false, true, IoF32, IO>), tg, dim3(256), 0, st, (const px3*)mid, o, height, width, tx, A);
            else hipLaunchKernelGGL((k_adjust_box<3, true, true, IO, IO>), tg, dim3(256), 0, st, s, o, height, width, tx, A);
        }
        if (hipGetLastError() != hipSuccess) return VRG_ERR_LAUNCH;
    }
    return VRG_OK;
}

// stand-alone conversions (the enhancer's sharpen -> per-frame-seeded grain order runs on fp32 tensors)
__global__ __launch_bounds__(256) void k_u8_to_f32(const bgr8* __restrict__ in, px3* __restrict__ out, int64_t pixels) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < pixels) store_px_stream(out + p, IoU8::load(in + p));
}
__global__ __launch_bounds__(256) void k_f32_to_u8(const px3* __restrict__ in, bgr8* __restrict__ out, int64_t pixels) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < pixels) IoU8::store(out + p, load_px_stream(in + p));
}

// Exact per-frame channel sums of uint8 frames: sum and sum of squares per channel as 64-bit integers (what
// PIL.ImageStat derives mean / stddev from in the reference's opening colour match, VRGDG_WorkflowRunnerNodes.py:4385-4392).
// Integer arithmetic: bit-exact and independent of the reduction order, so plain 64-bit atomics are deterministic.
__global__ __launch_bounds__(256) void k_u8_channel_sums(const uint8_t* __restrict__ frames, int64_t pixels, unsigned long long* __restrict__ sums) {
    const uint8_t* f = frames + (int64_t)blockIdx.y * pixels * 3;
    unsigned long long s[3] = {0, 0, 0}, q[3] = {0, 0, 0};
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t v = f[3 * p + c];
            s[c] += v;
            q[c] += v * v;
        }
    }
    __shared__ unsigned long long red[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        unsigned long long a = s[c], b = q[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 64);
            b += __shfl_down(b, o, 64);
        }
        if (lane == 0) { red[wave][2 * c] = a; red[wave][2 * c + 1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const unsigned long long t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(sums + (int64_t)blockIdx.y * 6 + threadIdx.x, t);
    }
}

}  // namespace vrg

using namespace vrg;

extern "C" {

int vrg_adjust_f32(const float* in, float* out, float* tmp, int64_t frames, int32_t height, int32_t width, const vrg_adjust_desc* d,
                   void* stream) {
    return launch_adjust<IoF32>(in, out, tmp, frames, height, width, d, (hipStream_t)stream);
}

int vrg_adjust_u8(const uint8_t* in, uint8_t* out, float* tmp, int64_t frames, int32_t height, int32_t width, const vrg_adjust_desc* d,
                  void* stream) {
    return launch_adjust<IoU8>(in, out, tmp, frames, height, width, d, (hipStream_t)stream);
}

int vrg_u8_channel_sums(const uint8_t* frames, int64_t n_frames, int32_t height, int32_t width, unsigned long long* sums, void* stream) {
    if (!frames || !sums || n_frames < 0 || height <= 0 || width <= 0) return VRG_ERR_BAD_ARG;
    if (n_frames == 0) return VRG_OK;
    if (n_frames > 65535) return VRG_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, (size_t)n_frames * 6 * sizeof(unsigned long long), st) != hipSuccess) return VRG_ERR_LAUNCH;
    const int64_t pixels = (int64_t)height * width;
    int64_t blocks = (pixels + 256 * 16 - 1) / (256 * 16);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(k_u8_channel_sums, dim3((uint32_t)blocks, (uint32_t)n_frames), dim3(256), 0, st, frames, pixels, sums);
    return hipGetLastError() == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH;
}

int vrg_u8bgr_to_f32rgb(const uint8_t* in, float* out, int64_t pixels, void* stream) {
    if (!in || !out || pixels < 0) return VRG_ERR_BAD_ARG;
    if (pixels == 0) return VRG_OK;
    if (pixels > (int64_t)0x7fffffff * 256) return VRG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_u8_to_f32, dim3((uint32_t)((pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const bgr8*>(in), reinterpret_cast<px3*>(out), pixels);
    return hipGetLastError() == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH;
}

int vrg_f32rgb_to_u8bgr(const float* in, uint8_t* out, int64_t pixels, void* stream) {
    if (!in || !out || pixels < 0) return VRG_ERR_BAD_ARG;
    if (pixels == 0) return VRG_OK;
    if (pixels > (int64_t)0x7fffffff * 256) return VRG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_f32_to_u8, dim3((uint32_t)((pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const px3*>(in), reinterpret_cast<bgr8*>(out), pixels);
    return hipGetLastError() == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH;
}

}  // extern "C"
