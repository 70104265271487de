/*
 * vrgdg_hip.h -- C ABI of libvrgdg_hip.so: the MI355X (gfx950 / CDNA4) implementation of the
 * per-pixel video post-processing hot path of comfyui-vrgamedevgirl.
 *
 * This is the drop-in boundary: plain pointers, sizes and scalars, no torch types.  The Python
 * node classes (comfyui-vrgamedevgirl_amd/nodes.py, VRGDG_IV_Adjustments.py) bind it with ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.  All image pointers are
 * DEVICE pointers to fp32 NHWC frames ([F][H][W][C], C innermost), values nominally in [0,1].
 * `stream` is a hipStream_t passed as void* (0 = the null stream); every entry point only
 * enqueues work on that stream and returns immediately.  No entry point allocates device memory;
 * scratch is supplied by the caller (sizes from the *_scratch_bytes helpers).
 *
 * Return value: 0 = ok; VRG_ERR_* otherwise (vrg_error_string() gives the text; the Python layer
 * raises RuntimeError / ValueError like the reference's nodes do).
 *
 * Arithmetic contract: every fp32 operation of the reference is performed
 * as its own correctly rounded fp32 operation in the reference's order -- the kernels are built
 * with -ffp-contract=off and use IEEE divide / sqrt.  Scalars that the reference computes in
 * Python doubles (1.0 - s, strength/10, ...) are computed by the CALLER in double and passed
 * here already rounded to fp32.
 */
#ifndef VRGDG_HIP_H_
#define VRGDG_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRG_ABI_VERSION 8

enum vrg_status {
    VRG_OK = 0,
    VRG_ERR_BAD_ARG = 1,      /* null pointer, negative size, C < 3 where RGB is required ... */
    VRG_ERR_UNSUPPORTED = 2,  /* valid request this build does not implement */
    VRG_ERR_LAUNCH = 3,       /* hipLaunchKernel / hipGetLastError failed */
    VRG_ERR_NO_DEVICE = 4
};

enum vrg_border {
    VRG_BORDER_REPLICATE = 0, /* reference CPU/numpy path: np.pad(mode="edge")            (nodes.py:188-192) */
    VRG_BORDER_ZERO = 1       /* reference use_gpu=True path: avg_pool2d / conv2d padding=1 (nodes.py:171, 257) */
};

enum vrg_stencil_op {
    VRG_STENCIL_UNSHARP = 0,        /* FastUnsharpSharpen.apply_unsharp     nodes.py:156-209 */
    VRG_STENCIL_LAPLACIAN = 1,      /* FastLaplacianSharpen.apply_laplacian nodes.py:234-289 */
    VRG_STENCIL_SOBEL = 2,          /* FastSobelSharpen.apply_sobel         nodes.py:314-384 */
    VRG_STENCIL_NONE = 3
};

/*
 * Noise stream description: the torch-HIP `randn` mapping (ATen DistributionTemplates.h:52-99,
 * rocrand_philox4x32_10.h, rocrand_normal.h:52-68).  The frames are cut into RNG chunks of
 * `chunk_frames` frames (FastFilmGrain: batch_size, nodes.py:46-51; per-frame seeding: 1,
 * VRGDG_StandaloneVideoEnhancerNodes.py:268-271).  Chunk j (absolute index chunk0 + j) draws
 *     randn(chunk_numel) with Philox key  seed0 + (chunk0+j)*seed_stride,
 *                              offset      offset0 + (chunk0+j)*offset_stride   (multiple of 4)
 * and element li of the chunk takes component (li / G) % 4 of call (li / G) / 4 of Philox
 * subsequence li % G.  `grid_threads` = G = grid.x*256 of torch's calc_execution_policy for
 * chunk_numel on this device.  All chunks of one call must have the same numel (the host issues
 * a second call for a ragged tail chunk).
 */
typedef struct vrg_noise_desc {
    uint64_t seed0;
    uint64_t seed_stride;
    uint64_t offset0;
    uint64_t offset_stride;
    int64_t  chunk0;        /* absolute index of the first chunk handled by this call */
    int32_t  chunk_frames;  /* frames per RNG chunk (>= 1) */
    uint32_t grid_threads;  /* G */
} vrg_noise_desc;

/* ---------------------------------------------------------------------------------------------
 * a1/a2  Film grain.  Replaces FastFilmGrain.apply_grain (nodes.py:41-66),
 * _apply_film_grain_tensor (VRGDG_LUTVideoTools.py:262-277) and _apply_seeded_grain
 * (VRGDG_StandaloneVideoEnhancerNodes.py:262-278).
 *   g_c = fl(fl(S*fl(k_c*n_c)) + fl(T*n_G)),  k = (2,1,3);  out = clamp(fl(x + fl(g_c*I)), 0, 1)
 * `frames` must be a whole number of RNG chunks except that the last chunk may not be ragged
 * (see vrg_noise_desc).  C is fixed at 3.
 * ------------------------------------------------------------------------------------------- */
int vrg_grain_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width,
                  float intensity, float sat, float one_minus_sat,
                  const vrg_noise_desc* noise, void* stream);

/* f1  Unsharp, then per-frame-seeded grain, in ONE pass over the frames (24 B/px instead of 48): the effect order of the stand-alone
 * enhancer, _apply_effects_batch = _apply_unsharp then _apply_seeded_grain (VRGDG_StandaloneVideoEnhancerNodes.py:233-294).
 * out = vrg_grain_f32(vrg_stencil3x3_f32(in, VRG_OP_UNSHARP, border, strength), ...) bit for bit; `noise` as for vrg_grain_f32 with
 * chunk_frames == 1 (one generator per frame).  in != out.  Returns VRG_ERR_UNSUPPORTED -- and the caller runs the two entry points
 * above -- unless width % 4 == 0, width * 3 / 4 >= 256, chunk_frames == 1 and both pointers are 16-byte aligned. */
int vrg_sharpen_grain_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width,
                          float strength, int32_t border, float intensity, float sat, float one_minus_sat,
                          const vrg_noise_desc* noise, void* stream);

/* f1 x f3  The same pass on DECODED frames, uint8 B,G,R in and out -- the stand-alone enhancer's render loop body
 * _tensor_to_frames(_apply_effects_batch(_frames_to_tensor(frames))) (VRGDG_StandaloneVideoEnhancerNodes.py:311-324, 278-294,
 * 417-421) as ONE kernel moving 3 + 3 B/px: v / 255 at the load, unsharp (border as above), per-frame-seeded grain,
 * clip(x * 255, 0, 255) truncated to uint8 at the store.  out = vrg_f32rgb_to_u8bgr(vrg_sharpen_grain_f32(vrg_u8bgr_to_f32rgb(in)))
 * byte for byte.  in != out.  Any width, height and pointer alignment (ABI v7: frames with width % 4 == 0, width * 3 / 4 >= 256 and
 * 4-byte aligned pointers run on the frame's dword grid; everything else -- 854 x 480, 1366 x 768, thumbnails, frames that start off a
 * dword -- in flat byte space with unaligned dword accesses, same bytes).  VRG_ERR_UNSUPPORTED (the caller runs that three-kernel
 * route) only for chunk_frames != 1 and for a batch of fewer than four bytes. */
int vrg_sharpen_grain_u8(const uint8_t* in, uint8_t* out, int64_t frames, int32_t height, int32_t width,
                         float strength, int32_t border, float intensity, float sat, float one_minus_sat,
                         const vrg_noise_desc* noise, void* stream);

/* Same arithmetic with the N(0,1) noise supplied by the caller (device pointer, same shape as
 * `in`): the noise-injection form used to prove arithmetic parity against the CPU reference. */
int vrg_grain_injected_f32(const float* in, const float* noise, float* out, int64_t pixels,
                           float intensity, float sat, float one_minus_sat, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a4/a5/a6  3D LUT trilinear apply + strength blend.  Replaces VRGDG_LUTS._apply_cube_lut and
 * the blend in apply_lut (VRGDG_IV_Adjustments.py:288-361), _apply_lut_tensor
 * (VRGDG_LUTVideoTools.py:172-185).
 * A LUT is prepared once: vrg_lut_prepare_f32 rewrites the parsed table `lut` (device fp32
 * [N][N][N][3] indexed [blue][green][red], as _parse_cube_file returns it) into the gather-friendly
 * form the kernels read -- (N-1)^2*N records of 12 floats, one per (b0, g0, red node), the four (g,b)
 * corner values of each channel copied verbatim -- so that a pixel fetches one contiguous 96-byte
 * run (red nodes r0, r0+1) instead of eight scattered corners.  `cells` must hold vrg_lut_cells_floats(N) floats, 16-byte aligned.  2 <= N <= 256.
 * `channels` >= 3; channels beyond RGB are copied through.  blend_mode: 1 = LUT only (blend>=1),
 * 2 = fl(fl(x*one_minus_blend) + fl(y*blend)).  (blend <= 0 is the caller's no-op.)
 * ------------------------------------------------------------------------------------------- */
int64_t vrg_lut_cells_floats(int32_t lut_size);
int vrg_lut_prepare_f32(const float* lut, int32_t lut_size, float* cells, void* stream);
int vrg_lut3d_f32(const float* in, float* out, int64_t pixels, int32_t channels,
                  const float* cells, int32_t lut_size,
                  const float domain_min[3], const float domain_max[3],
                  int32_t blend_mode, float blend, float one_minus_blend, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a9/a10/a11  3x3 stencils, any channel count.  out = clamp(x + strength*f(3x3), 0, 1).
 * Sum orders: see csrc/vrg_pixel_math.hpp (reference order for the replicate border; raster
 * (kh,kw) order over the non-zero taps for the zero border).
 * ------------------------------------------------------------------------------------------- */
int vrg_stencil3x3_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width,
                       int32_t channels, int32_t op, int32_t border, float strength, void* stream);

/* Arithmetic policy of the Lab transforms and the statistics transfer (a7/a8).
 *   VRG_CM_MATH_DEVICE (default): every element-wise op is the one torch-ROCm executes for it on this GPU -- what the
 *       reference computes when ComfyUI runs ColorMatchToReference on the MI355X (nodes.py:98-115): `tensor / python
 *       scalar` = x * fl(1/c), torch.pow = ocml powf, tensor / tensor = IEEE quotient.  Bit-equal to the restated kornia
 *       formulas evaluated by torch on the device (tests/test_gpu_parity.py), given the same statistics.
 *   VRG_CM_MATH_FAST: IEEE quotients (torch-CPU behaviour) and table-driven powers with <= 0.534 ulp error instead of
 *       ocml powf: a few ulp from either reference, 1.2-1.4x faster than the device policy (which reaches ocml's value through
 *       a cheaper logarithm plus a rounding test, and through ocml's own operation sequence where the test fails). */
enum vrg_cm_math { VRG_CM_MATH_DEVICE = 0, VRG_CM_MATH_FAST = 1 };

/* ---------------------------------------------------------------------------------------------
 * a7/a8  Colour match (nodes.py:91-124 + kornia.color Lab transforms).
 * Pass 1: per-frame Lab statistics.  stats[f][c] = {n, mean, M2} in fp64 (M2 = sum (x-mean)^2),
 * c = L,a,b.  Deterministic two-stage reduction (no atomics).  `scratch` must hold
 * vrg_lab_stats_scratch_bytes(frames) bytes.
 * Pass 2: out = clamp(lab_to_rgb(K*((lab-mu)/sigma*sigma_ref+mu_ref) + T*lab)).
 *   img_ms / ref_ms: device fp32 [frames][3][2] = {mean, std_unbiased + 1e-5} (vrg_lab_stats_finalize
 *   converts the fp64 triple); ref frame of image frame f = (ref_frames == 1) ? 0 : f % ref_frames.
 * ------------------------------------------------------------------------------------------- */
int64_t vrg_lab_stats_scratch_bytes(int64_t frames);
int vrg_lab_stats_f32(const float* in, int64_t frames, int32_t height, int32_t width,
                      double* stats, void* scratch, int32_t cm_math, void* stream);
int vrg_lab_stats_finalize(const double* stats, float* mean_std, int64_t frames, void* stream);
/* The same statistics with the BITS the reference gets on this GPU: `lab.mean(dim=[2,3])` / `lab.std(dim=[2,3]) + 1e-5` as
 * torch-ROCm evaluates them (nodes.py:99-100, 109-110) -- ATen's reduce_kernel with MeanOps / WelfordOps in fp32, whose value
 * depends on the launch geometry torch derives from the tensor shape ([chunk_frames,3,H,W] per call: the node's batch_size, or the
 * whole reference batch), on the order in which thread accumulators, lanes and warps are combined, and on which multiply-adds of
 * the Welford update hipcc contracted in libtorch_hip.so.  csrc/vrg_torch_stats.hip replays that computation (geometry of the
 * MI355X: 256 CUs; VRG_ERR_UNSUPPORTED elsewhere) on the interleaved Lab image `lab` ([frames][H][W][3] fp32: the `lab_out` of
 * vrg_chain_stats_lab_f32).  Frames are taken `chunk_frames` per reference call, the last call holds the remainder.
 * mean_std = device fp32 [frames][3][2] = {mean, std + eps} (eps = 1e-5f for the node), the layout vrg_colormatch_apply_f32 /
 * vrg_chain_desc::img_ms / ref_ms read.  With these statistics the whole colour match is bit-equal to the reference's formulas
 * evaluated by torch on the device (tests/test_gpu_parity.py). */
int vrg_lab_stats_torch_f32(const float* lab, int64_t frames, int32_t height, int32_t width, int32_t chunk_frames,
                            float* mean_std, float eps, void* stream);
/* The same with a caller-supplied scratch buffer (16-byte aligned, vrg_lab_stats_torch_scratch_bytes(frames) bytes -- 0 when the batch is
 * too large for the form that uses one; NULL = none): small batches of video-sized frames are then reduced by EIGHT half-block
 * workgroups per frame (one wave per SIMD: the dependent Welford update chains are issue bound next to a second wave) plus a finishing
 * kernel.  Same result bits. */
int64_t vrg_lab_stats_torch_scratch_bytes(int64_t frames);
int vrg_lab_stats_torch_ws_f32(const float* lab, int64_t frames, int32_t height, int32_t width, int32_t chunk_frames,
                               float* mean_std, float eps, void* scratch, int64_t scratch_bytes, void* stream);
/* The same statistics with the LATENCY form allowed for calls of at most two frames (one accumulator per lane: seven-wave workgroups,
 * 65 KB of LDS each -- 0.78 instead of 1.12 ms for one 4K frame on an otherwise idle GPU, slower than the automatic choice beside a
 * full-size pass): for the reference frame's statistics of a small step.  Same arguments, same result bits. */
int vrg_lab_stats_torch_lat_f32(const float* lab, int64_t frames, int32_t height, int32_t width, int32_t chunk_frames,
                                float* mean_std, float eps, void* scratch, int64_t scratch_bytes, void* stream);
int vrg_colormatch_apply_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width,
                             const float* img_ms, const float* ref_ms, int32_t ref_frames,
                             float k, float one_minus_k, int32_t cm_math, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: frames shard across ranks with no data-path collective; the one exchange is this --
 * (n, mean, M2) triples of DISJOINT pixel sets (the rows of a reference frame reduced on different GPUs, vrg_lab_stats_f32 on
 * each rank's row slice) are combined in place into the statistics of their union with two SUM all-reduces over RCCL / xGMI
 * enqueued on `stream`: (n, n*mean) gives the global mean, then M2 + n*(mean - mean_tot)^2 gives the global M2.  72 bytes per
 * reference frame: latency bound.  `comm` is the caller's ncclComm_t (RCCL's nccl.h), passed as void*; `stats` = `count`
 * triples (count = frames*3), `scratch` = vrg_stats_allreduce_scratch_bytes(count) bytes of device memory.  RCCL is resolved
 * at run time from the copy the process already loaded (the library links only the HIP runtime): VRG_ERR_UNSUPPORTED when
 * there is none.  The Python host performs the same arithmetic through torch.distributed (sharding.allreduce_stats).
 * ------------------------------------------------------------------------------------------- */
int64_t vrg_stats_allreduce_scratch_bytes(int64_t count);
int vrg_stats_allreduce(double* stats, int64_t count, void* comm, void* scratch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused chain: grain -> LUT -> colour match -> 3x3 sharpen in one pass over HBM (12 B/px read +
 * 12 B/px written; colour match adds the 12 B/px statistics pass).  Bit-identical to running the
 * stand-alone entry points one after the other.  Stages are switched by `stages`.
 * ------------------------------------------------------------------------------------------- */
#define VRG_STAGE_GRAIN      1
#define VRG_STAGE_LUT        2
#define VRG_STAGE_COLORMATCH 4
#define VRG_STAGE_SHARPEN    8
/* With VRG_STAGE_COLORMATCH: `in` already holds the Lab image of the colour-match input (written by
 * vrg_chain_stats_lab_f32), so grain / LUT / rgb_to_lab are not re-evaluated in the apply pass. */
#define VRG_STAGE_FROM_LAB   16

typedef struct vrg_chain_desc {
    int32_t stages;               /* VRG_STAGE_* bits */
    int32_t variant;              /* 0 = automatic; 1 = LDS-tile / point-wise kernels; 2 = register-resident wave-march kernel
                                     (chains it cannot take -- colour match, chunks > 0x60000000 elements -- use 1) */
    /* grain */
    float intensity, sat, one_minus_sat;
    vrg_noise_desc noise;
    /* LUT (cell-major table from vrg_lut_prepare_f32) */
    const float* lut; int32_t lut_size;
    float domain_min[3], domain_max[3];
    int32_t blend_mode; float blend, one_minus_blend;
    /* colour match (statistics of the LUT output are produced by vrg_chain_stats_f32) */
    const float* img_ms; const float* ref_ms; int32_t ref_frames;
    float k, one_minus_k;
    /* sharpen */
    int32_t stencil_op, border; float strength;
    /* colour-match arithmetic policy: enum vrg_cm_math (0 = device-exact, the default) */
    int32_t cm_math;
} vrg_chain_desc;

int vrg_fused_chain_f32(const float* in, float* out, int64_t frames, int32_t height, int32_t width,
                        const vrg_chain_desc* desc, void* stream);
/* Lab statistics of the grain->LUT output (the input of the colour-match stage), same layout and
 * as vrg_lab_stats_f32; `scratch` must hold vrg_chain_stats_scratch_bytes(...) bytes. */
int vrg_chain_stats_f32(const float* in, int64_t frames, int32_t height, int32_t width,
                        const vrg_chain_desc* desc, double* stats, void* scratch, void* stream);
/* Scratch size for vrg_chain_stats_f32 / vrg_chain_stats_lab_f32 with this descriptor (chains that start with
 * grain use a pass that shares the Philox work and keeps one partial record per (workgroup, strip)). */
int64_t vrg_chain_stats_scratch_bytes(int64_t frames, int32_t height, int32_t width, const vrg_chain_desc* desc);
/* Same pass, additionally storing the Lab image it reduces (`lab_out`, same shape as `in`): the apply pass
 * then runs with VRG_STAGE_COLORMATCH | VRG_STAGE_FROM_LAB (| VRG_STAGE_SHARPEN) on `lab_out`.  Trades
 * 12 B/px of extra HBM traffic for not evaluating grain, the LUT gathers and six powers twice -- the chain is
 * ALU / L1-request bound, not HBM bound.  Results are bit-identical to the recomputing form.
 * `stats` may be NULL (then `scratch` may be NULL too): only the Lab image is produced -- the form used with the device
 * statistics (vrg_lab_stats_torch_f32 reduces `lab_out`). */
int vrg_chain_stats_lab_f32(const float* in, float* lab_out, int64_t frames, int32_t height, int32_t width,
                            const vrg_chain_desc* desc, double* stats, void* scratch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 13-slider Adjust of the video routes.
 * Replaces _apply_adjust_tensor, VRGDG_LUTVideoTools.py:307-391: clamp, white balance (temperature / tint),
 * exposure, contrast, saturation, highlights / shadows / whites / blacks masks, clarity (k x k reflect box,
 * k = min(9, odd(H), odd(W))), sharpen (3x3 replicate box), fade, vignette, clamp.  The host rounds the
 * slider arithmetic (Python doubles, :309-315) once to fp32 and passes the terms below; the kernels keep the
 * reference's fp32 rounding order, including avg_pool2d's raster-order running sums.
 * `tmp` (same shape as `in`) is needed only when clarity and sharpen are both active; it may be NULL otherwise.
 * ------------------------------------------------------------------------------------------- */
typedef struct vrg_adjust_desc {
    int32_t enabled;                 /* 0: output = clamp(in, 0, 1) (:308) */
    float shift[3];                  /* temp/400 - tint/900, tint/450, -temp/400 - tint/900 (:319-326) */
    float exposure;                  /* 2 ** (exposure/100) (:327) */
    float contrast, saturation;      /* 1 + slider/100 (:328,330) */
    float highlights, shadows;       /* slider/220 (:336-337) */
    float whites, blacks;            /* slider/240 (:338-339) */
    int32_t has_clarity; float clarity;      /* slider != 0; slider/100 (:342-356) */
    int32_t has_sharpen; float sharpen;      /* slider > 0;  slider/100 (:358-372) */
    int32_t has_fade; float fade_mul, fade_add;   /* 1 - fade*0.35, fade*0.18 (:374-375) */
    int32_t has_vignette; float vignette;    /* slider > 0; slider/100 (:377-389) */
    /* `tensor / 0.45`, `/ 1.05` (:333-334, :388): the reference runs on the device its `device` argument names -- the
     * IEEE quotient on the CPU, x * fl32(1.0 / c) on the GPU (ATen BinaryDivTrueKernel).  enum vrg_adjust_div. */
    int32_t div_mode;
} vrg_adjust_desc;
enum vrg_adjust_div { VRG_ADJUST_DIV_IEEE = 0, VRG_ADJUST_DIV_DEVICE = 1 };

int vrg_adjust_f32(const float* in, float* out, float* tmp, int64_t frames, int32_t height, int32_t width,
                   const vrg_adjust_desc* desc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * uint8 BGR frames at the video I/O edge.
 * Replaces _frames_to_tensor / _tensor_to_frames (VRGDG_LUTVideoTools.py:736-752,
 * VRGDG_StandaloneVideoEnhancerNodes.py:311-324): `astype(float32) / 255.0` after the BGR->RGB swap on the way
 * in, `clip(x * 255.0, 0, 255).astype(uint8)` (truncation) and RGB->BGR on the way out.  In the *_u8 entry points
 * both conversions happen inside the kernel that does the work, so a route batch (_process_video_batch,
 * _process_film_grain_batch, _process_adjust_batch, :1365-1386) moves 3 + 3 B/px instead of 12 + 12; results are
 * identical to convert -> fp32 entry point -> convert.  Frames are [frames][height][width][3] uint8, B,G,R order.
 * ------------------------------------------------------------------------------------------- */
int vrg_u8bgr_to_f32rgb(const uint8_t* in, float* out, int64_t pixels, void* stream);
int vrg_f32rgb_to_u8bgr(const float* in, uint8_t* out, int64_t pixels, void* stream);
/* Exact per-frame channel sums of uint8 frames, `sums` = [frames][3 channels in memory order][sum, sum of squares]
 * as 64-bit integers (zeroed by the call).  Replaces PIL.ImageStat.Stat(...).sum / .sum2 in the opening colour match
 * (VRGDG_WorkflowRunnerNodes.py:4385-4392); mean / stddev follow on the host in double exactly as ImageStat does. */
int vrg_u8_channel_sums(const uint8_t* frames, int64_t frames_n, int32_t height, int32_t width, unsigned long long* sums,
                        void* stream);
/* The per-pixel step of the opening colour match with ffmpeg's filter arithmetic instead of this library's LUT stage
 * (reference filter graph lut3d + blend, VRGDG_WorkflowRunnerNodes.py:4407-4412): tetrahedral interpolation of the cube on
 * 8-bit B,G,R frames, truncation to 8 bits, then `A*(1-w)+B*w` in double per byte with the per-frame weight `weights[f]`
 * (device pointer, NULL = no blend), truncated.  `table` = the parsed .cube [N][N][N][3] fp32 (index [blue][green][red]) on
 * the device, `domain_min/max` = host float[3].  Restated from ffmpeg's published sources (libavfilter/vf_lut3d.c,
 * vf_blend.c); ffmpeg is absent here, so this entry point's parity is unpinned. */
int vrg_lut3d_tetra_u8(const uint8_t* in, uint8_t* out, int64_t frames, int64_t pixels_per_frame, const float* table,
                       int32_t lut_size, const float* domain_min, const float* domain_max, const double* weights, void* stream);
/* grain / LUT / 3x3 sharpen in any combination (no colour match: VRG_ERR_UNSUPPORTED); desc as for vrg_fused_chain_f32 */
int vrg_fused_chain_u8(const uint8_t* in, uint8_t* out, int64_t frames, int32_t height, int32_t width,
                       const vrg_chain_desc* desc, void* stream);
/* `tmp`: fp32, frames*height*width*3 floats, needed only when clarity and sharpen are both active */
int vrg_adjust_u8(const uint8_t* in, uint8_t* out, float* tmp, int64_t frames, int32_t height, int32_t width,
                  const vrg_adjust_desc* desc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Introspection
 * ------------------------------------------------------------------------------------------- */
int vrg_abi_version(void);
const char* vrg_error_string(int status);
/* multiProcessorCount and maxThreadsPerMultiProcessor of the current device (what torch's
 * calc_execution_policy reads); returns VRG_ERR_NO_DEVICE without a GPU. */
int vrg_device_info(int32_t* cu_count, int32_t* max_threads_per_cu);
/* HIP-event timing helper for bench.py: records an event on `stream` and returns elapsed ms
 * between two recorded events (torch.cuda.Event only sees torch's current stream). */
int vrg_event_create(void** ev);
int vrg_event_record(void* ev, void* stream);
int vrg_event_elapsed_ms(void* start, void* stop, float* ms);
int vrg_event_destroy(void* ev);

/* Raw N(0,1) stream of the given chunks, bit-identical to torch.randn on this device; `frame_elems` = H*W*3.  The node layer materialises the
 * noise of an RNG chunk of more than 2^29 elements with it, leaf by leaf as ATen splits such a randn (reference: nodes.py:51 with batch_size 0
 * or >= 22 4K frames), and the tests compare the stream itself against torch. */
int vrg_noise_f32(float* out, int64_t frames, int64_t frame_elems,
                  const vrg_noise_desc* noise, void* stream);

/* First-use self-check of the one toolchain-calibrated arithmetic form of the colour match (csrc/vrg_pixel_math.hpp dev_pow_ziv: table
 * logarithm + rounding test whose half-widths were measured against ROCm 7.0's ocml): out[i] = the power of in[i] exactly as call site
 * `site` of the Lab transforms evaluates it -- 0: sRGB -> linear (x^2.4 on [0.0625, 2]), 1: linear -> sRGB (x^(1/2.4) on [0.0031308, 4]),
 * 2: the Lab cube root (x^(1/3) on [0.008856, 4]).  The host compares with this ROCm's powf (torch.pow, which the reference's kornia calls:
 * nodes.py:98, 115) and reports a mismatch (ops.toolchain_status). */
int vrg_selfcheck_pow_f32(const float* in, float* out, int64_t n, int32_t site, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Host side of the node path (reference: nodes.py:50, 61-66 -- CPU tensors in, `images.to(device)` per batch)
 * ------------------------------------------------------------------------------------------- */
/* memcpy of `bytes` from `src` to `dst` (host pointers, not overlapping) split over `threads` host threads (0 = 8; at most 64; parts of
 * whole pages, none below 2 MiB).  The node layer stages pageable frames into a page-locked ring with it, so that upload, kernels and
 * download of a pageable batch overlap like those of a page-locked one.  Blocks until the bytes are there.  No device work. */
int vrg_host_copy(void* dst, const void* src, int64_t bytes, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* VRGDG_HIP_H_ */
