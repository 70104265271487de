/* vrgdg_hip_debug.h -- self-tests, element-wise probes and timing probes: the C ABI of libvrgdg_hip_debug.so, a library of its own (round 6).
 *
 * NOT part of the drop-in boundary (include/vrgdg_hip.h): nothing a host of the node pack calls.  These entry points exist for the
 * test suite (exhaustive device sweeps that PROVE an arithmetic substitution, element-wise pieces of the colour-match arithmetic to
 * compare with the torch op the reference executes, the launch geometry of the replayed reductions) and for the measurement tools
 * (tools/probe_gather.py, tools/copy_ceiling.py, tools/probe_valu.py).  They live in their own translation unit (csrc/vrg_probe.hip), linked into libvrgdg_hip_debug.so and into nothing the nodes load,
 * and are bound separately by _hip.py (`_DEBUG_SIGNATURES`); tests/test_abi.py holds both headers to the library's exports. */
#ifndef VRGDG_HIP_DEBUG_H_
#define VRGDG_HIP_DEBUG_H_
#include "vrgdg_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Device self-test: sweeps all 2^32 fp32 inputs through the FMA-based constant divisions of the kernels
 * (csrc/vrg_pixel_math.hpp div_const / div9) and counts disagreements with the IEEE quotient.
 * counts18 (device, 18 x u64): [0..8] mismatches for 1e-30 <= |x| <= 1e30 (expected 0 for all nine
 * constants), [9..17] mismatches outside that range. */
int vrg_selftest_divconst(unsigned long long* counts18, void* stream);
/* Device self-test: the trimmed correctly-rounded square root of the Box-Muller radius (csrc/vrg_pixel_math.hpp
 * sqrt_normal_range) against the backend's IEEE sqrt for all 2^32 Philox words; counts1[0] = mismatches (expected 0). */
int vrg_selftest_bm_radius(unsigned long long* counts1, void* stream);
/* Device self-test: the Welford update's division by the running count as the statistics kernels evaluate it (reciprocal + two FMAs,
 * csrc/vrg_tstats_body.hpp) against the IEEE quotient, for the counts n_first .. n_first + n_count - 1 (< 2^24) x all 2^23 fp32
 * significands; mismatches1[0] (device, zeroed by the caller) += mismatches (expected 0). */
int vrg_selftest_welford_division(unsigned long long* mismatches1, uint32_t n_first, uint32_t n_count, void* stream);
/* Device self-test of the DPP lane shifts the wave-march kernel relies on: out128[i] = value held by lane i-1,
 * out128[64+i] = value held by lane i+1, for lane values 0..63. */
int vrg_selftest_lanes(float* out128, void* stream);
/* Element-wise pieces of the colour-match arithmetic for the parity tests (n values, or n triples for op 5..8):
 * op 0 ocml powf(x, y); 1 x * fl(1/y); 2 x / y; 3 fast-policy pow_pos(x, y); 4 fast-policy cube root;
 * 5 / 6 rgb->Lab / Lab->rgb with the device policy; 7 / 8 the same with the fast policy; 9 / 10 / 11 dev_pow(x, y), the
 * transcription of ocml powf without its special-case scaffolding that the device policy evaluates: 9 any x > 0 and finite y,
 * 10 the flavour of sRGB -> linear (x >= 2^-20, 0 < y <= 4), 11 the flavour of linear -> sRGB and the Lab cube root
 * (x >= 2^-20, 0 < y <= 0.5); 12 / 13 / 14 dev_pow_ziv(x, y) as sRGB -> linear / linear -> sRGB / the Lab cube root call it
 * (table logarithm + rounding test, the transcription where the test fails or x is outside the call site's domain);
 * 15: 1.0 where the rounding test of dev_pow_ziv fails for (x, y), else 0.0; 16 / 17 ocml's ln x (epln as transcribed), head / tail;
 * 18 / 19 dev_pow_ziv's table ln x, head / tail; 20 (n triples {lab, mean, std} -> triples): the unscaled FMA form of (lab - mean) / std,
 * the division itself, and 1.0 where the conditions of the former hold (vrg_pixel_math.hpp, SigmaRecip). */
int vrg_debug_cm_math(const float* in, float* out, int64_t n, int32_t op, float y, void* stream);
/* Host-only (no GPU needed): the launch geometry vrg_lab_stats_torch_f32 derives -- torch's setReduceConfig on the MI355X -- for
 * `num_outputs` outputs of `reduce_len` contiguous fp32 elements, `vec` = 4 (mean) or 2 (std): cfg4 = {block_width, block_height,
 * reduction split across the rows (1) or one output per row (0), vectorised thread loop (1) or strided (0)}. */
int vrg_debug_torch_reduce_config(int64_t num_outputs, int64_t reduce_len, int32_t vec, int32_t* cfg4);
/* Timing probe for LUT record fetch patterns (tools/gpu_diag.py); `out` = one float per pixel (a checksum).
 * mode 0: 6 x 16 B per lane; 1: 3 x 16 B; 2: quad-cooperative 64-B fetches; 3: 64-B records; 4: cell-major 128-B aligned records;
 * 5 / 6: channel split -- one / two channels of the node table in LDS (8 ds_read per pixel), the rest gathered (4 / 2 x 16 B);
 * 9-12: one piece / pieces 0 and 5 / the six pieces by LDS-DMA / quad-cooperative LDS-DMA; 13-18: mode 0's requests with cache-policy
 * bits on the loads (none, sc0, sc1, nt, sc0 sc1, sc0 sc1 nt); 19: mode 12 over the cell-major table of mode 4. */
int vrg_debug_lut_fetch(const float* in, float* out, int64_t pixels, const float* cells, int32_t lut_size, int32_t mode, void* stream);
/* Issue-rate probe (tools/gpu_diag.py --valu): `blocks` x 256 threads each issue iters x 64 instructions of one kind
 * (mode 0 v_fma_f32, 1 v_mad_u64_u32, 2 v_log_f32, 3 v_pk_fma_f32, 4 v_xor_b32, 5 sqrt/sin/cos/rcp mix,
 * 6 v_cmp+v_cndmask pairs, 7 v_mul_f64/v_fma_f64; round 6, the instruction classes of the march's row step: 8 v_add_f32, 9 v_add_f32_dpp
 * wave_shr:1, 10 v_add_f32_dpp row_shr:1, 11 v_mov_b32_dpp quad_perm, 12 v_cndmask_b32, 13 v_max_f32, 14 v_mul_f32, 15 v_cvt_f32_u32,
 * 16 v_mul_hi_u32, 17 v_mul_lo_u32, 18 v_add_f32 -> s_nop 1 -> v_add_f32_dpp of its result (the padded hazard), 19 v_sin_f32,
 * 20 v_pk_mul_f32 / v_pk_add_f32, 21 dependent v_mul_f32 -> v_add_f32 pairs in 8 chains); `out` = blocks*256 floats.  Measures the VALU roofline. */
int vrg_debug_valu_rate(float* out, int32_t blocks, int32_t iters, int32_t mode, void* stream);
/* Streaming-copy ceiling: out[i] = in[i] over n_floats fp32 values (a multiple of 4; 16-byte aligned
 * pointers), 16 B per lane.  mode 0: plain loads / stores; 1: non-temporal; 2: non-temporal, four float4 per thread; 3: read only;
 * 4: write only (a constant).  What every streaming kernel of this library is priced against, next to the 8 TB/s spec peak. */
int vrg_debug_copy_f32(const float* in, float* out, int64_t n_floats, int32_t mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VRGDG_HIP_DEBUG_H_ */
