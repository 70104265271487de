// vrg_tstats_config.hpp -- the launch geometry torch's setReduceConfig picks for the reductions csrc/vrg_torch_stats.hip replays (host code;
// shared with the debug library's vrg_debug_torch_reduce_config, which reports it to the tests)
#pragma once
#include <stdint.h>

namespace vrg {

struct TsCfg { int bw, bh, split, vectorize; };

inline int ts_last_pow2(int n) {
    n |= n >> 1; n |= n >> 2; n |= n >> 4; n |= n >> 8; n |= n >> 16;
    const int r = n - (n >> 1);
    return r > 1 ? r : 1;
}
inline int64_t ts_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// setReduceConfig for iter.ndim() == 2, reduction over the contiguous fastest dimension, warp 64, 512 threads max.
// false: a geometry with ctas_per_output > 1 (not reachable for >= 2 outputs; kept as a guard).  `num_mp` = the device's CU count
// (256 on the MI355X): it enters only through target_grid_size = num_mp * (max_threads_per_mp / block threads), and ROCm's cap of
// max_threads_per_mp = 256 for 2-dim iterators makes that 0 for the 512-thread blocks of every call with more than one output --
// the geometry of the calls this file replays is therefore the same on any CU count (a compute-partitioned MI355X, another gfx950 SKU).
inline bool ts_config(int64_t num_outputs, int64_t n, int vec, TsCfg& c, int num_mp = 256) {
    int64_t dim0 = n;
    c.vectorize = dim0 >= 128;
    if (c.vectorize) dim0 /= vec;
    const int d0 = dim0 < 512 ? ts_last_pow2((int)dim0) : 512;
    const int d1 = num_outputs < 512 ? ts_last_pow2((int)num_outputs) : 512;
    int bw = d0 < 64 ? d0 : 64;
    const int bh = d1 < 512 / bw ? d1 : 512 / bw;
    bw = d0 < 512 / bh ? d0 : 512 / bh;
    int64_t vpt = ts_div_up(n, bw);
    const int thr = bh * 16 < 256 ? bh * 16 : 256;
    c.split = vpt >= thr;
    c.bw = bw; c.bh = bh;
    const int64_t step_in = (int64_t)bw * (c.split ? bh : 1), step_out = c.split ? 1 : bh;
    const int64_t grid_x = ts_div_up(num_outputs, step_out);
    const int max_tpm = grid_x == 1 ? 2048 : 256;          // `grid.x == grid.y == grid.z == 1` as C evaluates it
    const int64_t target = (int64_t)num_mp * (int64_t)(max_tpm / (bw * bh));
    vpt = ts_div_up(n, step_in);
    if (c.split && vpt >= 256 && grid_x <= target) {
        const int64_t c1 = ts_div_up(target, grid_x), c2 = ts_div_up(vpt, 16), c3 = ts_div_up(vpt, 256);
        int64_t ctas = (c1 < c2 ? c1 : c2) > c3 ? (c1 < c2 ? c1 : c2) : c3;
        if (ctas > 256) ctas = 256; else if (ctas > 128) ctas = 128; else if (ctas < 16) ctas = 1;
        if (ctas != 1) return false;
    }
    return true;
}

}  // namespace vrg
