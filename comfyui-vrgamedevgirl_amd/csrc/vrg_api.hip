// vrg_api.hip -- introspection and HIP-event helpers of the C ABI.
#include "vrg_common.hpp"

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

}  // namespace vrg

extern "C" {

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

int vrg_abi_version(void) { return VRG_ABI_VERSION; }

const char* vrg_error_string(int status) {
    switch (status) {
        case VRG_OK: return "ok";
        case VRG_ERR_BAD_ARG: return "invalid argument";
        case VRG_ERR_UNSUPPORTED: return "unsupported configuration";
        case VRG_ERR_LAUNCH: return "HIP kernel launch failed";
        case VRG_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

int vrg_device_info(int32_t* cu_count, int32_t* max_threads_per_cu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VRG_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return VRG_ERR_NO_DEVICE;
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (max_threads_per_cu) *max_threads_per_cu = prop.maxThreadsPerMultiProcessor;
    return VRG_OK;
}

int vrg_event_create(void** ev) {
    if (!ev) return VRG_ERR_BAD_ARG;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return VRG_ERR_NO_DEVICE;
    *ev = (void*)e;
    return VRG_OK;
}

int vrg_event_record(void* ev, void* stream) {
    return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH;
}

int vrg_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!ms) return VRG_ERR_BAD_ARG;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return VRG_ERR_LAUNCH;
    return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH;
}

int vrg_event_destroy(void* ev) { return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? VRG_OK : VRG_ERR_LAUNCH; }

}  // extern "C"
