// vrg_api.hip -- introspection and HIP-event helpers of the C ABI.
#include "vrg_common.hpp"

namespace vrg {

// dev_pow_ziv exactly as the three call sites of the Lab transforms call it (csrc/vrg_pixel_math.hpp: srgb -> linear, linear -> srgb, the Lab
// cube root -- exponent, domain and flavour of each), one value per thread: what ops.toolchain_selfcheck compares with this ROCm's powf
// (torch.pow) on the device at first use.
__global__ __launch_bounds__(256) void k_selfcheck_pow(const float* __restrict__ in, float* __restrict__ out, int64_t n, int site, DevMath dm) {
    __shared__ __attribute__((aligned(16))) float zivt[ZIV_TABLE_WORDS];
    ziv_table_fill(zivt, (int)threadIdx.x, 256);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    if (site == 0) out[i] = dev_pow_ziv<DEV_POW_OVF>(x, dm.e24, zivt, 0x3d800000u, 0x40000000u);
    else if (site == 1) out[i] = dev_pow_ziv<DEV_POW_UNIT>(x, dm.e1_24, zivt, 0x3b4d2e1cu, 0x40800000u);
    else out[i] = dev_pow_ziv<DEV_POW_UNIT>(x, dm.e1_3, zivt, 0x3c1118c2u, 0x40800000u);
}

}  // namespace vrg

extern "C" {

int vrg_selfcheck_pow_f32(const float* in, float* out, int64_t n, int32_t site, void* stream) {
    if (!in || !out || n <= 0 || site < 0 || site > 2) return VRG_ERR_BAD_ARG;
    const uint64_t blocks = (uint64_t)(n + 255) / 256;
    if (blocks > 0x7fffffffull) return VRG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(vrg::k_selfcheck_pow, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, in, out, n, site, vrg::host_dev_math());
    VRG_CHECK_LAUNCH();
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
