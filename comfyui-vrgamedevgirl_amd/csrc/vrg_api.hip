// vrg_api.hip -- introspection and HIP-event helpers of the C ABI.
#include "vrg_common.hpp"

extern "C" {

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
