// vrg_collective.hip -- the one exchange step of the path as a C-ABI entry point: the Lab statistics of a reference frame whose
// rows were reduced on different GPUs are combined with two SUM all-reduces over RCCL (xGMI) -- BASELINE config 5.
// The Python host does the same arithmetic through torch.distributed (sharding.allreduce_stats); this entry point is for hosts
// that own an ncclComm_t.  RCCL is resolved at run time, so libvrgdg_hip.so itself links nothing but the HIP runtime.
#include <dlfcn.h>

#include "vrg_common.hpp"

namespace vrg {

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0;      // ncclDataType_t::ncclFloat64, ncclRedOp_t::ncclSum (rccl.h)

static nccl_allreduce_fn resolve_allreduce() {
    static nccl_allreduce_fn fn = []() -> nccl_allreduce_fn {
        // a process that already talks RCCL (torch.distributed's bundled copy, or the system one) has it loaded: bind to THAT
        // copy -- the communicator handed in belongs to it -- and only otherwise load the system library
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))
                if (void* s = dlsym(h, "ncclAllReduce")) return reinterpret_cast<nccl_allreduce_fn>(s);
        }
        if (void* s = dlsym(RTLD_DEFAULT, "ncclAllReduce")) return reinterpret_cast<nccl_allreduce_fn>(s);
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))
                if (void* s = dlsym(h, "ncclAllReduce")) return reinterpret_cast<nccl_allreduce_fn>(s);
        }
        return nullptr;
    }();
    return fn;
}

// triples (n, mean, M2) -> (n, n * mean)
__global__ void k_stats_pack(const double* __restrict__ stats, double* __restrict__ first, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const double n = stats[3 * i], mean = stats[3 * i + 1];
    first[2 * i] = n;
    first[2 * i + 1] = n * mean;
}

// global mean from the summed (n, n * mean); this rank's M2 moved to it: M2 + n * (mean - mean_tot)^2   (Chan et al.)
__global__ void k_stats_shift(const double* __restrict__ stats, const double* __restrict__ first, double* __restrict__ second, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const double n = stats[3 * i], mean = stats[3 * i + 1], m2 = stats[3 * i + 2];
    const double n_tot = first[2 * i];
    const double mean_tot = first[2 * i + 1] / n_tot;
    const double delta = mean - mean_tot;
    const double t = n * delta;
    second[i] = m2 + t * delta;
}

__global__ void k_stats_unpack(double* __restrict__ stats, const double* __restrict__ first, const double* __restrict__ second, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const double n_tot = first[2 * i];
    stats[3 * i] = n_tot;
    stats[3 * i + 1] = first[2 * i + 1] / n_tot;
    stats[3 * i + 2] = second[i];
}

}  // namespace vrg

using namespace vrg;

extern "C" {

int64_t vrg_stats_allreduce_scratch_bytes(int64_t count) { return count < 0 ? 0 : count * 3 * (int64_t)sizeof(double); }

int vrg_stats_allreduce(double* stats, int64_t count, void* comm, void* scratch, void* stream) {
    if (!stats || !comm || !scratch || count < 0) return VRG_ERR_BAD_ARG;
    if (count == 0) return VRG_OK;
    const nccl_allreduce_fn allreduce = resolve_allreduce();
    if (!allreduce) return VRG_ERR_UNSUPPORTED;                 // no RCCL in this process and none installed
    hipStream_t st = (hipStream_t)stream;
    double* first = reinterpret_cast<double*>(scratch);
    double* second = first + 2 * count;
    const uint32_t blocks = (uint32_t)((count + 255) / 256);
    hipLaunchKernelGGL(k_stats_pack, dim3(blocks), dim3(256), 0, st, stats, first, count);
    VRG_CHECK_LAUNCH();
    if (allreduce(first, first, (size_t)(2 * count), NCCL_FLOAT64, NCCL_SUM, comm, st) != 0) return VRG_ERR_LAUNCH;
    hipLaunchKernelGGL(k_stats_shift, dim3(blocks), dim3(256), 0, st, stats, first, second, count);
    VRG_CHECK_LAUNCH();
    if (allreduce(second, second, (size_t)count, NCCL_FLOAT64, NCCL_SUM, comm, st) != 0) return VRG_ERR_LAUNCH;
    hipLaunchKernelGGL(k_stats_unpack, dim3(blocks), dim3(256), 0, st, stats, first, second, count);
    VRG_CHECK_LAUNCH();
    return VRG_OK;
}

}  // extern "C"
