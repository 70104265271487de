// Host side of the node path (include/vrgdg_hip.h: vrg_host_copy): pageable frames -> page-locked staging with several host threads.
// The reference's nodes take CPU tensors (nodes.py:50, 61-66: `images.to(device)` per batch); what ComfyUI hands them is pageable.  The HIP
// runtime's own pageable copy runs after everything queued on the device has drained (measured: every upload starts when the previous
// piece's download ends, profiles/r04_host_fed_timeline_runtime_pageable.json), so upload, kernels and download of a pageable batch
// serialise at 28 GB/s each way.  Copying the piece into a page-locked ring on the host instead leaves all three asynchronous; one thread
// moves ~10 GB/s, so the copy is split over several.  No device code here.
#include "vrg_common.hpp"
#include <cstring>
#include <thread>
#include <vector>

extern "C" int vrg_host_copy(void* dst, const void* src, int64_t bytes, int32_t threads) {
    if (bytes < 0 || threads < 0 || (bytes > 0 && (!dst || !src))) return VRG_ERR_BAD_ARG;
    if (bytes == 0 || dst == src) return VRG_OK;
    constexpr int64_t kGrain = int64_t(1) << 21;                 // no part below 2 MiB: thread start-up is ~20 us
    int64_t parts = threads == 0 ? 8 : threads;
    if (parts > 64) parts = 64;
    const int64_t most = (bytes + kGrain - 1) / kGrain;
    if (parts > most) parts = most;
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    if (parts <= 1) { std::memcpy(d, s, size_t(bytes)); return VRG_OK; }
    const int64_t per = ((bytes + parts - 1) / parts + 4095) & ~int64_t(4095);      // page-sized parts: no two threads share a destination page
    std::vector<std::thread> pool;
    pool.reserve(size_t(parts - 1));
    try {
        for (int64_t p = 1; p < parts; ++p) {
            const int64_t off = p * per;
            if (off >= bytes) break;
            const int64_t len = bytes - off < per ? bytes - off : per;
            pool.emplace_back([=] { std::memcpy(d + off, s + off, size_t(len)); });
        }
    } catch (...) {                                              // the host refused another thread: this one finishes the rest
        for (auto& t : pool) t.join();
        const int64_t done = int64_t(pool.size() + 1) * per;
        std::memcpy(d, s, size_t(per < bytes ? per : bytes));
        if (done < bytes) std::memcpy(d + done, s + done, size_t(bytes - done));
        return VRG_OK;
    }
    std::memcpy(d, s, size_t(per < bytes ? per : bytes));
    for (auto& t : pool) t.join();
    return VRG_OK;
}
