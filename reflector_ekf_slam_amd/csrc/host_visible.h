// host_visible.h -- fine-grained DEVICE memory the host writes directly (PCIe BAR), for the detectors' input buffers.
//
// hipExtMallocWithFlags(hipDeviceMallocFinegrained) returns device memory that the host can address on large-BAR platforms
// (this MI355X: a 28.8 KB memcpy costs 0.4 us of posted writes, scripts/probe/bar_write.hip).  Whether the BAR covers device
// memory is a device property (hipDeviceAttributeIsLargeBar): where it does not, where the allocation is refused, or with
// RDET_NO_BAR set in the environment, the caller falls back to pinned host memory + a copy.  (Rounds 1-2 probed the first word
// under a temporary SIGSEGV / SIGBUS handler instead: process-wide handlers and a siglongjmp out of them are not safe in a
// multi-threaded host such as a ROS node.)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

namespace host_visible {

// device memory of `bytes` that the host may memcpy into, or nullptr (then use a pinned staging buffer and hipMemcpyAsync)
inline void *alloc(size_t bytes)
{
    if (std::getenv("RDET_NO_BAR")) return nullptr;
    int dev = 0, large_bar = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess || !large_bar) {
        (void)hipGetLastError();
        return nullptr;
    }
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

}  // namespace host_visible
