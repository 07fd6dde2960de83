// host_visible.h -- fine-grained DEVICE memory the host writes directly (PCIe BAR), for the detectors' input buffers.
//
// hipExtMallocWithFlags(hipDeviceMallocFinegrained) returns device memory that the host can address on large-BAR platforms
// (this MI355X: a 28.8 KB memcpy costs 0.4 us of posted writes, scripts/probe/bar_write.hip).  Where the BAR does not cover the
// allocation the first host access would fault, so the first word is probed once under a SIGSEGV / SIGBUS guard; on a fault, a
// refused allocation, or RDET_NO_BAR set in the environment the caller falls back to pinned host memory + a copy.
#pragma once
#include <hip/hip_runtime.h>

#include <csetjmp>
#include <csignal>
#include <cstdlib>

namespace host_visible {

inline sigjmp_buf &jump() { static thread_local sigjmp_buf j; return j; }
inline void on_fault(int) { siglongjmp(jump(), 1); }

// true if the host can write and read back p[0]
inline bool probe(volatile unsigned *p)
{
    struct sigaction sa = {}, old_segv = {}, old_bus = {};
    sa.sa_handler = on_fault;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, &sa, &old_bus);
    bool ok = false;
    if (sigsetjmp(jump(), 1) == 0) {
        p[0] = 0x5eaf00du;
        ok = p[0] == 0x5eaf00du;
        p[0] = 0;
    }
    sigaction(SIGSEGV, &old_segv, nullptr);
    sigaction(SIGBUS, &old_bus, nullptr);
    return ok;
}

// device memory of `bytes` that the host may memcpy into, or nullptr (then use a pinned staging buffer and hipMemcpyAsync)
inline void *alloc(size_t bytes)
{
    if (std::getenv("RDET_NO_BAR")) return nullptr;
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (!probe((volatile unsigned *)p)) { (void)hipFree(p); return nullptr; }
    return p;
}

}  // namespace host_visible
