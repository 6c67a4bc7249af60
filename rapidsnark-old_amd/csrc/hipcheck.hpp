// Error discipline of libzkhip's device code: every HIP runtime call is checked, and every group
// of kernel launches is followed by hipGetLastError() — a bad launch configuration (zero or
// oversized grid, dynamic LDS over the limit) is NOT sticky and would otherwise surface only as
// a wrong result.  Errors travel as exceptions up to the C-ABI, which turns them into a status
// code + zk_last_error() (nothing throws across the boundary).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdexcept>
#include <string>

namespace zk {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw HipError(std::string(what) + ": " + hipGetErrorString(e));
}

// Once per DEVICE: a function attribute set through hipFuncSetAttribute belongs to the current device's copy of the
// kernel, and one process may drive eight devices (zk_multi_prover, the server's replicas).  need() is true until
// done() has run on the calling thread's current device; two threads racing on one device both set the attribute.
struct PerDeviceOnce {
    std::atomic<uint64_t> mask{0};
    static uint64_t bit() {
        int d = 0;
        hip_check(hipGetDevice(&d), "hipGetDevice");
        return 1ull << (d & 63);
    }
    bool need() const { return !(mask.load(std::memory_order_acquire) & bit()); }
    void done() { mask.fetch_or(bit(), std::memory_order_release); }
};

}   // namespace zk

// Every kernel launch of the library goes through ZK_LAUNCH: one process-wide relaxed counter, so that zk_prover_info can say how
// many launches the last proof of a prover took (the launch count of a rank's share is what bounds a sharded proof at small
// shares).
namespace zk {
inline std::atomic<uint64_t> g_kernel_launches{0};
}
#define ZK_LAUNCH(...)                                                        \
    do {                                                                      \
        ::zk::g_kernel_launches.fetch_add(1, std::memory_order_relaxed);      \
        hipLaunchKernelGGL(__VA_ARGS__);                                      \
    } while (0)

#define ZK_HIP(expr) ::zk::hip_check((expr), #expr)
// hipGetLastError is a thread-local read: cheap enough to run after every launcher
#define ZK_LAUNCH_OK(what) ::zk::hip_check(hipGetLastError(), "kernel launch (" what ")")
