// Error discipline of libzkhip's device code: every HIP runtime call is checked, and every group
// of kernel launches is followed by hipGetLastError() — a bad launch configuration (zero or
// oversized grid, dynamic LDS over the limit) is NOT sticky and would otherwise surface only as
// a wrong result.  Errors travel as exceptions up to the C-ABI, which turns them into a status
// code + zk_last_error() (nothing throws across the boundary).
#pragma once
#include <hip/hip_runtime.h>
#include <stdexcept>
#include <string>

namespace zk {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw HipError(std::string(what) + ": " + hipGetErrorString(e));
}

}   // namespace zk

#define ZK_HIP(expr) ::zk::hip_check((expr), #expr)
// hipGetLastError is a thread-local read: cheap enough to run after every launcher
#define ZK_LAUNCH_OK(what) ::zk::hip_check(hipGetLastError(), "kernel launch (" what ")")
