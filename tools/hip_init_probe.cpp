// What a fresh process pays before its first kernel runs (the one-shot CLI pays it for every proof):
//   hipcc --offload-arch=gfx950 -O2 -o tools/hip_init_probe tools/hip_init_probe.cpp && tools/hip_init_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int *p) { if (p) *p = 1; }
int main(int argc, char **argv) {
    const bool parallel = argc > 1;
    double t = now_ms(), t0 = t;
    auto lap = [&](const char *what) { double n = now_ms(); printf("%-44s %8.2f ms\n", what, n - t); t = n; };
    int n = 0;
    hipGetDeviceCount(&n); lap("hipGetDeviceCount (runtime init)");
    hipSetDevice(0); lap("hipSetDevice");
    void *p = nullptr;
    hipMalloc(&p, 1 << 20); lap("first hipMalloc (context)");
    hipStream_t s[12];
    if (!parallel) {
        for (int i = 0; i < 12; i++) {
            hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
            char b[64]; snprintf(b, sizeof b, "hipStreamCreate #%d", i); lap(b);
        }
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < 12; i++) th.emplace_back([&, i] { hipSetDevice(0); hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); });
        for (auto &x : th) x.join();
        lap("12 x hipStreamCreate on 12 threads");
    }
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s[0], (int *)p); hipStreamSynchronize(s[0]); lap("first kernel launch + sync (code object load)");
    for (int i = 1; i < 12; i++) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s[i], (int *)p); hipStreamSynchronize(s[i]); }
    lap("first launch on the other 11 streams");
    void *h = nullptr;
    hipHostMalloc(&h, 128 << 20, hipHostMallocDefault); lap("hipHostMalloc 128 MiB");
    void *big = nullptr;
    hipMalloc(&big, (size_t)4 << 30); lap("hipMalloc 4 GiB");
    hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); lap("hipEventCreate");
    printf("%-44s %8.2f ms\n", "total", now_ms() - t0);
    return 0;
}
