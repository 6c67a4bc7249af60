// Random-row gather bandwidth on MI355X: rows of 64 B / 128 B from tables of 256 MB .. 10 GB.
// Index stream is read coalesced (4 B per row), rows are summed into a register (no stores).
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_probe tools/gather_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ROW16>   // row size in 16-byte units
__global__ __launch_bounds__(256) void k_gather(const uint4 *table, const uint32_t *idx, uint64_t n, uint32_t per_lane, uint4 *sink) {
    uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = lane * per_lane;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < per_lane && lo + k < n; k++) {
        const uint4 *row = table + (uint64_t)idx[lo + k] * ROW16;
#pragma unroll
        for (int j = 0; j < ROW16; j++) {
            uint4 v = row[j];
            acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
        }
    }
    if (acc.x == 0x12345678u) sink[0] = acc;
}
// same, but consecutive lanes read consecutive index entries (lane-interleaved order)
template <int ROW16>
__global__ __launch_bounds__(256) void k_gather_il(const uint4 *table, const uint32_t *idx, uint64_t n, uint4 *sink) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4 *row = table + (uint64_t)idx[i] * ROW16;
#pragma unroll
        for (int j = 0; j < ROW16; j++) {
            uint4 v = row[j];
            acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
        }
    }
    if (acc.x == 0x12345678u) sink[0] = acc;
}

int main() {
    const uint64_t n = 54525952;     // 13 * 2^22 gathers, as one precomputed-table MSM at 2^22
    std::vector<uint32_t> h(n);
    uint32_t *d_idx; uint4 *d_sink; uint4 *d_table;
    const uint64_t max_bytes = 10ull << 30;
    CK(hipMalloc(&d_idx, n * 4)); CK(hipMalloc(&d_sink, 64)); CK(hipMalloc(&d_table, max_bytes));
    CK(hipMemset(d_table, 1, max_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int row = 64; row <= 128; row *= 2) {
        // 3584 MB ~ one window-precomputed G1 table at 2^22 (3.25 GiB), 6656 MB = the G2 table (6.5 GiB), 10240 MB ~ the three
        // G1 tables the A|B1|C launch gathers from at once (9.75 GiB)
        for (uint64_t mb : {256ull, 1024ull, 3584ull, 6656ull, 7168ull, 10240ull}) {
            uint64_t rows = (mb << 20) / row;
            uint64_t x = 88172645463325252ull;
            for (uint64_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x % rows); }
            CK(hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice));
            for (int mode = 0; mode < 3; mode++) {
                float best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) {          // lane walks 128 consecutive entries (the accumulate kernel's pattern)
                        uint32_t per = 128; uint64_t lanes = (n + per - 1) / per;
                        if (row == 64) hipLaunchKernelGGL(k_gather<4>, dim3((lanes + 255) / 256), dim3(256), 0, 0, d_table, d_idx, n, per, d_sink);
                        else hipLaunchKernelGGL(k_gather<8>, dim3((lanes + 255) / 256), dim3(256), 0, 0, d_table, d_idx, n, per, d_sink);
                    } else if (mode == 1) {   // many independent gathers in flight, full occupancy
                        if (row == 64) hipLaunchKernelGGL(k_gather_il<4>, dim3(256 * 8), dim3(256), 0, 0, d_table, d_idx, n, d_sink);
                        else hipLaunchKernelGGL(k_gather_il<8>, dim3(256 * 8), dim3(256), 0, 0, d_table, d_idx, n, d_sink);
                    } else {                  // 3 waves/SIMD only (the G1 accumulate occupancy)
                        if (row == 64) hipLaunchKernelGGL(k_gather_il<4>, dim3(256 * 3), dim3(256), 0, 0, d_table, d_idx, n, d_sink);
                        else hipLaunchKernelGGL(k_gather_il<8>, dim3(256 * 3), dim3(256), 0, 0, d_table, d_idx, n, d_sink);
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("row %3d B table %5llu MB mode %d: %7.3f ms  %7.1f GB/s  %6.2f G rows/s\n", row, (unsigned long long)mb, mode, best,
                       (double)n * row / best / 1e6, (double)n / best / 1e6);
            }
        }
    }
    return 0;
}
