// Does the scalar destination of v_mad_i64_i32 (the unused carry-out) cost issue cycles?
// hipcc gives every MAD of the field kernels the same dead pair (s[0:1]); this probe times 8 independent
// accumulator chains with (a) one fixed pair, (b) four pairs in rotation, (c) vcc, against v_mul_lo_u32
// (a plain quarter-rate instruction) and v_add_u32.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mad_sdst_probe.hip -o tools/mad_sdst_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 32768

#define BODY64(A0, A1, A2, A3)                                                             \
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;                                     \
    uint64_t r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7; \
    for (int i = 0; i < ITERS; i++) {                                                      \
        asm volatile(A0 : "+v"(r0) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc"); \
        asm volatile(A1 : "+v"(r1) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc"); \
        asm volatile(A2 : "+v"(r2) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc"); \
        asm volatile(A3 : "+v"(r3) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc"); \
        asm volatile(A0 : "+v"(r4) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc"); \
        asm volatile(A1 : "+v"(r5) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc"); \
        asm volatile(A2 : "+v"(r6) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc"); \
        asm volatile(A3 : "+v"(r7) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc"); \
    }                                                                                      \
    uint64_t x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);

__global__ void k_fixed(uint32_t *out, uint32_t seed) {
    BODY64("v_mad_i64_i32 %0, s[20:21], %1, %2, %0", "v_mad_i64_i32 %0, s[20:21], %1, %2, %0", "v_mad_i64_i32 %0, s[20:21], %1, %2, %0", "v_mad_i64_i32 %0, s[20:21], %1, %2, %0")
}
__global__ void k_rot(uint32_t *out, uint32_t seed) {
    BODY64("v_mad_i64_i32 %0, s[20:21], %1, %2, %0", "v_mad_i64_i32 %0, s[22:23], %1, %2, %0", "v_mad_i64_i32 %0, s[24:25], %1, %2, %0", "v_mad_i64_i32 %0, s[26:27], %1, %2, %0")
}
__global__ void k_vcc(uint32_t *out, uint32_t seed) {
    BODY64("v_mad_i64_i32 %0, vcc, %1, %2, %0", "v_mad_i64_i32 %0, vcc, %1, %2, %0", "v_mad_i64_i32 %0, vcc, %1, %2, %0", "v_mad_i64_i32 %0, vcc, %1, %2, %0")
}
__global__ void k_lshl_add(uint32_t *out, uint32_t seed) {
    BODY64("v_lshl_add_u64 %0, %0, 0, %0", "v_lshl_add_u64 %0, %0, 0, %0", "v_lshl_add_u64 %0, %0, 0, %0", "v_lshl_add_u64 %0, %0, 0, %0")
}
__global__ void k_ashr64(uint32_t *out, uint32_t seed) {
    BODY64("v_ashrrev_i64 %0, 1, %0", "v_ashrrev_i64 %0, 1, %0", "v_ashrrev_i64 %0, 1, %0", "v_ashrrev_i64 %0, 1, %0")
}
#define BODY32(A)                                                                          \
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;                                     \
    uint32_t r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7; \
    for (int i = 0; i < ITERS; i++) {                                                      \
        asm volatile(A : "+v"(r0) : "v"(a), "v"(b)); asm volatile(A : "+v"(r1) : "v"(a), "v"(b)); \
        asm volatile(A : "+v"(r2) : "v"(a), "v"(b)); asm volatile(A : "+v"(r3) : "v"(a), "v"(b)); \
        asm volatile(A : "+v"(r4) : "v"(a), "v"(b)); asm volatile(A : "+v"(r5) : "v"(a), "v"(b)); \
        asm volatile(A : "+v"(r6) : "v"(a), "v"(b)); asm volatile(A : "+v"(r7) : "v"(a), "v"(b)); \
    }                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
__global__ void k_mul_lo(uint32_t *out, uint32_t seed) { BODY32("v_mul_lo_u32 %0, %1, %0") }
__global__ void k_add(uint32_t *out, uint32_t seed) { BODY32("v_add_u32 %0, %1, %0") }
__global__ void k_and(uint32_t *out, uint32_t seed) { BODY32("v_and_b32 %0, %1, %0") }

template <class K>
static double run(K k, int blocks, uint32_t *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    for (int wps : {1, 3}) {
        int blocks = prop.multiProcessorCount * wps;
        uint32_t *out;
        hipMalloc(&out, (size_t)blocks * 256 * 4);
        double tadd = run(k_add, blocks, out);
        printf("waves/SIMD %d (times relative to v_add_u32 = 2 cycles per wave64)\n", wps);
#define R(name, k) { double t = run(k, blocks, out); printf("  %-34s %8.3f ms   %.2f cycles\n", name, t, 2.0 * t / tadd); }
        R("v_add_u32", k_add)
        R("v_and_b32", k_and)
        R("v_mul_lo_u32", k_mul_lo)
        R("v_lshl_add_u64", k_lshl_add)
        R("v_ashrrev_i64", k_ashr64)
        R("v_mad_i64_i32, sdst one fixed pair", k_fixed)
        R("v_mad_i64_i32, sdst four pairs", k_rot)
        R("v_mad_i64_i32, sdst vcc", k_vcc)
        hipFree(out);
    }
    return 0;
}
