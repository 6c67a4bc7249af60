// VALU issue-rate micro-benchmark for gfx950: decides the big-integer limb strategy
// (SURVEY §7 "first thing to measure on the GPU box: v_mad_u64_u32 issue rate").
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_valu.hip -o tools/ubench_valu
// Prints cycles per wave64 instruction per SIMD (relative to the measured v_add_u32 = 2 cyc
// assumption is NOT made; we print raw Ginstr/s and the ratio to v_add_u32).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../rapidsnark-old_amd/csrc/field.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define ITERS 32768
#define UNROLL 8

#define DEFINE_KERNEL32(name, ASM)                                                      \
    __global__ void name(uint32_t *out, uint32_t seed) {                                    \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;                                  \
        uint32_t r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7; \
        for (int i = 0; i < ITERS; i++) {                                                   \
            asm volatile(ASM : "+v"(r0) : "v"(a), "v"(b));                                  \
            asm volatile(ASM : "+v"(r1) : "v"(a), "v"(b));                                  \
            asm volatile(ASM : "+v"(r2) : "v"(a), "v"(b));                                  \
            asm volatile(ASM : "+v"(r3) : "v"(a), "v"(b));                                  \
            asm volatile(ASM : "+v"(r4) : "v"(a), "v"(b));                                  \
            asm volatile(ASM : "+v"(r5) : "v"(a), "v"(b));                                  \
            asm volatile(ASM : "+v"(r6) : "v"(a), "v"(b));                                  \
            asm volatile(ASM : "+v"(r7) : "v"(a), "v"(b));                                  \
        }                                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7; \
    }

#define DEFINE_KERNEL64(name, ASM, CLOB)                                                \
    __global__ void name(uint32_t *out, uint32_t seed) {                                    \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;                                  \
        uint64_t r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7; \
        for (int i = 0; i < ITERS; i++) {                                                   \
            asm volatile(ASM : "+v"(r0) : "v"(a), "v"(b) : CLOB);                           \
            asm volatile(ASM : "+v"(r1) : "v"(a), "v"(b) : CLOB);                           \
            asm volatile(ASM : "+v"(r2) : "v"(a), "v"(b) : CLOB);                           \
            asm volatile(ASM : "+v"(r3) : "v"(a), "v"(b) : CLOB);                           \
            asm volatile(ASM : "+v"(r4) : "v"(a), "v"(b) : CLOB);                           \
            asm volatile(ASM : "+v"(r5) : "v"(a), "v"(b) : CLOB);                           \
            asm volatile(ASM : "+v"(r6) : "v"(a), "v"(b) : CLOB);                           \
            asm volatile(ASM : "+v"(r7) : "v"(a), "v"(b) : CLOB);                           \
        }                                                                                   \
        uint64_t x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);    \
    }

DEFINE_KERNEL32(k_add_u32, "v_add_u32 %0, %1, %0")
DEFINE_KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
DEFINE_KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %1, %0")
DEFINE_KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEFINE_KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %1, %0")
DEFINE_KERNEL32(k_add3_u32, "v_add3_u32 %0, %1, %2, %0")
DEFINE_KERNEL32(k_mov_b32, "v_mov_b32 %0, %1")
DEFINE_KERNEL32(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
DEFINE_KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0", "vcc")
DEFINE_KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0", "memory")
DEFINE_KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %0, %0", "memory")
DEFINE_KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %0", "memory")

// carry-chain: add_co + 7 addc per "op" (8 instr)
__global__ void k_addc_chain(uint32_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x;
    uint32_t r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    for (int i = 0; i < ITERS; i++) {
        asm volatile(
            "v_add_co_u32 %0, vcc, %8, %0\n v_addc_co_u32 %1, vcc, %8, %1, vcc\n v_addc_co_u32 %2, vcc, %8, %2, vcc\n"
            "v_addc_co_u32 %3, vcc, %8, %3, vcc\n v_addc_co_u32 %4, vcc, %8, %4, vcc\n v_addc_co_u32 %5, vcc, %8, %5, vcc\n"
            "v_addc_co_u32 %6, vcc, %8, %6, vcc\n v_addc_co_u32 %7, vcc, %8, %7, vcc\n"
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

// real modmul chains: NCH independent Montgomery multiplications per thread
template <int NCH>
__global__ void k_modmul(uint32_t *out, uint32_t seed, int iters) {
    zk::Fq x[NCH], y;
    for (int k = 0; k < 8; k++) y.v[k] = seed * (k + 3) + threadIdx.x;
    y.v[7] &= 0x0fffffffu;
    for (int c = 0; c < NCH; c++) { x[c] = y; x[c].v[0] += c; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < NCH; c++) x[c] = zk::Fq::mul(x[c], y);
    }
    uint32_t o = 0;
    for (int c = 0; c < NCH; c++) for (int k = 0; k < 8; k++) o ^= x[c].v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = o;
}
__global__ void k_modadd(uint32_t *out, uint32_t seed, int iters) {
    zk::Fq x, y;
    for (int k = 0; k < 8; k++) y.v[k] = seed * (k + 3) + threadIdx.x;
    y.v[7] &= 0x0fffffffu;
    x = y; x.v[0] += 1;
    for (int i = 0; i < iters; i++) { x = zk::Fq::add(x, y); y = zk::Fq::sub(y, x); }
    uint32_t o = 0;
    for (int k = 0; k < 8; k++) o ^= x.v[k] ^ y.v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = o;
}

template <class K, class... Args>
static double time_kernel(K kern, int blocks, int threads, Args... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, args...);   // warm
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, args...);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    int threads = 256, wavesPerSimd = 8;
    int blocks = cus * wavesPerSimd;     // 256 thr = 4 waves = 1 wave/SIMD per block
    uint32_t *out;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
    double simds = cus * 4.0;
    double base = 0;
#define RUN(name, k, perIter)                                                              \
    {                                                                                      \
        double t = time_kernel(k, blocks, threads, out, 12345u);                           \
        double winstr = (double)blocks * (threads / 64) * ITERS * (perIter);               \
        double rate = winstr / t / simds;  /* wave-instr per second per SIMD */            \
        if (base == 0) base = rate;                                                        \
        printf("%-18s %8.3f ms  %7.3f Gwinstr/s/SIMD  x%.2f of v_add_u32 time (cyc@2.4GHz %.2f)\n", name, t * 1e3, rate * 1e-9, base / rate, 2.4e9 / rate); \
    }
    RUN("v_add_u32", k_add_u32, 8)
    RUN("v_mov_b32", k_mov_b32, 8)
    RUN("v_add3_u32", k_add3_u32, 8)
    RUN("v_fma_f32", k_fma_f32, 8)
    RUN("v_mul_lo_u32", k_mul_lo_u32, 8)
    RUN("v_mul_hi_u32", k_mul_hi_u32, 8)
    RUN("v_mad_u32_u24", k_mad_u32_u24, 8)
    RUN("v_mul_hi_u32_u24", k_mul_hi_u32_u24, 8)
    RUN("v_mad_u64_u32", k_mad_u64_u32, 8)
    RUN("v_lshl_add_u64", k_lshl_add_u64, 8)
    RUN("v_fma_f64", k_fma_f64, 8)
    RUN("v_mul_f64", k_mul_f64, 8)
    RUN("addc chain(8)", k_addc_chain, 8)

    // modmul throughput at several occupancies
    for (int wps = 1; wps <= 8; wps *= 2) {
        int b = cus * wps;
        int iters = 2000;
        double t1 = time_kernel(k_modmul<1>, b, threads, out, 777u, iters);
        double t4 = time_kernel(k_modmul<4>, b, threads, out, 777u, iters);
        double n1 = (double)b * threads * iters, n4 = n1 * 4;
        printf("modmul waves/SIMD=%d: 1-chain %.2f Gmul/s (%.0f cyc/wave-mul/SIMD @2.4GHz)  4-chain %.2f Gmul/s (%.0f cyc)\n", wps,
               n1 / t1 * 1e-9, 2.4e9 * simds * 64 / (n1 / t1), n4 / t4 * 1e-9, 2.4e9 * simds * 64 / (n4 / t4));
    }
    {
        int b = cus * 8, iters = 4000;
        double t = time_kernel(k_modadd, b, threads, out, 777u, iters);
        double n = (double)b * threads * iters * 2;
        printf("modadd/sub: %.2f Gop/s (%.0f cyc/wave-op/SIMD @2.4GHz)\n", n / t * 1e-9, 2.4e9 * simds * 64 / (n / t));
    }
    hipFree(out);
    return 0;
}
