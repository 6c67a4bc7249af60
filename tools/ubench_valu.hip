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

#define DEFINE_KERNEL64(name, ASM, ...)                                                \
    __global__ void name(uint32_t *out, uint32_t seed) {                                    \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;                                  \
        uint64_t r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7; \
        for (int i = 0; i < ITERS; i++) {                                                   \
            asm volatile(ASM : "+v"(r0) : "v"(a), "v"(b) : __VA_ARGS__);                           \
            asm volatile(ASM : "+v"(r1) : "v"(a), "v"(b) : __VA_ARGS__);                           \
            asm volatile(ASM : "+v"(r2) : "v"(a), "v"(b) : __VA_ARGS__);                           \
            asm volatile(ASM : "+v"(r3) : "v"(a), "v"(b) : __VA_ARGS__);                           \
            asm volatile(ASM : "+v"(r4) : "v"(a), "v"(b) : __VA_ARGS__);                           \
            asm volatile(ASM : "+v"(r5) : "v"(a), "v"(b) : __VA_ARGS__);                           \
            asm volatile(ASM : "+v"(r6) : "v"(a), "v"(b) : __VA_ARGS__);                           \
            asm volatile(ASM : "+v"(r7) : "v"(a), "v"(b) : __VA_ARGS__);                           \
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
DEFINE_KERNEL64(k_mad_i64_i32, "v_mad_i64_i32 %0, vcc, %1, %2, %0", "vcc")
DEFINE_KERNEL64(k_ashrrev_i64, "v_ashrrev_i64 %0, 29, %0", "memory")
DEFINE_KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0", "memory")
DEFINE_KERNEL64(k_mov_b64, "v_mov_b64 %0, %0", "memory")
DEFINE_KERNEL32(k_alignbit_b32, "v_alignbit_b32 %0, %1, %0, 29")
DEFINE_KERNEL32(k_ashrrev_i32, "v_ashrrev_i32 %0, 29, %0")
DEFINE_KERNEL32(k_and_b32, "v_and_b32 %0, %1, %0")
DEFINE_KERNEL32(k_and_lit, "v_and_b32 %0, 0x1fffffff, %0")
DEFINE_KERNEL32(k_sub_u32, "v_sub_u32 %0, %1, %0")
DEFINE_KERNEL32(k_bfe_i32, "v_bfe_i32 %0, %0, 0, 29")
DEFINE_KERNEL32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
DEFINE_KERNEL32(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
DEFINE_KERNEL32(k_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
DEFINE_KERNEL32(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
DEFINE_KERNEL32(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
// (the plain `v_cndmask_b32 ..., vcc` lines of this probe read 16-23 cycles per instruction; that is this probe's own loop — a proof-level
// A/B of the G2 level-1 kernel with its 63 in-loop v_cndmask_b32_e32 replaced by xor / sub / and on an opaque lane mask moved its
// launch 11.33 -> 11.29 ms, profiles/r04aw_ab_fq2_without_cndmask.txt: in compiled code the instruction costs what the others do)
// v_cndmask with a lane mask that is really there: e64 form on an SGPR pair, and the e32 form on vcc written once in front of the loop
__global__ void k_cndmask_e64(uint32_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t m = __ballot((threadIdx.x & 1u) != 0);
    uint32_t r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r0) : "v"(a), "s"(m));
        asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r1) : "v"(b), "s"(m));
        asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r2) : "v"(a), "s"(m));
        asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r3) : "v"(b), "s"(m));
        asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r4) : "v"(a), "s"(m));
        asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r5) : "v"(b), "s"(m));
        asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r6) : "v"(a), "s"(m));
        asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r7) : "v"(b), "s"(m));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}
__global__ void k_cndmask_vcc(uint32_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint32_t r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_cmp_lt_u32 vcc, %8, %9\n s_nop 4\n"
                     "v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %9, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %9, %3, vcc\n"
                     "v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %9, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %9, %7, vcc\n"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}
DEFINE_KERNEL32(k_bfi_b32, "v_bfi_b32 %0, %1, %2, %0")
DEFINE_KERNEL32(k_xor_b32, "v_xor_b32 %0, %1, %0")
DEFINE_KERNEL32(k_or_b32, "v_or_b32 %0, %1, %0")
DEFINE_KERNEL32(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
DEFINE_KERNEL32(k_add_co, "v_add_co_u32 %0, vcc, %1, %0")
DEFINE_KERNEL64(k_mad_i64_sgpr, "v_mad_i64_i32 %0, s[20:21], %1, %2, %0", "s20", "s21")
DEFINE_KERNEL64(k_mad_i64_zero, "v_mad_i64_i32 %0, vcc, %1, %2, 0", "vcc")
// two accumulators fed alternately by each statement: the chain of one never issues back to back
__global__ void k_mad_i64_pairs(uint32_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    for (int i = 0; i < ITERS / 2; i++) {
        asm volatile("v_mad_i64_i32 %0, s[20:21], %8, %9, %0\n v_mad_i64_i32 %1, s[22:23], %8, %9, %1\n v_mad_i64_i32 %2, s[20:21], %8, %9, %2\n v_mad_i64_i32 %3, s[22:23], %8, %9, %3\n"
                     "v_mad_i64_i32 %4, s[20:21], %8, %9, %4\n v_mad_i64_i32 %5, s[22:23], %8, %9, %5\n v_mad_i64_i32 %6, s[20:21], %8, %9, %6\n v_mad_i64_i32 %7, s[22:23], %8, %9, %7\n"
                     "v_mad_i64_i32 %0, s[20:21], %8, %9, %0\n v_mad_i64_i32 %1, s[22:23], %8, %9, %1\n v_mad_i64_i32 %2, s[20:21], %8, %9, %2\n v_mad_i64_i32 %3, s[22:23], %8, %9, %3\n"
                     "v_mad_i64_i32 %4, s[20:21], %8, %9, %4\n v_mad_i64_i32 %5, s[22:23], %8, %9, %5\n v_mad_i64_i32 %6, s[20:21], %8, %9, %6\n v_mad_i64_i32 %7, s[22:23], %8, %9, %7\n"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23");
    }
    uint64_t x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);
}
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
    RUN("v_mad_i64_i32", k_mad_i64_i32, 8)
    RUN("v_mad_i64_i32 -> sgpr pair", k_mad_i64_sgpr, 8)
    RUN("v_mad_i64_i32 addend 0", k_mad_i64_zero, 8)
    RUN("v_mad_i64_i32 block of 16", k_mad_i64_pairs, 8)
    RUN("v_ashrrev_i64", k_ashrrev_i64, 8)
    RUN("v_lshlrev_b64", k_lshlrev_b64, 8)
    RUN("v_mov_b64", k_mov_b64, 8)
    RUN("v_alignbit_b32", k_alignbit_b32, 8)
    RUN("v_ashrrev_i32", k_ashrrev_i32, 8)
    RUN("v_and_b32", k_and_b32, 8)
    RUN("v_and_b32 literal", k_and_lit, 8)
    RUN("v_sub_u32", k_sub_u32, 8)
    RUN("v_bfe_i32", k_bfe_i32, 8)
    RUN("v_lshl_add_u32", k_lshl_add_u32, 8)
    RUN("v_and_or_b32", k_and_or_b32, 8)
    RUN("v_lshlrev_b32", k_lshlrev_b32, 8)
    RUN("v_cndmask_b32", k_cndmask, 8)
    RUN("v_mov_b32 dpp", k_mov_dpp, 8)
    RUN("v_cndmask e64 sgpr", k_cndmask_e64, 8)
    RUN("v_cndmask e32 vcc", k_cndmask_vcc, 8)
    RUN("v_bfi_b32", k_bfi_b32, 8)
    RUN("v_xor_b32", k_xor_b32, 8)
    RUN("v_or_b32", k_or_b32, 8)
    RUN("v_lshrrev_b32", k_lshrrev_b32, 8)
    RUN("v_add_co_u32", k_add_co, 8)
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
