// Does a SERIAL v_mad_i64_i32 chain (carry of column k as the addend of column k+1's first MAD: no 64-bit
// add per column) beat what hipcc makes of field29.hpp's mul (independent column chains + one
// v_lshl_add_u64 per column)?  tools/, not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <random>
#include "../rapidsnark-old_amd/csrc/curve29.hpp"
using namespace zk;

__device__ __forceinline__ void mad(int64_t &acc, int32_t a, int32_t b) {
    uint64_t cy;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "v"(b));
}
__device__ __forceinline__ void madk(int64_t &acc, int32_t a, int32_t k) {       // k: compile-time constant -> SGPR / literal
    uint64_t cy;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "s"(k));
}

__device__ __forceinline__ Fq29 mul_serial(const Fq29 &a, const Fq29 &b) {
    int64_t acc = 0;
    int32_t m[9];
    Fq29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) madk(acc, m[i], Fq29::P[k - i]);
        m[k] = (int32_t)(((uint32_t)acc * Fq29::N0INV) & (uint32_t)Fq29::MASK);
        madk(acc, m[k], Fq29::P[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 8; i <= 8; i++) madk(acc, m[i], Fq29::P[k - i]);
        r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)Fq29::MASK);
        acc >>= 29;
    }
    r.l[8] = (int32_t)acc;
    return r;
}

template <int NCH, bool SERIAL>
__global__ void k_mul(unsigned *out, const Fq *a, int iters) {
    Fq29 y = Fq29::load(a[threadIdx.x & 255]), x[NCH];
    for (int c = 0; c < NCH; c++) { x[c] = y; x[c].l[0] ^= c; }
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int c = 0; c < NCH; c++) x[c] = SERIAL ? mul_serial(x[c], y) : Fq29::mul(x[c], y);
    unsigned o = 0;
    for (int c = 0; c < NCH; c++) for (int k = 0; k < 9; k++) o ^= (unsigned)x[c].l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = o;
}
__global__ void k_check(const Fq *a, const Fq *b, int n, unsigned *bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fq29 X = Fq29::from_mont256(a[i]), Y = Fq29::from_mont256(b[i]);
    Fq29 D = Fq29::sub_nc(X, Y);                     // signed, lazy operand
    if (!(Fq29::to_mont256(mul_serial(X, Y)) == Fq29::to_mont256(Fq29::mul(X, Y)))) atomicOr(bad, 1u);
    if (!(Fq29::to_mont256(mul_serial(D, Y)) == Fq29::to_mont256(Fq29::mul(D, Y)))) atomicOr(bad, 2u);
}

template <class K, class... A>
static double timeit(K k, int blocks, int threads, A... args) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, args...); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, args...); hipEventRecord(e1, 0);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main() {
    const int n = 1 << 16;
    std::mt19937_64 rng(7);
    const uint64_t q[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    std::vector<uint64_t> ha(n * 4), hb(n * 4);
    auto gen = [&](uint64_t *o) { for (;;) { for (int k = 0; k < 4; k++) o[k] = rng(); o[3] &= 0x3fffffffffffffffull; if (o[3] < q[3]) return; } };
    for (int i = 0; i < n; i++) { gen(&ha[i * 4]); gen(&hb[i * 4]); }
    Fq *da, *db; unsigned *dbad, *dout;
    hipMalloc(&da, n * 32); hipMalloc(&db, n * 32); hipMalloc(&dbad, 4);
    hipMemcpy(da, ha.data(), n * 32, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n * 32, hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 4);
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, da, db, n, dbad);
    unsigned bad; hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    printf("serial-chain product vs field29 mul: flags 0x%x (%s)\n", bad, bad ? "FAIL" : "identical on 65536 inputs");
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    hipMalloc(&dout, (size_t)cus * 8 * 256 * 4);
    for (int wps = 1; wps <= 4; wps++) {
        int b = cus * wps, iters = 2000;
        double nm = (double)b * 256 * iters;
        double t0 = timeit(k_mul<1, false>, b, 256, dout, da, iters), t1 = timeit(k_mul<1, true>, b, 256, dout, da, iters);
        double t0_3 = timeit(k_mul<3, false>, b, 256, dout, da, iters), t1_3 = timeit(k_mul<3, true>, b, 256, dout, da, iters);
        printf("waves/SIMD=%d: compiler %.1f | serial %.1f Gmul/s   (3 independent products per lane: compiler %.1f | serial %.1f)\n", wps,
               nm / t0 * 1e-9, nm / t1 * 1e-9, 3 * nm / t0_3 * 1e-9, 3 * nm / t1_3 * 1e-9);
    }
    return 0;
}
