// Debug probe: where/when do the workgroups of k_msm_accum run?  (tools/, not product code)
#include "../rapidsnark-old_amd/csrc/msm.hip"
#include <stdio.h>
#include <vector>
#include <random>
using namespace zk;

template <class F>
__global__ __launch_bounds__(256) void k_accum_logged(XYZZ<F> *buckets, const uint32_t *offsets, const uint32_t *entries,
                                                      const Affine<F> *points, uint32_t total, uint64_t *log) {
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t w0 = wall_clock64();
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < total) {
        uint32_t e = offsets[b];
        const uint32_t end = offsets[b + 1];
        XYZZ<F> acc = XYZZ<F>::inf();
        while (e < end) {
            uint32_t ent = entries[e++];
            Affine<F> P = load_affine(points + (ent & 0x7fffffffu));
            if (ent >> 31) P.y = F::neg(P.y);
            madd(acc, P);
        }
        store_xyzz(buckets + b, acc);
    }
    if (threadIdx.x == 0) {
        log[blockIdx.x * 4 + 0] = w0;
        log[blockIdx.x * 4 + 1] = wall_clock64();
        log[blockIdx.x * 4 + 2] = ((uint64_t)xcc << 32) | hwid;
        log[blockIdx.x * 4 + 3] = __builtin_readcyclecounter() - t0;
    }
}

int main(int argc, char **argv) {
    int nb = argc > 1 ? atoi(argv[1]) : 13312;     // buckets (threads)
    int per = argc > 2 ? atoi(argv[2]) : 128;      // entries per bucket
    int npts = 1 << 16;
    std::mt19937_64 rng(1);
    std::vector<uint32_t> off(nb + 1), ent((size_t)nb * per);
    for (int i = 0; i <= nb; i++) off[i] = i * per;
    for (auto &e : ent) e = rng() % npts;
    std::vector<uint32_t> pts((size_t)npts * 16);
    for (auto &p : pts) p = (uint32_t)rng() & 0x0fffffffu;
    uint32_t *d_off, *d_ent; G1Affine *d_pts; G1XYZZ *d_b; uint64_t *d_log;
    int blocks = (nb + 255) / 256;
    hipMalloc(&d_off, off.size() * 4); hipMalloc(&d_ent, ent.size() * 4); hipMalloc(&d_pts, pts.size() * 4);
    hipMalloc(&d_b, (size_t)nb * sizeof(G1XYZZ)); hipMalloc(&d_log, blocks * 32);
    hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ent, ent.data(), ent.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_pts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_msm_accum<Fq>, dim3(blocks), dim3(256), 0, 0, d_b, d_off, d_ent, d_pts, 0u, 0u, (uint32_t)nb);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("k_msm_accum<Fq>  buckets=%d per=%d : %.3f ms\n", nb, per, ms);
    }
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_accum_logged<Fq>, dim3(blocks), dim3(256), 0, 0, d_b, d_off, d_ent, d_pts, (uint32_t)nb, d_log);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k_accum_logged  : %.3f ms\n", ms);
    std::vector<uint64_t> lg(blocks * 4);
    hipMemcpy(lg.data(), d_log, blocks * 32, hipMemcpyDeviceToHost);
    uint64_t tmin = ~0ull;
    for (int i = 0; i < blocks; i++) tmin = lg[i * 4] < tmin ? lg[i * 4] : tmin;
    for (int i = 0; i < blocks && i < 60; i++) {
        uint32_t hw = (uint32_t)lg[i * 4 + 2], xcc = (uint32_t)(lg[i * 4 + 2] >> 32);
        printf("wg %3d xcc %u se %u cu %u simd %u  start %8.1f us  end %8.1f us  cycles %llu\n", i, xcc & 0xf, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3,
               (lg[i * 4] - tmin) / 100.0, (lg[i * 4 + 1] - tmin) / 100.0, (unsigned long long)lg[i * 4 + 3]);
    }
    return 0;
}
