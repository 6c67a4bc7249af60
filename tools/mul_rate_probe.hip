// Issue rate of the Montgomery product and of the G1 mixed addition with operands in registers, per build of the
// arithmetic (tools/, not product code).  Build three times:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rapidsnark-old_amd/csrc tools/mul_rate_probe.hip -o tools/mul_rate_probe     (default build)
//   ... -DZK_STMT_MAD -o tools/mul_rate_probe_stmt            (one asm statement per MAD)
//   ... -DZK_COMPILER_MAD -o tools/mul_rate_probe_c           (C column sums: what hipcc makes of them)
#include "../rapidsnark-old_amd/csrc/msm.hip"
#include <stdio.h>
using namespace zk;

template <int MODE>      // 0: x = mul(x, y)   1: two lone products per iteration   2: mul2 (interleaved pair)   3: G1 madd   4: G2 madd, Fq2 split across a lane pair
__global__ __launch_bounds__(256) void k_loop(uint32_t *out, const Affine<Fq> *pts, uint32_t iters) {
    typedef Fq29 FR;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Affine<FR> P0 = load_affine(pts + (t & 1023u)), P1 = load_affine(pts + ((t + 7u) & 1023u));
    uint32_t x = 0;
    if (MODE == 4) {
        Affine<Fq2s> Q0{Fq2s{P0.x}, Fq2s{P0.y}}, Q1{Fq2s{P1.x}, Fq2s{P1.y}};
        XYZZ<Fq2s> acc{Fq2s{P1.x}, Fq2s{P1.y}, Fq2s::one(), Fq2s::one()};
        for (uint32_t i = 0; i < iters; i++) {
            madd(acc, (i & 1u) ? Q1 : Q0);
            Q0.x.v.l[0] ^= (int32_t)(i & 3u);
        }
        for (int k = 0; k < 9; k++) x ^= (uint32_t)(acc.x.v.l[k] ^ acc.y.v.l[k] ^ acc.zz.v.l[k] ^ acc.zzz.v.l[k]);
    } else if (MODE == 3) {
        XYZZ<FR> acc = XYZZ<FR>::from_affine(P1);
        for (uint32_t i = 0; i < iters; i++) {
            madd(acc, (i & 1u) ? P1 : P0);
            P0.x.l[0] ^= (int32_t)(i & 3u);
        }
        G1Acc o;
        LaneModel<Fq>::store(&o, acc);
        for (int k = 0; k < 36; k++) x ^= (uint32_t)o.l[k];
    } else {
        FR a = P0.x, b = P0.y, y = P1.x;
        for (uint32_t i = 0; i < iters; i++) {
            if (MODE == 0) a = FR::mul(a, y);
            else if (MODE == 1) { a = FR::mul(a, y); b = FR::mul(b, y); }
            else FR::mul2(a, a, y, b, b, y);
            y.l[0] ^= (int32_t)(i & 3u);
        }
        for (int k = 0; k < 9; k++) x ^= (uint32_t)(a.l[k] ^ b.l[k]);
    }
    out[t] = x;
}

template <class K>
static double run(K k, int blocks, uint32_t *out, const Affine<Fq> *pts, uint32_t iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, pts, iters);
    hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, pts, iters);
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
    const int cus = prop.multiProcessorCount;
    Affine<Fq> *pts; uint32_t *out;
    hipMalloc(&pts, 1024 * sizeof(Affine<Fq>));
    std::vector<uint32_t> h(1024 * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) >> 3;
    hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    const uint32_t iters = 2000;
    const double ghz = 1.96;
#if defined(ZK_COMPILER_MAD)
    const char *build = "C column sums everywhere";
#elif defined(ZK_STMT_MAD)
    const char *build = "one asm statement per MAD";
#else
    const char *build = "default: pairs as asm blocks with alternating chains, lone products as C column sums";
#endif
    printf("build: %s\n", build);
    for (int wps = 1; wps <= 3; wps++) {
        const int blocks = cus * wps;
        const double f = 1e-3 * ghz * 1e9 / ((double)iters * wps);
        printf("%d wave(s)/SIMD: cycles per wave-level product: lone %.0f | two lone per iteration %.0f | interleaved pair %.0f ; G1 mixed addition %.0f cycles ; G2 mixed addition (lane pair) %.0f cycles\n", wps,
               run(k_loop<0>, blocks, out, pts, iters) * f, run(k_loop<1>, blocks, out, pts, iters) * f / 2, run(k_loop<2>, blocks, out, pts, iters) * f / 2,
               run(k_loop<3>, blocks, out, pts, iters) * f, run(k_loop<4>, blocks, out, pts, iters) * f);
    }
    return 0;
}
