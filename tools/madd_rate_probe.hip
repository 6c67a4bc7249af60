// How fast does the G1 mixed addition of the level-1 kernel issue when NOTHING else is in the way — no gathers, no
// bucket logic, operands in registers?  (tools/, not product code.)  Prints cycles per wave-level addition per SIMD at
// 1..4 waves per SIMD, next to the 2318-instruction count and the cost-model figure of DESIGN.md section 6.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rapidsnark-old_amd/csrc tools/madd_rate_probe.hip -o tools/madd_rate_probe
#include "../rapidsnark-old_amd/csrc/msm.hip"
#include <stdio.h>
using namespace zk;

template <int MAXW>
__global__ __launch_bounds__(256, MAXW) void k_madd_loop(uint32_t *out, const Affine<Fq> *pts, uint32_t iters) {
    typedef Fq29 FR;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Affine<FR> P0 = load_affine(pts + (t & 1023u)), P1 = load_affine(pts + ((t + 7u) & 1023u));      // (load_affine: words -> 29-bit limbs)
    XYZZ<FR> acc = XYZZ<FR>::from_affine(P1);
    for (uint32_t i = 0; i < iters; i++) {
        madd(acc, (i & 1u) ? P1 : P0);
        P0.x.l[0] ^= (int32_t)(i & 3u);          // keeps the operands from being hoisted
    }
    G1Acc o;
    LaneModel<Fq>::store(&o, acc);
    uint32_t x = 0;
    for (int k = 0; k < 36; k++) x ^= (uint32_t)o.l[k];
    out[t] = x;
}

template <class K>
static double run(K k, int blocks, uint32_t *out, const Affine<Fq> *pts, uint32_t iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, pts, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, pts, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    Affine<Fq> *pts; uint32_t *out;
    hipMalloc(&pts, 1024 * sizeof(Affine<Fq>));
    std::vector<uint32_t> h(1024 * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) >> 3;      // arbitrary field elements (< 2^29 per word)
    hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    const uint32_t iters = 2000;
    const double ghz = 1.96;      // the shader clock measured under this load
    for (int wps = 1; wps <= 4; wps++) {
        const int blocks = cus * wps;      // 256 threads = one wave per SIMD per block
        double ms = wps <= 3 ? run(k_madd_loop<3>, blocks, out, pts, iters) : run(k_madd_loop<4>, blocks, out, pts, iters);
        double cyc = ms * 1e-3 * ghz * 1e9 / ((double)iters * wps);
        printf("%d wave(s)/SIMD: %.3f ms, %.0f cycles per wave-addition per SIMD at %.2f GHz (2318 instructions: %.2f cycles each)\n", wps, ms, cyc, ghz, cyc / 2318.0);
    }
    return 0;
}
