// Issue rate of the Montgomery product and of the G1 mixed addition with operands in registers, per build of the
// arithmetic (tools/, not product code).  Build three times:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rapidsnark-old_amd/csrc tools/mul_rate_probe.hip -o tools/mul_rate_probe     (default build)
//   ... -DZK_STMT_MAD -o tools/mul_rate_probe_stmt            (one asm statement per MAD)
//   ... -DZK_COMPILER_MAD -o tools/mul_rate_probe_c           (C column sums: what hipcc makes of them)
#include "../rapidsnark-old_amd/csrc/msm.hip"
#include <stdio.h>
using namespace zk;

// ---- Fq2 product in ONE lane, three ways (round 5: is a three-product Fq2 multiplication worth it on this arithmetic?)
//   f2_school_blocks : c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0 as two fused double products, chains alternating (run2, asm blocks): 4 x 81 + 2 x 81 MADs
//   f2_school_c      : the same as C column sums (what hipcc makes of them) — the like-for-like partner of the next one
//   f2_karatsuba_c   : v0 = a0 b0, v1 = a1 b1, v2 = (a0 - a1)(b1 - b0) as column sums; column k of c0 = v0 - v1, of c1 = v2 + v0 + v1; two
//                      reductions: 3 x 81 + 2 x 81 MADs, but the three column sums have to be MERGED per column (64-bit adds) before
//                      the two reduction chains can take them
struct F2 { Fq29 a, b; };
__device__ __forceinline__ F2 f2_school_blocks(const F2 &x, const F2 &y) {
    F2 r;
    const Fq29 nb = Fq29::neg_lazy(x.b);
    Fq29::run2(r.a, Fq29::JMulAdd2{x.a, y.a, nb, y.b}, r.b, Fq29::JMulAdd2{x.a, y.b, x.b, y.a});
    return r;
}
__device__ __forceinline__ void f2_columns(F2 &r, const F2 &x, const F2 &y, bool karatsuba) {
    typedef Fq29 F;
    int64_t acc0 = 0, acc1 = 0;
    int32_t m0[9], m1[9];
    int32_t s[9], t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { s[i] = x.a.l[i] - x.b.l[i]; t[i] = y.b.l[i] - y.a.l[i]; }
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
        int64_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            v0 += (int64_t)x.a.l[i] * y.a.l[k - i];
            v1 += (int64_t)x.b.l[i] * y.b.l[k - i];
            if (karatsuba) v2 += (int64_t)s[i] * t[k - i];
            else { v2 += (int64_t)x.a.l[i] * y.b.l[k - i]; v3 += (int64_t)x.b.l[i] * y.a.l[k - i]; }
        }
        acc0 += v0 - v1;
        acc1 += karatsuba ? v2 + v0 + v1 : v2 + v3;
#pragma unroll
        for (int i = (k < 9 ? 0 : k - 8); i < (k < 9 ? k : 9); i++) {
            acc0 += (int64_t)m0[i] * F::PS(k - i);
            acc1 += (int64_t)m1[i] * F::PS(k - i);
        }
        if (k < 9) {
            m0[k] = F::mont_m(acc0, k == 8); m1[k] = F::mont_m(acc1, k == 8);
            acc0 += (int64_t)m0[k] * F::PS(0); acc1 += (int64_t)m1[k] * F::PS(0);
        } else {
            r.a.l[k - 9] = F::out_limb(acc0); r.b.l[k - 9] = F::out_limb(acc1);
        }
        acc0 >>= 29; acc1 >>= 29;
    }
    r.a.l[8] = (int32_t)acc0; r.b.l[8] = (int32_t)acc1;
}
template <int WHICH>
__global__ __launch_bounds__(256) void k_f2(uint32_t *out, const Affine<Fq> *pts, uint32_t iters, uint32_t *mismatch) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Affine<Fq29> P0 = load_affine(pts + (t & 1023u)), P1 = load_affine(pts + ((t + 7u) & 1023u));
    F2 x{P0.x, P0.y}, y{P1.x, P1.y};
    if (WHICH == 3) {                  // check: the three forms agree (canonical representatives) over a chain of products
        F2 a = x, b = x, c = x;
        for (uint32_t i = 0; i < 64; i++) {
            a = f2_school_blocks(a, y); f2_columns(b, F2(b), y, false); f2_columns(c, F2(c), y, true);
            a.a = Fq29::carry(a.a); a.b = Fq29::carry(a.b); b.a = Fq29::carry(b.a); b.b = Fq29::carry(b.b); c.a = Fq29::carry(c.a); c.b = Fq29::carry(c.b);
        }
        const Fq A0 = Fq29::store(a.a), A1 = Fq29::store(a.b), B0 = Fq29::store(b.a), B1 = Fq29::store(b.b), C0 = Fq29::store(c.a), C1 = Fq29::store(c.b);
        bool bad = false;
        for (int k = 0; k < 8; k++) bad |= A0.v[k] != B0.v[k] || A1.v[k] != B1.v[k] || A0.v[k] != C0.v[k] || A1.v[k] != C1.v[k];
        if (bad) atomicAdd(mismatch, 1u);
        return;
    }
    for (uint32_t i = 0; i < iters; i++) {
        if (WHICH == 0) x = f2_school_blocks(x, y);
        else { F2 r; f2_columns(r, x, y, WHICH == 2); x = r; }
        x.a = Fq29::carry(x.a); x.b = Fq29::carry(x.b);          // operands back to non-negative limbs (what the callers of a Karatsuba form would pay too)
        y.a.l[0] ^= (int32_t)(i & 3u);
    }
    uint32_t o = 0;
    for (int k = 0; k < 9; k++) o ^= (uint32_t)(x.a.l[k] ^ x.b.l[k]);
    out[t] = o;
}

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
    // ---- Fq2 product in one lane: schoolbook with one reduction per component (blocks / C) against Karatsuba with merged columns (C)
    uint32_t *mm;
    hipMalloc(&mm, 4);
    hipMemset(mm, 0, 4);
    hipLaunchKernelGGL(k_f2<3>, dim3(cus), dim3(256), 0, 0, out, pts, iters, mm);
    uint32_t bad = 1;
    hipMemcpy(&bad, mm, 4, hipMemcpyDeviceToHost);
    printf("Fq2 product forms agree on %d lanes x 64 chained products: %s\n", cus * 256, bad ? "NO" : "yes");
    auto run_f2 = [&](auto kern, int blocks) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, pts, iters, mm);
        hipDeviceSynchronize();
        double best = 1e30;
        for (int r = 0; r < 3; r++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, pts, iters, mm);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    for (int wps = 1; wps <= 3; wps++) {
        const int blocks = cus * wps;
        const double f = 1e-3 * ghz * 1e9 / ((double)iters * wps);
        printf("%d wave(s)/SIMD: cycles per wave-level Fq2 product (+ two carries): schoolbook, asm blocks %.0f | schoolbook, C columns %.0f | Karatsuba, C columns %.0f\n", wps,
               run_f2(k_f2<0>, blocks) * f, run_f2(k_f2<1>, blocks) * f, run_f2(k_f2<2>, blocks) * f);
    }
    return 0;
}
