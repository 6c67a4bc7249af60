// Batched-affine bucket additions on gfx950, priced by measurement: B independent affine
// pair-additions per lane, ONE field inversion per workgroup (Montgomery's trick: in-lane prefix products, a wave scan
// over the lane totals with ds_bpermute shuffles, the wave totals through LDS), operands in registers — against the
// XYZZ mixed addition of the level-1 kernels (curve29.hpp madd: 10 products, 7180-7200 cycles per wave-level addition
// and SIMD at three waves per SIMD, tools/mul_rate_probe).  tools/, not product code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rapidsnark-old_amd/csrc tools/batch_affine_probe.hip -o tools/batch_affine_probe
//
// What one lane-batch costs, in wave-level products:   phase A  B-1   (c_k = c_(k-1) * d_k,  d_k = x2_k - x1_k)
//                                                      scan     12 (inclusive prefix + suffix over 64 lanes) + 6 (four wave totals) + 2
//                                                      inverse  one Fermat power per WORKGROUP (wave 0; the other waves wait at a barrier)
//                                                      phase B  5 B  (1/d_k = u c_(k-1); u *= d_k; lambda; lambda^2; lambda (x1 - x3))
// so 6 B + 19 per lane plus the inversion: 8.4 products per addition at B = 8 before the inversion, against 10.
// MODE 0: everything; MODE 1: without the inversion (its result replaced by the total itself: wrong sums, right
// instruction stream) — the difference is the price of the inversion at this sharing; MODE 2: the XYZZ mixed addition.
// The operands of a batch are derived from two resident points by adding small constants to a limb (one instruction per
// coordinate): a real kernel would have to hold 4 x 9 limbs per pair across the inversion (288 VGPRs at B = 8) or gather
// every point twice — this probe leaves that out on purpose: it is the optimistic bound.
#include "../rapidsnark-old_amd/csrc/msm.hip"
#include <stdio.h>
using namespace zk;

typedef Fq29 FR;

__device__ __forceinline__ FR shfl_up9(const FR &v, uint32_t d) {
    FR r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = __shfl_up(v.l[i], d);
    return r;
}
__device__ __forceinline__ FR shfl_dn9(const FR &v, uint32_t d) {
    FR r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = __shfl_down(v.l[i], d);
    return r;
}
__device__ __forceinline__ FR bcast9(const FR &v, int lane) {
    FR r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = __shfl(v.l[i], lane);
    return r;
}
__device__ __forceinline__ FR sel9(bool c, const FR &a, const FR &b) {
    FR r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
// operand k of a batch: the resident coordinate plus a small lane- and k-dependent constant in limb 1 (stays a tight operand)
__device__ __forceinline__ FR tweak(const FR &a, uint32_t k) {
    FR r = a;
    r.l[1] += (int32_t)k;
    return r;
}

// 1 / (product of the whole workgroup's lane totals), handed to every lane as the inverse of ITS total.
// lds: 4 wave totals + the inverse, 9 limbs each.
template <int MODE>
__device__ __forceinline__ FR workgroup_inverse_of_lane_total(const FR &T, int32_t *lds) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // inclusive prefix P_l = T_0 .. T_l and suffix S_l = T_l .. T_63 (Hillis-Steele, 6 + 6 products)
    FR P = T, S = T;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const FR up = shfl_up9(P, d), dn = shfl_dn9(S, d);
        FR p2, s2;
        FR::mul2(p2, P, up, s2, S, dn);
        P = sel9(lane >= d, p2, P);
        S = sel9(lane + d < 64, s2, S);
    }
    const FR Wtot = bcast9(P, 63);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) lds[wave * 9 + i] = Wtot.l[i];
    }
    __syncthreads();
    FR W[4];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int i = 0; i < 9; i++) W[w].l[i] = lds[w * 9 + i];
    // product of the other waves' totals (3 products, uniform code: ones where a factor is this wave's own)
    const FR one = FR::one();
    FR others = sel9(wave == 0, one, W[0]);
    others = FR::mul(others, sel9(wave == 1, one, W[1]));
    others = FR::mul(others, sel9(wave == 2, one, W[2]));
    others = FR::mul(others, sel9(wave == 3, one, W[3]));
    if (wave == 0) {                       // one wave inverts the grand total (all its lanes redundantly: same wave time as one lane)
        const FR G = FR::mul(Wtot, others);
        const FR GI = MODE == 0 ? FR::inv(G) : G;
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; i++) lds[36 + i] = GI.l[i];
        }
    }
    __syncthreads();
    FR GI;
#pragma unroll
    for (int i = 0; i < 9; i++) GI.l[i] = lds[36 + i];
    // 1 / T_l = GI * others * P_(l-1) * S_(l+1)
    const FR Pb = sel9(lane > 0, shfl_up9(P, 1), one), Sa = sel9(lane < 63, shfl_dn9(S, 1), one);
    FR a, b;
    FR::mul2(a, GI, others, b, Pb, Sa);
    __syncthreads();                       // lds is reused by the next batch
    return FR::mul(a, b);
}

template <int B, int MODE>
__global__ __launch_bounds__(256) void k_batch_affine(uint32_t *out, uint32_t *bad, const Affine<Fq> *pts, uint32_t iters) {
    __shared__ int32_t lds[45];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Affine<FR> P0 = load_affine(pts + (t & 1023u)), P1 = load_affine(pts + ((t + 7u) & 1023u));
    uint32_t x = 0, nbad = 0;
    if (MODE == 2) {
        XYZZ<FR> acc = XYZZ<FR>::from_affine(P1);
        for (uint32_t i = 0; i < iters * B; i++) {
            madd(acc, (i & 1u) ? P1 : P0);
            P0.x.l[0] ^= (int32_t)(i & 3u);
        }
        G1Acc o;
        LaneModel<Fq>::store(&o, acc);
        for (int k = 0; k < 36; k++) x ^= (uint32_t)o.l[k];
        out[t] = x;
        return;
    }
    for (uint32_t it = 0; it < iters; it++) {
        // ---- phase A: prefix products of the denominators
        FR c[B];
        c[0] = FR::sub_nc(tweak(P1.x, 0), tweak(P0.x, 64));
#pragma unroll
        for (int k = 1; k < B; k++) c[k] = FR::mul(c[k - 1], FR::sub_nc(tweak(P1.x, k), tweak(P0.x, 64 + 3 * k)));
        // ---- one inversion per workgroup
        FR u = workgroup_inverse_of_lane_total<MODE>(c[B - 1], lds);
        // ---- phase B: back-substitution + the additions themselves
        FR sx = FR::zero(), sy = FR::zero();
#pragma unroll
        for (int k = B - 1; k >= 0; k--) {
            const FR x1 = tweak(P0.x, 64 + 3 * k), x2 = tweak(P1.x, k), y1 = tweak(P0.y, k), y2 = tweak(P1.y, 5 * k);
            const FR d = FR::sub_nc(x2, x1);
            FR dinv;
            if (k > 0) { FR un; FR::mul2(dinv, u, c[k - 1], un, u, d); u = un; }
            else dinv = u;
            const FR lam = FR::mul(FR::sub_nc(y2, y1), dinv);
            const FR l2 = FR::sqr(lam);
            FR x3;
#pragma unroll
            for (int i = 0; i < 9; i++) x3.l[i] = l2.l[i] - x1.l[i] - x2.l[i];
            x3 = FR::carry(x3);
            const FR y3 = FR::sub(FR::mul(lam, FR::sub_nc(x1, x3)), y1);
            if (MODE == 0 && it == 0) {          // checker (first batch only): the same sum through the XYZZ mixed addition
                XYZZ<FR> a = XYZZ<FR>::from_affine(Affine<FR>{x1, y1});
                madd(a, Affine<FR>{x2, y2});
                // x3 * zz == X and y3 * zzz == Y
                if (!FR::sub(FR::mul(x3, a.zz), a.x).is_zero() || !FR::sub(FR::mul(y3, a.zzz), a.y).is_zero()) nbad++;
            }
            sx = FR::add(sx, x3);
            sy = FR::add(sy, y3);
        }
        // feed the sums back so that no batch can be hoisted or dropped
        P0.x.l[0] ^= sx.l[0] & 3;
        P0.y.l[0] ^= sy.l[0] & 3;
    }
    for (int k = 0; k < 9; k++) x ^= (uint32_t)(P0.x.l[k] ^ P0.y.l[k]);
    out[t] = x;
    if (nbad) atomicAdd(bad, nbad);
}

template <class K>
static double run(K k, int blocks, uint32_t *out, uint32_t *bad, const Affine<Fq> *pts, uint32_t iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, bad, pts, iters);
    hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, bad, pts, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

template <int B>
static void report(int cus, uint32_t *out, uint32_t *bad, const Affine<Fq> *pts, double ghz) {
    const uint32_t iters = 2048 / B;            // the same number of additions per lane for every B
    for (int wps = 1; wps <= 3; wps++) {
        const int blocks = cus * wps;
        const double f = 1e-3 * ghz * 1e9 / ((double)iters * B * wps);     // cycles per wave-level addition and SIMD
        hipMemset(bad, 0, 4);
        const double full = run(k_batch_affine<B, 0>, blocks, out, bad, pts, iters) * f;
        uint32_t nbad = 0;
        hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
        const double noinv = run(k_batch_affine<B, 1>, blocks, out, bad, pts, iters) * f;
        const double xyzz = run(k_batch_affine<B, 2>, blocks, out, bad, pts, iters) * f;
        printf("B = %2d, %d wave(s)/SIMD: batched affine %6.0f cycles per addition (without the inversion %6.0f) | XYZZ mixed addition %6.0f | ratio %.3f | checker mismatches %u\n",
               B, wps, full, noinv, xyzz, full / xyzz, nbad);
    }
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    Affine<Fq> *pts; uint32_t *out, *bad;
    hipMalloc(&pts, 1024 * sizeof(Affine<Fq>));
    std::vector<uint32_t> h(1024 * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) >> 3;      // arbitrary field elements (< 2^29 per word)
    hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    hipMalloc(&bad, 4);
    const double ghz = 1.96;      // as tools/mul_rate_probe.hip, so the two files compare
    printf("batched-affine pair additions, one inversion per 256-lane workgroup, operands in registers (cycles at %.2f GHz per wave-level addition and SIMD)\n", ghz);
    report<4>(cus, out, bad, pts, ghz);
    report<8>(cus, out, bad, pts, ghz);
    report<16>(cus, out, bad, pts, ghz);
    return 0;
}
