// Synthetic point tables for benchmarks and full-size parity checks:
// P_i = P0 + i*Q, affine Montgomery, generated on the GPU — 2^22..2^24 valid, distinct curve
// points in milliseconds.  With P0 = k0*G, Q = kq*G every discrete log is known, so an MSM
// over the table is checkable in Fr alone (sum s_i*(k0+i*kq)) at sizes no CPU oracle reaches.
// Pass 1: each lane jumps to its segment start by double-and-add on the index, then walks
// the chain with mixed adds.  Pass 2: XYZZ -> affine with one Fermat inversion per segment
// (Montgomery's batch-inversion trick over the segment's ZZ*ZZZ products).
#include "kernels.hpp"
#include "hipcheck.hpp"

namespace zk {

#define CHAIN_SEG 64u

template <class F>
__device__ __forceinline__ F ld(const F *p);
template <>
__device__ __forceinline__ Fq ld<Fq>(const Fq *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 lo = q[0], hi = q[1];
    Fq r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
template <>
__device__ __forceinline__ Fq2 ld<Fq2>(const Fq2 *p) { return Fq2{ld(&p->a), ld(&p->b)}; }
__device__ __forceinline__ void st(Fq *p, const Fq &r) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
__device__ __forceinline__ void st(Fq2 *p, const Fq2 &r) { st(&p->a, r.a); st(&p->b, r.b); }

template <class F>
__global__ __launch_bounds__(64) void k_chain_walk(XYZZ<F> *tmp, Affine<F> P0, Affine<F> Q, uint64_t n) {
    uint64_t lo = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * CHAIN_SEG;
    if (lo >= n) return;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int bit = 63 - __clzll(lo | 1ull); bit >= 0; bit--) {
        acc = dbl(acc);
        if ((lo >> bit) & 1ull) madd(acc, Q);
    }
    madd(acc, P0);
    uint64_t hi = lo + CHAIN_SEG < n ? lo + CHAIN_SEG : n;
    for (uint64_t i = lo; i < hi; i++) {
        st(&tmp[i].x, acc.x); st(&tmp[i].y, acc.y); st(&tmp[i].zz, acc.zz); st(&tmp[i].zzz, acc.zzz);
        madd(acc, Q);
    }
}

template <class F>
__global__ __launch_bounds__(64) void k_chain_normalize(Affine<F> *out, const XYZZ<F> *tmp, F *pref, uint64_t n) {
    uint64_t lo = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * CHAIN_SEG;
    if (lo >= n) return;
    uint64_t hi = lo + CHAIN_SEG < n ? lo + CHAIN_SEG : n;
    F acc = F::one();
    for (uint64_t i = lo; i < hi; i++) {
        st(&pref[i], acc);
        F t = F::mul(ld(&tmp[i].zz), ld(&tmp[i].zzz));
        if (!t.is_zero()) acc = F::mul(acc, t);          // infinity in the chain: skipped
    }
    F inv = F::inv(acc);
    for (uint64_t i = hi; i-- > lo;) {
        F zz = ld(&tmp[i].zz), zzz = ld(&tmp[i].zzz);
        F t = F::mul(zz, zzz);
        if (t.is_zero()) {
            st(&out[i].x, F::zero()); st(&out[i].y, F::zero());
            continue;
        }
        F ii = F::mul(inv, ld(&pref[i]));                // 1/(zz*zzz)
        inv = F::mul(inv, t);
        st(&out[i].x, F::mul(ld(&tmp[i].x), F::mul(ii, zzz)));   // X/ZZ
        st(&out[i].y, F::mul(ld(&tmp[i].y), F::mul(ii, zz)));    // Y/ZZZ
    }
}

template <class F>
static void chain(Affine<F> *d_out, XYZZ<F> *d_tmp, F *d_pref, const Affine<F> &P0, const Affine<F> &Q, uint64_t n, hipStream_t s) {
    uint64_t segs = (n + CHAIN_SEG - 1) / CHAIN_SEG;
    uint32_t blocks = (uint32_t)((segs + 63) / 64);
    ZK_LAUNCH(k_chain_walk<F>, dim3(blocks), dim3(64), 0, s, d_tmp, P0, Q, n);
    ZK_LAUNCH(k_chain_normalize<F>, dim3(blocks), dim3(64), 0, s, d_out, (const XYZZ<F> *)d_tmp, d_pref, n);
    ZK_LAUNCH_OK("synthetic chain");
}

// Batch fixed-base multiplication out[i] = k_i * B (a trapdoor-valid zkey is nothing but
// five such batches over the evaluations A_i(tau), B_i(tau), ... computed in Fr).  One lane per
// scalar, plain double-and-add over all 256 bits (no table: 5 x 2^20 points take well under a second),
// then the same batched normalisation as the chains.  k_i = 0 gives the all-zero infinity encoding.
template <class F>
__global__ __launch_bounds__(64) void k_fixed_base(XYZZ<F> *tmp, Affine<F> B, const uint32_t *scalars, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8];
#pragma unroll
    for (int j = 0; j < 8; j++) k[j] = scalars[i * 8 + j];
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int bit = 255; bit >= 0; bit--) {
        acc = dbl(acc);
        if ((k[bit >> 5] >> (bit & 31)) & 1u) madd(acc, B);
    }
    st(&tmp[i].x, acc.x); st(&tmp[i].y, acc.y); st(&tmp[i].zz, acc.zz); st(&tmp[i].zzz, acc.zzz);
}
template <class F>
static void fixed_base(Affine<F> *d_out, XYZZ<F> *d_tmp, F *d_pref, const Affine<F> &B, const uint32_t *d_scalars, uint64_t n, hipStream_t s) {
    ZK_LAUNCH(k_fixed_base<F>, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, s, d_tmp, B, d_scalars, n);
    uint64_t segs = (n + CHAIN_SEG - 1) / CHAIN_SEG;
    ZK_LAUNCH(k_chain_normalize<F>, dim3((uint32_t)((segs + 63) / 64)), dim3(64), 0, s, d_out, (const XYZZ<F> *)d_tmp, d_pref, n);
    ZK_LAUNCH_OK("fixed-base batch");
}
void launch_fixed_base_g1(G1Affine *d_out, G1XYZZ *d_tmp, Fq *d_pref, const G1Affine &B, const uint32_t *d_scalars, uint64_t n, hipStream_t s) {
    fixed_base<Fq>(d_out, d_tmp, d_pref, B, d_scalars, n, s);
}
void launch_fixed_base_g2(G2Affine *d_out, G2XYZZ *d_tmp, Fq2 *d_pref, const G2Affine &B, const uint32_t *d_scalars, uint64_t n, hipStream_t s) {
    fixed_base<Fq2>(d_out, d_tmp, d_pref, B, d_scalars, n, s);
}

void launch_chain_g1(G1Affine *d_out, G1XYZZ *d_tmp, Fq *d_pref, const G1Affine &P0, const G1Affine &Q, uint64_t n, hipStream_t s) {
    chain<Fq>(d_out, d_tmp, d_pref, P0, Q, n, s);
}
void launch_chain_g2(G2Affine *d_out, G2XYZZ *d_tmp, Fq2 *d_pref, const G2Affine &P0, const G2Affine &Q, uint64_t n, hipStream_t s) {
    chain<Fq2>(d_out, d_tmp, d_pref, P0, Q, n, s);
}

}   // namespace zk
