// Pippenger multi-scalar multiplication over BN254 G1 / G2 for gfx950 — replaces ffiasm
// ParallelMultiexp behind Curve::multiMulByScalar (call sites src/groth16.cpp:173,183,190,197,204).
//
// MI355X design (ffiasm keeps nThreads x 2^c per-thread bucket arrays; that makes no sense here):
//   1. k_msm_count   : signed c-bit digits of every scalar -> histogram per (window, bucket)
//   2. k_msm_scan    : exclusive scan -> bucket offsets
//   3. k_msm_scatter : counting sort of (point index | sign) into bucket order
//      (1-3 run ONCE per scalar vector: the witness sort is shared by MSM A, B1, B2, C,
//       which the reference recomputes four times, src/groth16.cpp:183-204)
//   4. k_msm_accum   : one lane per bucket walks its sorted run and mixed-adds the gathered
//      affine points into an XYZZ accumulator kept in VGPRs (next point prefetched)
//   5. k_msm_reduce_chunks / k_msm_reduce_final : sum_k (k+1)*B_k per window via chunked
//      running sums + an LDS tree
//   6. host: Horner over the W window sums (host_tail.cpp) — 256 serial doublings are
//      30x faster on one CPU core than on one GPU lane.
// Signed digits halve the bucket count; scalars are reduced mod r first so any 256-bit
// input is accepted like the reference's raw-byte interface.
#include "kernels.hpp"

namespace zk {

#define REDUCE_CHUNK 16u
#define REDUCE_THREADS 256u

MsmPlan make_msm_plan(uint64_t n, uint32_t window_bits) {
    MsmPlan p;
    uint32_t c = window_bits;
    if (c == 0) {
        uint32_t lg = 0;
        while ((1ull << (lg + 1)) <= n) lg++;
        c = lg > 6 ? lg - 6 : 2;     // ~128 points per bucket on random scalars
        if (c > 16) c = 16;
    }
    if (c < 2) c = 2;
    if (c > 20) c = 20;
    p.c = c;
    p.W = (255 + c - 1) / c;
    p.nbuckets = 1u << (c - 1);
    return p;
}

template <class F>
__device__ __forceinline__ F load_el(const F *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 lo = q[0], hi = q[1];
    F r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
template <class F>
__device__ __forceinline__ void store_el(F *p, const F &r) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
__device__ __forceinline__ Fq2 load_el(const Fq2 *p) { return Fq2{load_el(&p->a), load_el(&p->b)}; }
__device__ __forceinline__ void store_el(Fq2 *p, const Fq2 &r) {
    store_el(&p->a, r.a);
    store_el(&p->b, r.b);
}
template <class F>
__device__ __forceinline__ Affine<F> load_affine(const Affine<F> *p) {
    return Affine<F>{load_el(&p->x), load_el(&p->y)};
}
template <class F>
__device__ __forceinline__ XYZZ<F> load_xyzz(const XYZZ<F> *p) {
    return XYZZ<F>{load_el(&p->x), load_el(&p->y), load_el(&p->zz), load_el(&p->zzz)};
}
template <class F>
__device__ __forceinline__ void store_xyzz(XYZZ<F> *p, const XYZZ<F> &v) {
    store_el(&p->x, v.x);
    store_el(&p->y, v.y);
    store_el(&p->zz, v.zz);
    store_el(&p->zzz, v.zzz);
}

// Walk the signed c-bit digits of scalar s (standard form; reduced mod r here) and call
// f(window, bucket_index = |d|-1, negative) for every non-zero digit d in [-2^(c-1), 2^(c-1)].
template <class Fn>
__device__ __forceinline__ void for_each_digit(Fr s, uint32_t c, uint32_t W, Fn f) {
    // any 256-bit value is < 6r: bring it below r (never loops for well-formed inputs)
    for (int k = 0; k < 6; k++) {
        Fr d;
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = subb(s.v[i], FrParams::P[i], bw);
        if (bw) break;
        s = d;
    }
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    uint64_t buf = 0;
    uint32_t nb = 0, w = 0, carry = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        buf |= (uint64_t)s.v[k] << nb;
        nb += 32;
        while (nb >= c && w + 1 < W) {
            uint32_t d = ((uint32_t)buf & mask) + carry;
            buf >>= c;
            nb -= c;
            if (d > half) {
                carry = 1;
                uint32_t nd = (1u << c) - d;          // |digit|; 0 when raw = 2^c-1 and carry = 1
                if (nd) f(w, nd - 1u, true);
            } else {
                carry = 0;
                if (d) f(w, d - 1u, false);
            }
            w++;
        }
    }
    // top window: whatever is left (value < 2^254 and W*c >= 255 => d <= 2^(c-1))
    uint32_t d = (uint32_t)buf + carry;
    if (d) f(W - 1, d - 1u, false);
}

__global__ __launch_bounds__(256) void k_msm_count(uint32_t *counts, const Fr *scalars, uint64_t n, MsmPlan p) {
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        Fr s = load_el(scalars + i);
        for_each_digit(s, p.c, p.W, [&](uint32_t w, uint32_t b, bool) { atomicAdd(&counts[w * p.nbuckets + b], 1u); });
    }
}

__global__ __launch_bounds__(256) void k_msm_scatter(uint32_t *entries, uint32_t *cursor, const Fr *scalars, uint64_t n, MsmPlan p) {
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        Fr s = load_el(scalars + i);
        for_each_digit(s, p.c, p.W, [&](uint32_t w, uint32_t b, bool neg) {
            uint32_t pos = atomicAdd(&cursor[w * p.nbuckets + b], 1u);
            entries[pos] = (uint32_t)i | (neg ? 0x80000000u : 0u);
        });
    }
}

// single workgroup exclusive scan (total_buckets <= a few million): each of 1024 lanes owns a
// contiguous slice; slice sums are scanned through LDS.
__global__ __launch_bounds__(1024) void k_msm_scan(uint32_t *offsets, uint32_t *cursor, const uint32_t *counts, uint32_t total) {
    __shared__ uint32_t sums[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (total + 1023u) / 1024u;
    const uint32_t lo = tid * per;
    const uint32_t hi = lo + per < total ? lo + per : total;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += counts[i];
    sums[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = tid >= d ? sums[tid - d] : 0;
        __syncthreads();
        sums[tid] += v;
        __syncthreads();
    }
    uint32_t run = sums[tid] - s;
    for (uint32_t i = lo; i < hi; i++) {
        offsets[i] = run;
        cursor[i] = run;
        run += counts[i];
    }
    if (tid == 1023) offsets[total] = sums[1023];
}

// One lane per bucket.  Adjacent lanes own adjacent buckets of one window, so run lengths
// within a wave are similar (Poisson around n/2^(c-1)) and divergence stays at the tail.
template <class F>
__global__ __launch_bounds__(256) void k_msm_accum(XYZZ<F> *buckets, const uint32_t *offsets, const uint32_t *entries,
                                                   const Affine<F> *points, uint32_t idx_min, uint32_t idx_sub, uint32_t total) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= total) return;
    uint32_t e = offsets[b];
    const uint32_t end = offsets[b + 1];
    XYZZ<F> acc = XYZZ<F>::inf();
    // software prefetch: the next point's 64/128 bytes are in flight while this one is added
    Affine<F> nextP = Affine<F>::inf();
    bool nextNeg = false;
    auto fetch = [&](uint32_t pos) {
        uint32_t ent = entries[pos];
        uint32_t idx = ent & 0x7fffffffu;
        nextNeg = (ent >> 31) != 0;
        if (idx >= idx_min) nextP = load_affine(points + (idx - idx_sub));
        else nextP = Affine<F>::inf();
    };
    if (e < end) fetch(e);
    while (e < end) {
        Affine<F> P = nextP;
        bool ng = nextNeg;
        e++;
        if (e < end) fetch(e);
        if (ng) P.y = F::neg(P.y);
        madd(acc, P);
    }
    store_xyzz(buckets + b, acc);
}

// Lane per chunk of REDUCE_CHUNK buckets: running sums give A = sum (j+1)*B[lo+j], T = sum B;
// X = A + lo*T is the chunk's share of sum_k (k+1)*B_k.
template <class F>
__global__ __launch_bounds__(128) void k_msm_reduce_chunks(XYZZ<F> *scratch, const XYZZ<F> *buckets, uint32_t nbuckets,
                                                           uint32_t chunk, uint32_t total_chunks) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_chunks) return;
    const uint32_t chunks_per_window = nbuckets / chunk;
    const uint32_t cw = t % chunks_per_window;          // chunk index inside its window
    const XYZZ<F> *B = buckets + (uint64_t)t * chunk;   // windows (and MSMs) are laid back to back
    XYZZ<F> run = XYZZ<F>::inf(), sum = XYZZ<F>::inf();
    for (int j = (int)chunk - 1; j >= 0; j--) {
        add(run, load_xyzz(B + j));
        add(sum, run);
    }
    // sum += (cw*chunk) * run   — double-and-add, MSB first
    uint32_t k = cw * chunk;
    if (k) {
        XYZZ<F> m = XYZZ<F>::inf();
        for (int bit = 31 - __clz(k); bit >= 0; bit--) {
            m = dbl(m);
            if ((k >> bit) & 1u) add(m, run);
        }
        add(sum, m);
    }
    store_xyzz(scratch + t, sum);
}

// One workgroup per (msm, window): strided serial sums, then an LDS tree.
template <class F>
__global__ __launch_bounds__(REDUCE_THREADS) void k_msm_reduce_final(XYZZ<F> *window_sums, const XYZZ<F> *scratch,
                                                                     uint32_t chunks_per_window) {
    extern __shared__ uint32_t lds_raw[];
    XYZZ<F> *lds = reinterpret_cast<XYZZ<F> *>(lds_raw);
    const XYZZ<F> *X = scratch + (uint64_t)blockIdx.x * chunks_per_window;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t i = threadIdx.x; i < chunks_per_window; i += REDUCE_THREADS) add(acc, load_xyzz(X + i));
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = REDUCE_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            XYZZ<F> o = lds[threadIdx.x + s];
            add(acc, o);
            lds[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) store_xyzz(window_sums + blockIdx.x, acc);
}

static inline uint32_t reduce_chunk_for(MsmPlan p) { return p.nbuckets < REDUCE_CHUNK ? p.nbuckets : REDUCE_CHUNK; }

uint64_t msm_reduce_scratch_points(uint32_t n_msm, MsmPlan p) {
    return (uint64_t)n_msm * p.W * (p.nbuckets / reduce_chunk_for(p));
}

void launch_msm_count(uint32_t *counts, const Fr *scalars, uint64_t n, MsmPlan p, hipStream_t s) {
    if (!n) return;
    uint64_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_msm_count, dim3((uint32_t)g), dim3(256), 0, s, counts, scalars, n, p);
}
void launch_msm_scan(uint32_t *offsets, uint32_t *cursor, const uint32_t *counts, uint32_t total, hipStream_t s) {
    hipLaunchKernelGGL(k_msm_scan, dim3(1), dim3(1024), 0, s, offsets, cursor, counts, total);
}
void launch_msm_scatter(uint32_t *entries, uint32_t *cursor, const Fr *scalars, uint64_t n, MsmPlan p, hipStream_t s) {
    if (!n) return;
    uint64_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_msm_scatter, dim3((uint32_t)g), dim3(256), 0, s, entries, cursor, scalars, n, p);
}
void launch_msm_accum_g1(G1XYZZ *buckets, const uint32_t *offsets, const uint32_t *entries, const G1Affine *points,
                         uint32_t idx_min, uint32_t idx_sub, uint32_t total, hipStream_t s) {
    hipLaunchKernelGGL(k_msm_accum<Fq>, dim3((total + 255) / 256), dim3(256), 0, s, buckets, offsets, entries, points, idx_min, idx_sub, total);
}
void launch_msm_accum_g2(G2XYZZ *buckets, const uint32_t *offsets, const uint32_t *entries, const G2Affine *points,
                         uint32_t idx_min, uint32_t idx_sub, uint32_t total, hipStream_t s) {
    hipLaunchKernelGGL(k_msm_accum<Fq2>, dim3((total + 255) / 256), dim3(256), 0, s, buckets, offsets, entries, points, idx_min, idx_sub, total);
}

template <class F>
static void launch_reduce(XYZZ<F> *window_sums, XYZZ<F> *scratch, const XYZZ<F> *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s) {
    uint32_t chunk = reduce_chunk_for(p);
    uint32_t cpw = p.nbuckets / chunk;
    uint32_t total_chunks = n_msm * p.W * cpw;
    hipLaunchKernelGGL(k_msm_reduce_chunks<F>, dim3((total_chunks + 127) / 128), dim3(128), 0, s, scratch, buckets, p.nbuckets, chunk, total_chunks);
    hipLaunchKernelGGL(k_msm_reduce_final<F>, dim3(n_msm * p.W), dim3(REDUCE_THREADS), REDUCE_THREADS * sizeof(XYZZ<F>), s, window_sums, scratch, cpw);
}
void launch_msm_reduce_g1(G1XYZZ *ws, G1XYZZ *scratch, const G1XYZZ *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s) {
    launch_reduce<Fq>(ws, scratch, buckets, n_msm, p, s);
}
void launch_msm_reduce_g2(G2XYZZ *ws, G2XYZZ *scratch, const G2XYZZ *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s) {
    launch_reduce<Fq2>(ws, scratch, buckets, n_msm, p, s);
}

}   // namespace zk
