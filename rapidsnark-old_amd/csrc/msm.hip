// Pippenger multi-scalar multiplication over BN254 G1 / G2 for gfx950 — replaces ffiasm
// ParallelMultiexp behind Curve::multiMulByScalar (call sites src/groth16.cpp:173,183,190,197,204).
//
// MI355X design (ffiasm keeps nThreads x 2^c per-thread bucket arrays; that makes no sense here):
//   1. k_msm_digits       : signed c-bit digits of every scalar, window-major 16-bit codes
//   2. k_msm_count_lds    : per-(window, slice) histograms entirely in LDS; k_scan_* : exclusive scan
//   3. k_msm_scatter_lds  : counting sort of (point index | sign) into bucket order, LDS-ranked
//      (1-3 run ONCE per scalar vector: the witness sort is shared by MSM A, B1, B2, C,
//       which the reference recomputes four times, src/groth16.cpp:183-204)
//   4. k_msm_accum_l1/_ln : load-balanced segmented accumulation — every lane mixed-adds a
//      fixed-size chunk of the bucket-sorted list into XYZZ accumulators in VGPRs (next point
//      prefetched); runs cut by chunk edges are merged by recursively shrinking levels
//   5. k_msm_reduce_chunks / k_msm_reduce_final : sum_k (k+1)*B_k per window via chunked
//      running sums + an LDS tree
//   6. host: Horner over the W window sums (host_tail.cpp) — 256 serial doublings are
//      30x faster on one CPU core than on one GPU lane.
// Signed digits halve the bucket count; scalars are reduced mod r first so any 256-bit
// input is accepted like the reference's raw-byte interface.
#include "kernels.hpp"
#include "field29.hpp"

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
    if (c > 16) c = 16;             // the LDS histogram sort holds 2^(c-1) counters per workgroup
    p.c = c;
    p.W = (256 + c - 1) / c;        // W*c >= 256: the top digit is never negative (symmetric recoding)
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

// Register representation of the MSM kernels: 9x29-bit signed limbs (field29.hpp).  HBM keeps
// canonical 256-bit words of the SAME (2^261) Montgomery form; Reg<> converts at load/store.
typedef Fp2T<Fq29> Fq2r;
template <class FM> struct Reg;
template <> struct Reg<Fq> {
    typedef Fq29 type;
    __device__ __forceinline__ static Fq29 load(const Fq *p) { return Fq29::load(load_el(p)); }
    __device__ __forceinline__ static void store(Fq *p, const Fq29 &r) { store_el(p, Fq29::store(r)); }
    __device__ __forceinline__ static void store256(Fq *p, const Fq29 &r) { store_el(p, Fq29::to_mont256(r)); }
};
template <> struct Reg<Fq2> {
    typedef Fq2r type;
    __device__ __forceinline__ static Fq2r load(const Fq2 *p) { return Fq2r{Reg<Fq>::load(&p->a), Reg<Fq>::load(&p->b)}; }
    __device__ __forceinline__ static void store(Fq2 *p, const Fq2r &r) { Reg<Fq>::store(&p->a, r.a); Reg<Fq>::store(&p->b, r.b); }
    __device__ __forceinline__ static void store256(Fq2 *p, const Fq2r &r) { Reg<Fq>::store256(&p->a, r.a); Reg<Fq>::store256(&p->b, r.b); }
};
#define REGF typename Reg<F>::type

template <class F>
__device__ __forceinline__ Affine<REGF> load_affine(const Affine<F> *p) {
    return Affine<REGF>{Reg<F>::load(&p->x), Reg<F>::load(&p->y)};
}
template <class F>
__device__ __forceinline__ XYZZ<REGF> load_xyzz(const XYZZ<F> *p) {
    return XYZZ<REGF>{Reg<F>::load(&p->x), Reg<F>::load(&p->y), Reg<F>::load(&p->zz), Reg<F>::load(&p->zzz)};
}
template <class F>
__device__ __forceinline__ void store_xyzz(XYZZ<F> *p, const XYZZ<REGF> &v) {
    Reg<F>::store(&p->x, v.x);
    Reg<F>::store(&p->y, v.y);
    Reg<F>::store(&p->zz, v.zz);
    Reg<F>::store(&p->zzz, v.zzz);
}
// final window sums leave the device in the zkey's own 2^256 Montgomery form
template <class F>
__device__ __forceinline__ void store_xyzz_mont256(XYZZ<F> *p, const XYZZ<REGF> &v) {
    Reg<F>::store256(&p->x, v.x);
    Reg<F>::store256(&p->y, v.y);
    Reg<F>::store256(&p->zz, v.zz);
    Reg<F>::store256(&p->zzz, v.zzz);
}

template <class F>
__device__ __forceinline__ Affine<REGF> to_reg_affine(const Affine<F> &w);
template <>
__device__ __forceinline__ Affine<Fq29> to_reg_affine<Fq>(const Affine<Fq> &w) {
    return Affine<Fq29>{Fq29::load(w.x), Fq29::load(w.y)};
}
template <>
__device__ __forceinline__ Affine<Fq2r> to_reg_affine<Fq2>(const Affine<Fq2> &w) {
    return Affine<Fq2r>{Fq2r{Fq29::load(w.x.a), Fq29::load(w.x.b)}, Fq2r{Fq29::load(w.y.a), Fq29::load(w.y.b)}};
}

// zkey tables arrive as x*2^256; convert every coordinate to x*2^261 in place (once, at create)
__global__ __launch_bounds__(256) void k_fq_to_internal(Fq *coords, uint64_t n) {
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st)
        store_el(coords + i, Fq29::store(Fq29::from_mont256(load_el(coords + i))));
}
void launch_fq_to_internal(Fq *coords, uint64_t n, hipStream_t s) {
    if (!n) return;
    uint64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_fq_to_internal, dim3((uint32_t)g), dim3(256), 0, s, coords, n);
}

// ---------------------------------------------------------------- digits + LDS counting sort
// Signed c-bit digits d in [-2^(c-1), 2^(c-1) - 1] (a window value >= 2^(c-1) becomes negative
// and carries into the next window).  Digits are stored window-major as 16-bit codes:
// bit 15 = sign, bits 0..14 = |d| - 1, 0x7FFF = zero digit (+2^(c-1) never occurs, so that
// code is free even at c = 16).
//
// MI355X-specific: a whole window's histogram (2^(c-1) <= 32768 counters = 128 KiB) fits in
// one CU's 160 KiB LDS.  Workgroup (window w, slice s) histograms its slice of the scalars with
// LDS atomics only and writes counts[w][bucket][s]; one exclusive scan turns that into the
// start of every (bucket, slice) run; the scatter workgroups rank their entries with LDS
// atomics again.  No global atomics at all (the first version spent 83 % of its cycles
// waiting on them).  The order of entries inside a bucket depends on LDS arbitration; the sum
// does not.
#define DIGIT_ZERO 0x7FFFu
#define SORT_SLICES 16u
#define SORT_THREADS 1024u

__global__ __launch_bounds__(256) void k_msm_digits(uint16_t *digits, const Fr *scalars, uint64_t n, MsmPlan p) {
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t c = p.c, W = p.W;
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        Fr s = load_el(scalars + i);
        // any 256-bit value is < 6r: bring it below r (never loops for well-formed inputs)
        for (int k = 0; k < 6; k++) {
            Fr d;
            u32 bw = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) d.v[j] = subb(s.v[j], FrParams::P[j], bw);
            if (bw) break;
            s = d;
        }
        uint64_t buf = 0;
        uint32_t nb = 0, w = 0, carry = 0;
        auto emit = [&](uint32_t raw) {
            uint32_t d = raw + carry;
            const bool neg = d >= half;                  // digits in [-2^(c-1), 2^(c-1) - 1]
            carry = neg ? 1u : 0u;
            uint32_t mag = neg ? (1u << c) - d : d;      // 0 when raw = 2^c - 1 and carry = 1
            uint32_t code = mag ? ((mag - 1u) | (neg ? 0x8000u : 0u)) : DIGIT_ZERO;
            digits[(uint64_t)w * n + i] = (uint16_t)code;
            w++;
        };
#pragma unroll
        for (int k = 0; k < 8; k++) {
            buf |= (uint64_t)s.v[k] << nb;
            nb += 32;
            while (nb >= c && w + 1 < W) {
                emit((uint32_t)buf & mask);
                buf >>= c;
                nb -= c;
            }
        }
        emit((uint32_t)buf & mask);   // top window: value < 2^254 and W*c >= 256 => never negative
    }
}

__global__ __launch_bounds__(SORT_THREADS) void k_msm_count_lds(uint32_t *counts, const uint16_t *digits, uint64_t n, MsmPlan p) {
    extern __shared__ uint32_t hist[];
    const uint32_t w = blockIdx.x, slice = blockIdx.y, nb = p.nbuckets;
    for (uint32_t b = threadIdx.x; b < nb; b += SORT_THREADS) hist[b] = 0;
    __syncthreads();
    const uint64_t lo = n * slice / SORT_SLICES, hi = n * (slice + 1) / SORT_SLICES;
    const uint16_t *d = digits + (uint64_t)w * n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += SORT_THREADS) {
        uint32_t code = d[i];
        if (code != DIGIT_ZERO) atomicAdd(&hist[code & 0x7FFFu], 1u);
    }
    __syncthreads();
    uint32_t *out = counts + (uint64_t)w * nb * SORT_SLICES + slice;
    for (uint32_t b = threadIdx.x; b < nb; b += SORT_THREADS) out[(uint64_t)b * SORT_SLICES] = hist[b];
}

__global__ __launch_bounds__(SORT_THREADS) void k_msm_scatter_lds(uint32_t *entries, const uint32_t *starts, const uint16_t *digits,
                                                                   uint64_t n, MsmPlan p) {
    extern __shared__ uint32_t cursor[];
    const uint32_t w = blockIdx.x, slice = blockIdx.y, nb = p.nbuckets;
    const uint32_t *in = starts + (uint64_t)w * nb * SORT_SLICES + slice;
    for (uint32_t b = threadIdx.x; b < nb; b += SORT_THREADS) cursor[b] = in[(uint64_t)b * SORT_SLICES];
    __syncthreads();
    const uint64_t lo = n * slice / SORT_SLICES, hi = n * (slice + 1) / SORT_SLICES;
    const uint16_t *d = digits + (uint64_t)w * n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += SORT_THREADS) {
        uint32_t code = d[i];
        if (code != DIGIT_ZERO) {
            uint32_t pos = atomicAdd(&cursor[code & 0x7FFFu], 1u);
            entries[pos] = (uint32_t)i | ((code & 0x8000u) << 16);
        }
    }
}

// offsets[k] = start of bucket k = starts[k * SORT_SLICES]; offsets[total] = grand total
__global__ __launch_bounds__(256) void k_msm_compact_offsets(uint32_t *offsets, const uint32_t *starts, uint32_t total) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= total) offsets[k] = starts[(uint64_t)k * SORT_SLICES];
}

// Exclusive scan in three coalesced launches: per-block (4096 elements) local scan + block
// sums, scan of the block sums (one block), add-back.  offsets[total] = grand total.
#define SCAN_BLOCK 1024u
#define SCAN_ELEMS 4096u
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t &block_total) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    if (wave == 0) {
        uint32_t s = lane < (SCAN_BLOCK / 64) ? lds[lane] : 0;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            uint32_t y = __shfl_up(s, d);
            if (lane >= (uint32_t)d) s += y;
        }
        if (lane < (SCAN_BLOCK / 64)) lds[lane] = s;      // inclusive wave totals
    }
    __syncthreads();
    uint32_t wave_off = wave ? lds[wave - 1] : 0;
    block_total = lds[SCAN_BLOCK / 64 - 1];
    return wave_off + x - v;
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_local(uint32_t *offsets, uint32_t *block_sums, const uint32_t *counts, uint32_t total) {
    __shared__ uint32_t lds[SCAN_BLOCK / 64];
    const uint32_t base = blockIdx.x * SCAN_ELEMS + threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = base + j < total ? counts[base + j] : 0;
    uint32_t sum = v[0] + v[1] + v[2] + v[3], bt;
    uint32_t ex = block_exclusive_scan(sum, lds, bt);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (base + j < total) offsets[base + j] = ex;
        ex += v[j];
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = bt;
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_sums(uint32_t *block_sums, uint32_t nblocks, uint32_t *grand_total) {
    __shared__ uint32_t lds[SCAN_BLOCK / 64];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += SCAN_BLOCK) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < nblocks ? block_sums[i] : 0, bt;
        uint32_t ex = block_exclusive_scan(v, lds, bt) + carry_s;
        if (i < nblocks) block_sums[i] = ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += bt;
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand_total = carry_s;
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_add(uint32_t *offsets, const uint32_t *block_sums, uint32_t total) {
    const uint32_t base = blockIdx.x * SCAN_ELEMS + threadIdx.x * 4;
    const uint32_t add = block_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (base + j < total) offsets[base + j] += add;
}

// ---------------------------------------------------------------- load-balanced accumulation
// A lane-per-bucket walk is hopeless on real data: the top window of uniformly random
// scalars has only a handful of non-empty buckets (r ~ 2^253.6), and real witnesses pile
// half their entries into bucket "1" of window 0.  Instead EVERY lane adds exactly
// ACC_CHUNK consecutive entries of the bucket-sorted list, whatever buckets they span:
//   * a bucket run that starts and ends inside the chunk is complete -> buckets[b];
//   * a run cut by the chunk's left edge goes to the lane's HEAD slot, one cut by the right
//     edge to its TAIL slot (at most one of each), tagged with its bucket and STARTS/ENDS flags;
//   * the slot list (2 per lane, still bucket-sorted) is reduced by the same algorithm with
//     XYZZ inputs (k_msm_accum_ln), shrinking ~ACC_CHUNK_N/2 per level until one lane is left.
// Work per lane is constant, so the kernel time is flat in the scalar distribution.  The chunk
// is 128 entries, halved (to 32) for small or sharded MSMs so every SIMD still gets ~3 waves.
#define ACC_CHUNK_MAX 128u  // affine points per lane, level 1 (halved until >= ~3 waves/SIMD of lanes exist)
#define ACC_CHUNK_MIN 32u
#define ACC_CHUNK_N 32u     // slots per lane, levels >= 2
#define SLOT_EMPTY 0xffffffffu
#define FLAG_STARTS 1u
#define FLAG_ENDS 2u

template <class F>
__global__ __launch_bounds__(256) void k_msm_accum_l1(XYZZ<F> *buckets, const uint32_t *offsets, const uint32_t *entries,
                                                      const Affine<F> *points, uint32_t idx_min, uint32_t idx_sub,
                                                      uint32_t nbuckets_total, XYZZ<F> *out_part, uint32_t *out_key,
                                                      uint32_t *out_flag, uint32_t nlanes, uint32_t ACC_CHUNK) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlanes) return;
    const uint32_t E = offsets[nbuckets_total];
    const uint64_t lo64 = (uint64_t)t * ACC_CHUNK;
    uint32_t hkey = SLOT_EMPTY, tkey = SLOT_EMPTY, hflag = 0, tflag = 0;
    if (lo64 < E) {
        const uint32_t lo = (uint32_t)lo64;
        const uint32_t hi = (E - lo > ACC_CHUNK) ? lo + ACC_CHUNK : E;
        // last b with offsets[b] <= lo  (skips empty buckets that share the same offset)
        uint32_t bl = 0, br = nbuckets_total - 1;
        while (bl < br) {
            uint32_t mid = (bl + br + 1) >> 1;
            if (offsets[mid] <= lo) bl = mid; else br = mid - 1;
        }
        uint32_t b = bl;
        uint32_t bend = offsets[b + 1];
        bool started_before = offsets[b] < lo;
        typedef REGF FR;
        XYZZ<FR> acc = XYZZ<FR>::inf();
        Affine<F> nextP;                 // raw words: the next point is in flight while this one is added
        bool nextNeg = false, nextSkip = false;
        auto fetch = [&](uint32_t pos) {
            uint32_t ent = entries[pos];
            uint32_t idx = ent & 0x7fffffffu;
            nextNeg = (ent >> 31) != 0;
            nextSkip = idx < idx_min;
            const Affine<F> *src = points + (nextSkip ? 0 : idx - idx_sub);
            nextP.x = load_el(&src->x);
            nextP.y = load_el(&src->y);
        };
        uint32_t e = lo;
        fetch(e);
        while (e < hi) {
            Affine<F> Pw = nextP;
            bool ng = nextNeg, skip = nextSkip;
            e++;
            if (e < hi) fetch(e);
            if (!skip) {
                Affine<FR> P = to_reg_affine<F>(Pw);
                if (ng) P.y = FR::neg(P.y);
                madd(acc, P);
            }
            if (e == bend || e == hi) {              // the run of bucket b ends here (or is cut)
                const bool ends = (e == bend);
                if (!started_before && ends) {
                    store_xyzz(buckets + b, acc);
                } else if (started_before) {
                    store_xyzz(out_part + 2 * (uint64_t)t, acc);
                    hkey = b;
                    hflag = ends ? FLAG_ENDS : 0u;
                } else {
                    store_xyzz(out_part + 2 * (uint64_t)t + 1, acc);
                    tkey = b;
                    tflag = FLAG_STARTS;
                }
                if (e < hi) {                        // next non-empty bucket
                    do { b++; bend = offsets[b + 1]; } while (bend == e);
                    started_before = false;
                    acc = XYZZ<FR>::inf();
                }
            }
        }
    }
    out_key[2 * (uint64_t)t] = hkey;
    out_flag[2 * (uint64_t)t] = hflag;
    out_key[2 * (uint64_t)t + 1] = tkey;
    out_flag[2 * (uint64_t)t + 1] = tflag;
}

// Pairwise merge between level 1 and the generic levels: a bucket cut by exactly ONE chunk edge
// (the overwhelmingly common case — mean run length ~ chunk length) is the TAIL slot of lane t
// plus the HEAD slot of lane t+1 that also ENDS.  One general add per lane, full occupancy,
// and both slots are retired; what is left for the serial-ish generic levels is only the
// buckets spanning three or more chunks (top window, skewed witnesses).
template <class F>
__global__ __launch_bounds__(256) void k_msm_accum_pair(XYZZ<F> *buckets, const XYZZ<F> *part, uint32_t *key, const uint32_t *flag,
                                                        uint32_t nlanes) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t + 1 >= nlanes) return;
    const uint32_t kt = key[2 * (uint64_t)t + 1], kh = key[2 * (uint64_t)t + 2];
    if (kt == SLOT_EMPTY || kt != kh) return;
    if (!(flag[2 * (uint64_t)t + 2] & FLAG_ENDS)) return;       // continues further: generic levels
    typedef REGF FR;
    XYZZ<FR> a = load_xyzz(part + 2 * (uint64_t)t + 1);
    add(a, load_xyzz(part + 2 * (uint64_t)t + 2));
    store_xyzz(buckets + kt, a);
    key[2 * (uint64_t)t + 1] = SLOT_EMPTY;
    key[2 * (uint64_t)t + 2] = SLOT_EMPTY;
}

// Levels >= 2: the same chunked segmented sum over a bucket-sorted slot list of XYZZ partials.
template <class F>
__global__ __launch_bounds__(128) void k_msm_accum_ln(XYZZ<F> *buckets, const XYZZ<F> *in_part, const uint32_t *in_key,
                                                      const uint32_t *in_flag, uint32_t nitems, XYZZ<F> *out_part,
                                                      uint32_t *out_key, uint32_t *out_flag, uint32_t nlanes) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlanes) return;
    const uint32_t lo = t * ACC_CHUNK_N;
    const uint32_t hi = lo + ACC_CHUNK_N < nitems ? lo + ACC_CHUNK_N : nitems;
    uint32_t hkey = SLOT_EMPTY, tkey = SLOT_EMPTY, hflag = 0, tflag = 0;
    typedef REGF FR;
    uint32_t cur = SLOT_EMPTY, cflag = 0;
    XYZZ<FR> acc = XYZZ<FR>::inf();
    auto flush = [&]() {
        if (cur == SLOT_EMPTY) return;
        if ((cflag & FLAG_STARTS) && (cflag & FLAG_ENDS)) {
            store_xyzz(buckets + cur, acc);
        } else if (!(cflag & FLAG_STARTS)) {         // continues a bucket begun in an earlier lane
            store_xyzz(out_part + 2 * (uint64_t)t, acc);
            hkey = cur;
            hflag = cflag & FLAG_ENDS;
        } else {                                     // starts here, continues in a later lane
            store_xyzz(out_part + 2 * (uint64_t)t + 1, acc);
            tkey = cur;
            tflag = FLAG_STARTS;
        }
    };
    for (uint32_t i = lo; i < hi; i++) {
        uint32_t k = in_key[i];
        if (k == SLOT_EMPTY) continue;
        uint32_t fl = in_flag[i];
        if (k != cur) {
            flush();
            cur = k;
            cflag = fl & FLAG_STARTS;
            acc = XYZZ<FR>::inf();
        }
        cflag = (cflag & FLAG_STARTS) | (fl & FLAG_ENDS);
        add(acc, load_xyzz(in_part + i));
    }
    flush();
    out_key[2 * (uint64_t)t] = hkey;
    out_flag[2 * (uint64_t)t] = hflag;
    out_key[2 * (uint64_t)t + 1] = tkey;
    out_flag[2 * (uint64_t)t + 1] = tflag;
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
    typedef REGF FR;
    XYZZ<FR> run = XYZZ<FR>::inf(), sum = XYZZ<FR>::inf();
    for (int j = (int)chunk - 1; j >= 0; j--) {
        add(run, load_xyzz(B + j));
        add(sum, run);
    }
    // sum += (cw*chunk) * run   — double-and-add, MSB first
    uint32_t k = cw * chunk;
    if (k) {
        XYZZ<FR> m = XYZZ<FR>::inf();
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
    typedef REGF FR;
    XYZZ<FR> *lds = reinterpret_cast<XYZZ<FR> *>(lds_raw);
    const XYZZ<F> *X = scratch + (uint64_t)blockIdx.x * chunks_per_window;
    XYZZ<FR> acc = XYZZ<FR>::inf();
    for (uint32_t i = threadIdx.x; i < chunks_per_window; i += REDUCE_THREADS) add(acc, load_xyzz(X + i));
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = REDUCE_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            XYZZ<FR> o = lds[threadIdx.x + s];
            add(acc, o);
            lds[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) store_xyzz_mont256(window_sums + blockIdx.x, acc);    // back to the zkey's 2^256 form
}

static inline uint32_t reduce_chunk_for(MsmPlan p) { return p.nbuckets < REDUCE_CHUNK ? p.nbuckets : REDUCE_CHUNK; }

uint64_t msm_reduce_scratch_points(uint32_t n_msm, MsmPlan p) {
    return (uint64_t)n_msm * p.W * (p.nbuckets / reduce_chunk_for(p));
}

// exclusive scan of counts[0..total) -> out[0..total], out[total] = grand total;
// out must hold total + 1 + msm_scan_extra_words(total) words (block sums live past the end)
static void launch_scan(uint32_t *out, const uint32_t *counts, uint32_t total, hipStream_t s) {
    uint32_t nblocks = (total + SCAN_ELEMS - 1) / SCAN_ELEMS;
    uint32_t *block_sums = out + total + 1;
    hipLaunchKernelGGL(k_scan_local, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, out, block_sums, counts, total);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, block_sums, nblocks, out + total);
    hipLaunchKernelGGL(k_scan_add, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, out, (const uint32_t *)block_sums, total);
}
uint32_t msm_scan_extra_words(uint32_t total) { return (total + SCAN_ELEMS - 1) / SCAN_ELEMS; }

MsmSortSizes msm_sort_sizes(uint64_t n, MsmPlan p) {
    MsmSortSizes z;
    uint64_t tb = (uint64_t)p.W * p.nbuckets;
    z.digits_u16 = (n ? n : 1) * p.W;
    z.counts_u32 = tb * SORT_SLICES;
    z.starts_u32 = tb * SORT_SLICES + 1 + msm_scan_extra_words((uint32_t)(tb * SORT_SLICES));
    z.offsets_u32 = tb + 1;
    z.entries_u32 = (n ? n : 1) * p.W;
    return z;
}

// digits -> LDS histograms -> scan -> LDS-ranked scatter -> compact bucket offsets
void launch_msm_sort(uint32_t *offsets, uint32_t *entries, uint16_t *digits, uint32_t *counts, uint32_t *starts,
                     const Fr *scalars, uint64_t n, MsmPlan p, hipStream_t s) {
    const uint32_t tb = p.W * p.nbuckets;
    const size_t lds = (size_t)p.nbuckets * 4;
    static bool attr_set = false;
    if (!attr_set) {   // > 64 KiB of dynamic LDS needs the opt-in (160 KiB per CU on gfx950)
        (void)hipFuncSetAttribute((const void *)k_msm_count_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_msm_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (n) {
        uint64_t g = (n + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(k_msm_digits, dim3((uint32_t)g), dim3(256), 0, s, digits, scalars, n, p);
    }
    hipLaunchKernelGGL(k_msm_count_lds, dim3(p.W, SORT_SLICES), dim3(SORT_THREADS), lds, s, counts, (const uint16_t *)digits, n, p);
    launch_scan(starts, counts, tb * SORT_SLICES, s);
    hipLaunchKernelGGL(k_msm_scatter_lds, dim3(p.W, SORT_SLICES), dim3(SORT_THREADS), lds, s, entries, (const uint32_t *)starts,
                       (const uint16_t *)digits, n, p);
    hipLaunchKernelGGL(k_msm_compact_offsets, dim3((tb + 256) / 256), dim3(256), 0, s, offsets, (const uint32_t *)starts, tb);
}

// workspace: level-1 slots (2 per lane) + level-2 slots + ... (geometric: < 2.2x level 1)
static inline uint32_t accum_chunk_for(uint64_t max_entries) {
    uint32_t chunk = ACC_CHUNK_MAX;
    while (chunk > ACC_CHUNK_MIN && max_entries / chunk < 3u * 1024u * 64u) chunk >>= 1;   // 256 CUs x 4 SIMDs x 3 waves
    return chunk;
}
static inline uint64_t accum_l1_lanes(uint64_t max_entries) {
    uint32_t chunk = accum_chunk_for(max_entries);
    return (max_entries + chunk - 1) / chunk;
}
uint64_t msm_accum_workspace_slots(uint64_t max_entries) {
    uint64_t lanes = accum_l1_lanes(max_entries ? max_entries : 1);
    uint64_t total = 0;
    for (;;) {
        uint64_t slots = 2 * lanes;
        total += slots;
        if (lanes == 1) break;
        lanes = (slots + ACC_CHUNK_N - 1) / ACC_CHUNK_N;
    }
    return total;
}

template <class F>
static void launch_accum(XYZZ<F> *buckets, const uint32_t *offsets, const uint32_t *entries, const Affine<F> *points,
                         uint32_t idx_min, uint32_t idx_sub, uint32_t total_buckets, uint64_t max_entries,
                         XYZZ<F> *ws_part, uint32_t *ws_key, uint32_t *ws_flag, hipStream_t s, hipEvent_t *ev) {
    // empty buckets are never written by the kernels: infinity is the all-zero pattern
    (void)hipMemsetAsync(buckets, 0, (size_t)total_buckets * sizeof(XYZZ<F>), s);
    uint64_t lanes = accum_l1_lanes(max_entries ? max_entries : 1);
    if (ev) (void)hipEventRecord(ev[0], s);           // tight bracket around the level-1 kernel (roofline timing)
    hipLaunchKernelGGL(k_msm_accum_l1<F>, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, s, buckets, offsets, entries,
                       points, idx_min, idx_sub, total_buckets, ws_part, ws_key, ws_flag, (uint32_t)lanes,
                       accum_chunk_for(max_entries ? max_entries : 1));
    if (ev) (void)hipEventRecord(ev[1], s);
    if (lanes > 1)
        hipLaunchKernelGGL(k_msm_accum_pair<F>, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, s, buckets,
                           (const XYZZ<F> *)ws_part, ws_key, (const uint32_t *)ws_flag, (uint32_t)lanes);
    uint64_t off = 0;
    while (lanes > 1) {          // a single lane has no cut runs: everything it saw was complete
        uint64_t items = 2 * lanes;
        uint64_t nl = (items + ACC_CHUNK_N - 1) / ACC_CHUNK_N;
        uint64_t noff = off + items;
        hipLaunchKernelGGL(k_msm_accum_ln<F>, dim3((uint32_t)((nl + 127) / 128)), dim3(128), 0, s, buckets, ws_part + off,
                           ws_key + off, ws_flag + off, (uint32_t)items, ws_part + noff, ws_key + noff, ws_flag + noff, (uint32_t)nl);
        off = noff;
        lanes = nl;
    }
}

void launch_msm_accum_g1(G1XYZZ *buckets, const uint32_t *offsets, const uint32_t *entries, const G1Affine *points,
                         uint32_t idx_min, uint32_t idx_sub, uint32_t total, uint64_t max_entries, G1XYZZ *ws_part,
                         uint32_t *ws_key, uint32_t *ws_flag, hipStream_t s, hipEvent_t *ev) {
    launch_accum<Fq>(buckets, offsets, entries, points, idx_min, idx_sub, total, max_entries, ws_part, ws_key, ws_flag, s, ev);
}
void launch_msm_accum_g2(G2XYZZ *buckets, const uint32_t *offsets, const uint32_t *entries, const G2Affine *points,
                         uint32_t idx_min, uint32_t idx_sub, uint32_t total, uint64_t max_entries, G2XYZZ *ws_part,
                         uint32_t *ws_key, uint32_t *ws_flag, hipStream_t s, hipEvent_t *ev) {
    launch_accum<Fq2>(buckets, offsets, entries, points, idx_min, idx_sub, total, max_entries, ws_part, ws_key, ws_flag, s, ev);
}

template <class F>
static void launch_reduce(XYZZ<F> *window_sums, XYZZ<F> *scratch, const XYZZ<F> *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s) {
    uint32_t chunk = reduce_chunk_for(p);
    uint32_t cpw = p.nbuckets / chunk;
    uint32_t total_chunks = n_msm * p.W * cpw;
    hipLaunchKernelGGL(k_msm_reduce_chunks<F>, dim3((total_chunks + 127) / 128), dim3(128), 0, s, scratch, buckets, p.nbuckets, chunk, total_chunks);
    hipLaunchKernelGGL(k_msm_reduce_final<F>, dim3(n_msm * p.W), dim3(REDUCE_THREADS), REDUCE_THREADS * sizeof(XYZZ<typename Reg<F>::type>), s, window_sums, scratch, cpw);
}
void launch_msm_reduce_g1(G1XYZZ *ws, G1XYZZ *scratch, const G1XYZZ *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s) {
    launch_reduce<Fq>(ws, scratch, buckets, n_msm, p, s);
}
void launch_msm_reduce_g2(G2XYZZ *ws, G2XYZZ *scratch, const G2XYZZ *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s) {
    launch_reduce<Fq2>(ws, scratch, buckets, n_msm, p, s);
}

}   // namespace zk
