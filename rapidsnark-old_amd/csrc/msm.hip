// Pippenger multi-scalar multiplication over BN254 G1 / G2 for gfx950 — replaces ffiasm
// ParallelMultiexp behind Curve::multiMulByScalar (call sites src/groth16.cpp:173,183,190,197,204).
//
// MI355X design (ffiasm keeps nThreads x 2^c per-thread bucket arrays; that makes no sense here):
//   1. k_msm_digits       : signed c-bit digits of every scalar, window-major 32-bit codes (bit 31 = sign, low bits =
//      bucket key; with window-precomputed tables every window shares ONE bucket set, and a batch of B scalar vectors
//      — several small proofs in one set of launches — gets one set per vector)
//   2. k_bin_count / k_bin_scatter : partition the codes by their high key bits into <= 256 bins (LDS-staged);
//      k_bin_count_lds / k_bin_scatter_lds : counting sort of every bin with its histogram in LDS; k_scan_* between
//      (1-2 run ONCE per scalar vector: the witness sort is shared by MSM A, B1, B2, C,
//       which the reference recomputes four times, src/groth16.cpp:183-204)
//   3. k_msm_accum_l1 / _l1_g2s : load-balanced segmented accumulation — every lane mixed-adds an equal share of
//      the bucket-sorted list into XYZZ accumulators in VGPRs (next point prefetched; G2 split across lane pairs);
//      k_msm_accum_pair / _wave : runs cut by chunk edges are merged by wave-parallel segmented scans, level by level
//   4. bucket reduction sum_k (k+1)*B_k per bucket set: k_msm_reduce_chunks / _tree (chunked running sums + LDS tree)
//      for large sets, k_msm_reduce_bits_block / _top (one binary tree of bit sums, c-1 additions deep) for small ones
//   5. host (host_tail.cpp): the serial rest — Horner over the windows (plain tables) or over the c bit sums, and the
//      final assembly: serial doublings are 30x faster on one CPU core than on one GPU lane.
// Signed digits halve the bucket count; scalars are reduced mod r first so any 256-bit
// input is accepted like the reference's raw-byte interface.
#include <stdlib.h>
#include "kernels.hpp"
#include "hipcheck.hpp"
#include "common.hpp"
#include "field29.hpp"
#include "curve29.hpp"
#include <string.h>

namespace zk {

#define REDUCE_CHUNK 16u
#define REDUCE_THREADS 256u

MsmPlan make_msm_plan(uint64_t n, uint32_t window_bits, uint32_t precomp, uint32_t batch) {
    if (precomp > 2) throw std::invalid_argument("table mode: 0 (as in the zkey), 1 (a row per window) or 2 (a row per second window)");
    MsmPlan p;
    uint32_t lg = 0;
    while ((1ull << (lg + 1)) <= n) lg++;
    uint32_t c = window_bits;
    if (precomp) {
        // all windows share one bucket set (tables hold 2^(c*j) P): the reduction is paid once, so
        // the window can grow until ~100 entries per bucket remain.  c = 17 buys nothing (W = 16).
        if (c == 0) {
            c = lg > 2 ? lg - 2 : 2;
            if (lg == 20) c = 19;     // 14 windows instead of 15 pay for four times the buckets only here (2^20: 9.40 -> 9.16 ms per proof;
                                      // 2^16 .. 2^19 measured neutral or worse with a wider window)
            if (lg == 21) c = 20;     // 13 instead of 14: pays since the split bucket reduction (17.0 -> 16.55 ms, profiles/r05ze_window_sweep.txt)
            if (batch > 1) c++;       // a batch shares the fixed costs of a set of launches: one window fewer pays (2^16 x 8: 0.76 -> 0.69 ms per proof)
            if (c > 20) c = 20;       // the size-based choice stops at 2^19 buckets per set: a 22-bit window (12 additions per point) was measured at
                                      // 2^24, where the reductions are cheapest — see the cap below
        }
        if (c > 22) c = 22;           // an explicit width may go to 22 (2^21 buckets per set: 8192-bucket bins in the sort's second level, 64-bucket
                                      // lanes in the split reduction; parity-tested) — it does not pay: profiles/r06t_ab_window_22_at_2p24.txt
    } else {
        if (c == 0) c = lg > 6 ? lg - 6 : 2;     // ~128 points per bucket on random scalars
        if (c > 16) c = 16;                      // one window's histogram must fit one CU's LDS
    }
    if (c < 2) c = 2;
    p.c = c;
    p.W = (256 + c - 1) / c;        // W*c >= 256: the top digit is never negative
    {
        // Scalars are reduced below r < 2^254 first (k_msm_digits), so 255 bits (254 + the carry of the signed recoding) are
        // enough — IF the top window's largest value, r >> ((W-1)*c), plus a carry still stays below 2^(c-1) (then its digit is
        // never negative).  That saves a window exactly when c divides 255: c = 15 (W 18 -> 17) and c = 17 (16 -> 15); for
        // c = 3 the check fails (r >> 252 = 3) and the 256-bit rule stays.
        const uint32_t W255 = (255 + c - 1) / c;
        if (W255 < p.W) {
            const uint32_t sh = (W255 - 1) * c;                   // < 256
            uint64_t top = 0;                                     // r >> sh (fits: 254 - sh <= c - 1 <= 19 bits)
            for (int k = 7; k >= 0; k--) {
                const int lo_bit = 32 * k;
                if (lo_bit + 32 <= (int)sh) break;
                const uint64_t w = FrParams::P[k];
                top |= lo_bit >= (int)sh ? w << (lo_bit - sh) : w >> (sh - lo_bit);
            }
            if (top + 1 < (1ull << (c - 1))) p.W = W255;
        }
    }
    p.nbuckets = 1u << (c - 1);
    p.precomp = precomp;
    p.sets = precomp ? precomp : p.W;
    p.batch = 1;
    p.batch_n = 0;
    if (batch > 1) {
        if (precomp != 1) throw std::invalid_argument("batched MSMs need window-precomputed tables with a row per window");
        p.batch = batch;
        p.batch_n = (uint32_t)n;
        p.sets = batch;
    }
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
// a table row is read exactly once per MSM: ZK_L1_NT_GATHER (measurement builds) loads it with the non-temporal hint
template <class F>
__device__ __forceinline__ F load_row_el(const F *p) {
#if defined(ZK_L1_NT_GATHER)
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u *q = reinterpret_cast<const v4u *>(p);
    v4u lo = __builtin_nontemporal_load(q), hi = __builtin_nontemporal_load(q + 1);
    F r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
#else
    return load_el(p);
#endif
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
    ZK_LAUNCH(k_fq_to_internal, dim3((uint32_t)g), dim3(256), 0, s, coords, n);
    ZK_LAUNCH_OK("fq_to_internal");
}

// Bucket sums, partial sums and reduction scratch live in HBM as the ACCUMULATORS' OWN LIMBS (G1Acc /
// G2Acc, kernels.hpp): 36 (72) int32, moved as nine 16-byte accesses per lane.  Storing them as canonical
// 256-bit words cost ~95 instructions per coordinate (exact reduction, conditional +p, limb -> word
// packing) in the most divergent spot of the level-1 loop — a bucket run ends in ~46 % of a wave's
// iterations (64 lanes, ~104 entries per bucket), and the other 63 lanes wait — and the same again to
// unpack at every load.  Lazy values are valid operands everywhere (field29.hpp); infinity stays the
// all-zero pattern.
__device__ __forceinline__ void load36(int32_t *dst, const int32_t *src) {
    const uint4 *q = reinterpret_cast<const uint4 *>(src);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint4 t = q[i];
        dst[4 * i] = (int32_t)t.x; dst[4 * i + 1] = (int32_t)t.y; dst[4 * i + 2] = (int32_t)t.z; dst[4 * i + 3] = (int32_t)t.w;
    }
}
__device__ __forceinline__ void store36(int32_t *dst, const int32_t *src) {
    uint4 *q = reinterpret_cast<uint4 *>(dst);
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = make_uint4((uint32_t)src[4 * i], (uint32_t)src[4 * i + 1], (uint32_t)src[4 * i + 2], (uint32_t)src[4 * i + 3]);
}
__device__ __forceinline__ XYZZ<Fq29> unpack36(const int32_t *w) {
    XYZZ<Fq29> v;
#pragma unroll
    for (int i = 0; i < 9; i++) { v.x.l[i] = w[i]; v.y.l[i] = w[9 + i]; v.zz.l[i] = w[18 + i]; v.zzz.l[i] = w[27 + i]; }
    return v;
}
__device__ __forceinline__ void pack36(int32_t *w, const Fq29 &x, const Fq29 &y, const Fq29 &zz, const Fq29 &zzz) {
#pragma unroll
    for (int i = 0; i < 9; i++) { w[i] = x.l[i]; w[9 + i] = y.l[i]; w[18 + i] = zz.l[i]; w[27 + i] = zzz.l[i]; }
}

// Lane model of the merge / reduction kernels: one lane per G1 element; a lane PAIR per G2 element
// (even lane = real components, odd lane = imaginary components, curve29.hpp Fq2s) — half the
// registers per lane, twice the lanes, no spills.
template <class F> struct LaneModel;
template <> struct LaneModel<Fq> {
    static constexpr uint32_t LPE = 1;      // lanes per element
    typedef Fq29 R;
    typedef G1Acc Mem;
    __device__ __forceinline__ static XYZZ<R> load(const G1Acc *p) {
        int32_t w[36];
        load36(w, p->l);
        return unpack36(w);
    }
    __device__ __forceinline__ static void store(G1Acc *p, const XYZZ<R> &v) {
        int32_t w[36];
        pack36(w, v.x, v.y, v.zz, v.zzz);
        store36(p->l, w);
    }
    __device__ __forceinline__ static void store256(G1XYZZ *p, const XYZZ<R> &v) { store_xyzz_mont256(p, v); }
};
template <> struct LaneModel<Fq2> {
    static constexpr uint32_t LPE = 2;
    typedef Fq2s R;
    typedef G2Acc Mem;
    __device__ __forceinline__ static XYZZ<R> load(const G2Acc *p) {      // memory: [component][x | y | zz | zzz][limb]
        int32_t w[36];
        load36(w, p->l + 36 * (threadIdx.x & 1u));
        const XYZZ<Fq29> t = unpack36(w);
        return XYZZ<R>{R{t.x}, R{t.y}, R{t.zz}, R{t.zzz}};
    }
    __device__ __forceinline__ static void store(G2Acc *p, const XYZZ<R> &v) {
        int32_t w[36];
        pack36(w, v.x.v, v.y.v, v.zz.v, v.zzz.v);
        store36(p->l + 36 * (threadIdx.x & 1u), w);
    }
    __device__ __forceinline__ static void store256(G2XYZZ *p, const XYZZ<R> &v) {   // x.a x.b y.a y.b zz.a zz.b zzz.a zzz.b, canonical words
        Fq *c = reinterpret_cast<Fq *>(p) + (threadIdx.x & 1u);
        Reg<Fq>::store256(c, v.x.v); Reg<Fq>::store256(c + 2, v.y.v); Reg<Fq>::store256(c + 4, v.zz.v); Reg<Fq>::store256(c + 6, v.zzz.v);
    }
};
#define ACCMEM typename LaneModel<F>::Mem

// ---------------------------------------------------------------- digits + two-level LDS counting sort
// Signed c-bit digits d in [-2^(c-1), 2^(c-1) - 1] (a window value >= 2^(c-1) becomes negative
// and carries into the next window; +2^(c-1) never occurs).  Every non-zero digit becomes one
// entry (table row | sign) keyed by its bucket; the sort brings the entries into bucket order.
//
// The key space (sets * 2^(c-1) buckets, 2^19 at 2^22) is sorted in two LDS-only levels:
//   1. k_bin_count / k_bin_scatter partition the codes by their high key bits into <= 256 bins,
//      staged through LDS so that the partitioned copy is written in coalesced runs;
//   2. k_bin_count_lds / k_bin_scatter_lds counting-sort every bin (<= 2^15 buckets, normally
//      2^11) with its histogram in LDS, `slices` workgroups per bin.
// No global atomics at all (the first version spent 83 % of its cycles waiting on them).  The
// order of entries inside a bucket depends on LDS arbitration; the sum does not.
#define CODE32_ZERO 0x7FFFFFFFu
#ifndef SORT_THREADS
#define SORT_THREADS 1024u      // (512 / 256 in measurement builds: workgroups that fit beside a level-1 launch's waves)
#endif

// 32-bit codes, window-major: bit 31 = sign, bits 0..30 = bucket key, 0x7FFFFFFF = zero digit.
// key = (|d| - 1) + w * nbuckets with per-window bucket sets, |d| - 1 with window-precomputed
// tables (one shared set).
__global__ __launch_bounds__(256) void k_msm_digits(uint32_t *digits, const Fr *scalars, uint64_t n, MsmPlan p) {
    ZK_CHAIN_PRIO();
    uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t c = p.c, W = p.W;
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    // bucket set of window w: its own with plain tables, the one shared set with a row per window, set w & 1 with a row per second window
    const uint32_t set_stride = p.precomp == 1 ? 0u : p.nbuckets, set_mask = p.precomp == 2 ? 1u : 0xffffffffu;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        Fr s = load_el(scalars + i);
        const uint32_t vec_base = p.batch > 1 ? (uint32_t)(i / p.batch_n) * p.nbuckets : 0u;      // bucket set of this scalar's vector
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
            uint32_t code = mag ? ((mag - 1u + (w & set_mask) * set_stride + vec_base) | (neg ? 0x80000000u : 0u)) : CODE32_ZERO;
            digits[(uint64_t)w * n + i] = code;
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

// offsets[k] = start of bucket k = starts[k * slices]; offsets[total] = grand total
__global__ __launch_bounds__(256) void k_msm_compact_offsets(uint32_t *offsets, const uint32_t *starts, uint32_t total, uint32_t slices) {
    ZK_CHAIN_PRIO();
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= total) offsets[k] = starts[(uint64_t)k * slices];
}

// The bins are SMALL (2^11 buckets) on purpose: the second-level scatter writes 4-byte entries
// at random inside its bin's output range, and only when the ranges being written at one time
// fit the XCD's 4 MiB L2 do those writes leave the L2 as whole lines (measured at 2^22: 1.76 ms
// with 2^15-bucket bins, 0.48 ms with 2^11).  The workgroups of one bin therefore run on one XCD
// (block id -> XCD is round-robin) and each XCD walks its bins in order, `slices` workgroups at
// a time.
#define BIN_MAX 256u

__global__ __launch_bounds__(SORT_THREADS) void k_bin_count(uint32_t *bin_counts, const uint32_t *codes, uint64_t total, uint32_t nbins,
                                                            uint32_t nblocks, uint32_t shift, uint32_t span) {
    ZK_CHAIN_PRIO();
    __shared__ uint32_t hist[BIN_MAX];
    if (threadIdx.x < BIN_MAX) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * span;
    for (uint32_t k = threadIdx.x; k < span; k += SORT_THREADS) {
        uint64_t i = base + k;
        if (i < total) {
            uint32_t code = codes[i];
            if ((code & 0x7FFFFFFFu) != CODE32_ZERO) atomicAdd(&hist[(code & 0x7FFFFFFFu) >> shift], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < nbins) bin_counts[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// First-level partition, staged through LDS: a workgroup ranks its `span` items per bin with LDS
// atomics, lays them out bin-major in LDS, and writes them out so that consecutive lanes hit
// consecutive addresses.  (Scattering straight from registers costs one cache-line request per
// lane per store — 109 M line requests per sort at 2^22 — and was the larger half of the sort.)
#define BIN_ITEMS 8u           // items per thread; span = BIN_ITEMS * SORT_THREADS
__global__ __launch_bounds__(SORT_THREADS) void k_bin_scatter(uint16_t *lo, uint32_t *val, const uint32_t *bin_starts, const uint32_t *codes,
                                                              uint64_t total, uint32_t nbins, uint32_t nblocks, uint32_t shift, uint32_t span, uint64_t n,
                                                              uint32_t set_shift, uint32_t batch_n, uint32_t tstride) {
    ZK_CHAIN_PRIO();
    extern __shared__ uint32_t smem[];
    uint32_t *cnt = smem;                         // [BIN_MAX] per-bin count, then LDS start
    uint32_t *gdelta = smem + BIN_MAX;            // [BIN_MAX] global start - LDS start
    uint32_t *st_dst = smem + 2 * BIN_MAX;        // [span]
    uint32_t *st_val = st_dst + span;             // [span]
    uint16_t *st_lo = (uint16_t *)(st_val + span);   // [span]
    const uint32_t tid = threadIdx.x;
    if (tid < BIN_MAX) cnt[tid] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * span;
    const uint32_t lomask = (1u << shift) - 1u;
    // tables with a row per `tstride` windows: item w * n + i reads row (w / tstride) * n + i.  The block's first window by one
    // (uniform) division, the items' by comparison.
    const uint64_t w_base = tstride > 1 && n ? base / n : 0, r_base = tstride > 1 && n ? base - w_base * n : 0;      // (n = 0: an empty shard — no item is used)
    uint32_t code[BIN_ITEMS], rank[BIN_ITEMS];
#pragma unroll
    for (uint32_t k = 0; k < BIN_ITEMS; k++) {
        uint64_t i = base + (uint64_t)k * SORT_THREADS + tid;
        code[k] = i < total ? codes[i] : CODE32_ZERO;
    }
#pragma unroll
    for (uint32_t k = 0; k < BIN_ITEMS; k++) {
        uint32_t mag = code[k] & 0x7FFFFFFFu;
        rank[k] = mag != CODE32_ZERO ? atomicAdd(&cnt[mag >> shift], 1u) : 0u;
    }
    __syncthreads();
    // exclusive scan of the (<= 256) bin counts by wave 0: four bins per lane
    if (tid < 64) {
        uint32_t c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
        uint32_t sum = c0 + c1 + c2 + c3, x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(x, d);
            if (tid >= (uint32_t)d) x += y;
        }
        uint32_t off = x - sum;
        uint32_t o[4] = {off, off + c0, off + c0 + c1, off + c0 + c1 + c2};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t bin = 4 * tid + q;
            cnt[bin] = o[q];
            gdelta[bin] = bin < nbins ? bin_starts[(uint64_t)bin * nblocks + blockIdx.x] - o[q] : 0u;
        }
        if (tid == 63) smem[2 * BIN_MAX + 2 * span + span / 2] = x;      // total kept past st_lo
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < BIN_ITEMS; k++) {
        uint32_t mag = code[k] & 0x7FFFFFFFu;
        if (mag != CODE32_ZERO) {
            uint32_t bin = mag >> shift, slot = cnt[bin] + rank[k];
            // table row: the flattened index j*n + i itself with window-precomputed tables
            // (set_shift = 32), the point index i otherwise (key >> set_shift = window j)
            uint64_t i = base + (uint64_t)k * SORT_THREADS + tid - (uint64_t)(set_shift < 32 ? mag >> set_shift : 0u) * n;
            if (tstride > 1) {
                uint64_t w = w_base, r = r_base + (uint64_t)k * SORT_THREADS + tid;
                while (n && r >= n) { r -= n; w++; }
                i = (w / tstride) * n + r;
            }
            if (batch_n) {          // batched vectors share the table: flattened j * n + (v * batch_n + r)  ->  row j * batch_n + r
                const uint64_t j = i / n, r = (i % n) % batch_n;
                i = j * batch_n + r;
            }
            st_dst[slot] = slot + gdelta[bin];
            st_val[slot] = (uint32_t)i | (code[k] & 0x80000000u);
            st_lo[slot] = (uint16_t)(mag & lomask);
        }
    }
    __syncthreads();
    const uint32_t kept = smem[2 * BIN_MAX + 2 * span + span / 2];
    for (uint32_t sidx = tid; sidx < kept; sidx += SORT_THREADS) {
        uint32_t d = st_dst[sidx];
        val[d] = st_val[sidx];
        lo[d] = st_lo[sidx];
    }
}

// 1-D grid -> (bin, slice): XCD x (= block id mod 8) owns bins x, x+8, x+16, ... and walks them
// in dispatch order, `slices` consecutive workgroups per bin.
__device__ __forceinline__ bool bin_slice_of_block(uint32_t nbins, uint32_t slices, uint32_t &bin, uint32_t &slice) {
    const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
    bin = (k / slices) * 8u + xcd;
    slice = k % slices;
    return bin < nbins;
}

// bin b occupies items [bin_starts[b*nblocks], bin_starts[(b+1)*nblocks]) (the scan array ends with the total)
__global__ __launch_bounds__(SORT_THREADS) void k_bin_count_lds(uint32_t *counts, const uint16_t *lo, const uint32_t *bin_starts,
                                                                uint32_t nblocks, uint32_t buckets_per_bin, uint32_t nbins, uint32_t slices,
                                                                uint32_t total_buckets) {
    ZK_CHAIN_PRIO();
    extern __shared__ uint32_t hist[];
    uint32_t b, slice;
    if (!bin_slice_of_block(nbins, slices, b, slice)) return;
    const uint32_t first = b * buckets_per_bin, nb = total_buckets - first < buckets_per_bin ? total_buckets - first : buckets_per_bin;
    for (uint32_t k = threadIdx.x; k < nb; k += SORT_THREADS) hist[k] = 0;
    __syncthreads();
    const uint64_t bs = bin_starts[(uint64_t)b * nblocks], be = bin_starts[(uint64_t)(b + 1) * nblocks];
    const uint64_t len = be - bs, s0 = bs + len * slice / slices, s1 = bs + len * (slice + 1) / slices;
    for (uint64_t i = s0 + threadIdx.x; i < s1; i += SORT_THREADS) atomicAdd(&hist[lo[i]], 1u);
    __syncthreads();
    uint32_t *out = counts + (uint64_t)first * slices + slice;
    for (uint32_t k = threadIdx.x; k < nb; k += SORT_THREADS) out[(uint64_t)k * slices] = hist[k];
}

__global__ __launch_bounds__(SORT_THREADS) void k_bin_scatter_lds(uint32_t *entries, const uint32_t *starts, const uint16_t *lo,
                                                                  const uint32_t *val, const uint32_t *bin_starts, uint32_t nblocks,
                                                                  uint32_t buckets_per_bin, uint32_t nbins, uint32_t slices, uint32_t total_buckets) {
    ZK_CHAIN_PRIO();
    extern __shared__ uint32_t cursor[];
    uint32_t b, slice;
    if (!bin_slice_of_block(nbins, slices, b, slice)) return;
    const uint32_t first = b * buckets_per_bin, nb = total_buckets - first < buckets_per_bin ? total_buckets - first : buckets_per_bin;
    const uint32_t *in = starts + (uint64_t)first * slices + slice;
    for (uint32_t k = threadIdx.x; k < nb; k += SORT_THREADS) cursor[k] = in[(uint64_t)k * slices];
    __syncthreads();
    const uint64_t bs = bin_starts[(uint64_t)b * nblocks], be = bin_starts[(uint64_t)(b + 1) * nblocks];
    const uint64_t len = be - bs, s0 = bs + len * slice / slices, s1 = bs + len * (slice + 1) / slices;
    uint64_t i = s0 + threadIdx.x;
    for (; i + 3 * SORT_THREADS < s1; i += 4 * SORT_THREADS) {      // four independent loads in flight
        uint32_t k0 = lo[i], k1 = lo[i + SORT_THREADS], k2 = lo[i + 2 * SORT_THREADS], k3 = lo[i + 3 * SORT_THREADS];
        uint32_t v0 = val[i], v1 = val[i + SORT_THREADS], v2 = val[i + 2 * SORT_THREADS], v3 = val[i + 3 * SORT_THREADS];
        entries[atomicAdd(&cursor[k0], 1u)] = v0;
        entries[atomicAdd(&cursor[k1], 1u)] = v1;
        entries[atomicAdd(&cursor[k2], 1u)] = v2;
        entries[atomicAdd(&cursor[k3], 1u)] = v3;
    }
    for (; i < s1; i += SORT_THREADS) entries[atomicAdd(&cursor[lo[i]], 1u)] = val[i];
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
    ZK_CHAIN_PRIO();
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
    ZK_CHAIN_PRIO();
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
    ZK_CHAIN_PRIO();
    const uint32_t base = blockIdx.x * SCAN_ELEMS + threadIdx.x * 4;
    const uint32_t add = block_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (base + j < total) offsets[base + j] += add;
}

// Second-level scatter, STAGED: the direct version above sends every entry to L2 as a 4-byte request of its own (54.5 M requests per
// sort at 2^22: 55 % of its wave cycles stalled at issue, profiles/r05u_sort_kernel_counters.txt).  Here a workgroup takes its slice of
// the bin in tiles of STAGE_CAP entries, counting-sorts a tile INSIDE LDS (rank by LDS atomic, exclusive scan of the 2^11 bucket counts,
// placement), and writes it out in bucket order: the entries of one bucket go to consecutive addresses, so consecutive lanes store
// consecutive dwords and the address path merges them — about one request per (tile, bucket) run instead of one per entry.  The global
// position of a run is the slice's cursor of that bucket (from the scan over (bucket, slice) counts, as before), advanced tile by tile.
// For bins of at most 2^11 buckets and SORT_THREADS = 1024 (= SCAN_BLOCK); other plans keep the direct kernel.
#ifndef STAGE_CAP
#define STAGE_CAP 8192u
#endif
// The kernel's LDS layout (2048 bucket counters as two words per thread: loc[tid], loc[tid + SCAN_BLOCK]; a 16-entry area of
// wave totals; PER = STAGE_CAP / SCAN_BLOCK entries per thread; positions in a tile as u16) is written for exactly these values:
static_assert(SCAN_BLOCK == 1024u && STAGE_CAP % SCAN_BLOCK == 0 && STAGE_CAP <= 65536u, "k_bin_scatter_staged: retune its LDS layout with SCAN_BLOCK / STAGE_CAP");
__global__ __launch_bounds__(SCAN_BLOCK) void k_bin_scatter_staged(uint32_t *entries, const uint32_t *starts, const uint16_t *lo, const uint32_t *val,
                                                                   const uint32_t *bin_starts, uint32_t nblocks, uint32_t buckets_per_bin, uint32_t nbins,
                                                                   uint32_t slices, uint32_t total_buckets) {
    ZK_CHAIN_PRIO();
    extern __shared__ uint32_t sm[];
    uint32_t b, slice;
    if (!bin_slice_of_block(nbins, slices, b, slice)) return;
    const uint32_t first = b * buckets_per_bin, nb = total_buckets - first < buckets_per_bin ? total_buckets - first : buckets_per_bin;
    uint32_t *loc = sm;                               // [2048] rank counters of the tile, then its exclusive bucket offsets
    uint32_t *gcur = sm + 2048;                       // [2048] this slice's global cursor per bucket
    uint32_t *scan_lds = sm + 4096;                   // [16]
    uint32_t *sval = sm + 4096 + 16;                  // [STAGE_CAP]
    uint16_t *slo = (uint16_t *)(sval + STAGE_CAP);   // [STAGE_CAP]
    const uint32_t tid = threadIdx.x;
    const uint32_t *in = starts + (uint64_t)first * slices + slice;
    for (uint32_t k = tid; k < 2048u; k += SCAN_BLOCK) gcur[k] = k < nb ? in[(uint64_t)k * slices] : 0u;
    const uint64_t bs = bin_starts[(uint64_t)b * nblocks], be = bin_starts[(uint64_t)(b + 1) * nblocks];
    const uint64_t len = be - bs, s0 = bs + len * slice / slices, s1 = bs + len * (slice + 1) / slices;
    constexpr uint32_t PER = STAGE_CAP / SCAN_BLOCK;  // entries per thread and tile
    for (uint64_t t0 = s0; t0 < s1; t0 += STAGE_CAP) {
        const uint32_t cnt = (uint32_t)(s1 - t0 < STAGE_CAP ? s1 - t0 : STAGE_CAP);
        loc[tid] = 0;
        loc[tid + SCAN_BLOCK] = 0;
        __syncthreads();                              // (also: the previous tile's write-out has read loc / sval)
        uint32_t key[PER], v[PER], rank[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t p = k * SCAN_BLOCK + tid;
            key[k] = p < cnt ? lo[t0 + p] : 0xFFFFFFFFu;
            v[k] = p < cnt ? val[t0 + p] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) rank[k] = key[k] != 0xFFFFFFFFu ? atomicAdd(&loc[key[k]], 1u) : 0u;
        __syncthreads();
        // exclusive scan of the 2048 counts: two per thread
        const uint32_t c0 = loc[2 * tid], c1 = loc[2 * tid + 1];
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(c0 + c1, scan_lds, total);
        __syncthreads();
        loc[2 * tid] = ex;
        loc[2 * tid + 1] = ex + c0;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            if (key[k] != 0xFFFFFFFFu) {
                const uint32_t pos = loc[key[k]] + rank[k];
                sval[pos] = v[k];
                slo[pos] = (uint16_t)key[k];
            }
        }
        __syncthreads();
        for (uint32_t p = tid; p < cnt; p += SCAN_BLOCK) {
            const uint32_t kk = slo[p];
            entries[gcur[kk] + (p - loc[kk])] = sval[p];
        }
        __syncthreads();
        // advance the cursors by the tile's counts: count of bucket k = loc[k + 1] - loc[k] (cnt - loc[k] for the last one)
        const uint32_t a0 = loc[2 * tid], a1 = loc[2 * tid + 1], a2 = 2 * tid + 2 < 2048u ? loc[2 * tid + 2] : cnt;
        gcur[2 * tid] += a1 - a0;
        gcur[2 * tid + 1] += a2 - a1;
        __syncthreads();                              // the next tile zeroes loc in another thread-to-word pattern
    }
}

// ---------------------------------------------------------------- load-balanced accumulation
// A lane-per-bucket walk is hopeless on real data: the top window of uniformly random
// scalars has only a handful of non-empty buckets (r ~ 2^253.6), and real witnesses pile
// half their entries into bucket "1" of window 0.  Instead EVERY lane adds exactly
// ACC_CHUNK consecutive entries of the bucket-sorted list, whatever buckets they span:
//   * a bucket run that starts and ends inside the chunk is complete -> buckets[b];
//   * a run cut by the chunk's left edge goes to the lane's HEAD slot, one cut by the right
//     edge to its TAIL slot (at most one of each), tagged with its bucket and STARTS/ENDS flags;
//   * the slot list (2 per lane, still bucket-sorted) is reduced by wave-parallel segmented scans over
//     XYZZ partials (k_msm_accum_wave), shrinking 32x (G2: 16x) per level until one wave is left.
// Work per lane is constant, so the kernel time is flat in the scalar distribution.  Because it is constant,
// the grid runs in lock-step "rounds" of as many workgroups as fit on the chip at once (G1: 3 waves/SIMD =
// 768 workgroups, G2: 2 waves/SIMD = 512), and a last round that is only partly full costs a whole round:
// at 2^22 a fixed chunk of 128 gave 1664 workgroups = 2.17 rounds, i.e. the kernel ran at 72% of its own
// rate.  So the host launches a WHOLE number of rounds of lanes (accum_lanes_for) and the chunk is whatever
// divides the entries evenly among them (32..160 entries; below 32 fewer lanes are launched instead).  That is a LONE proof's
// plan; one submitted beside others takes 128..1280 entries per lane (AccumTail::chunk_min / chunk_max, set in prover_pipeline.hip):
// the chip is shared anyway, and fewer lanes leave fewer cut runs to merge.
#define ACC_CHUNK_MAX 160u  // affine points per lane, level 1: more than this and another round of lanes is launched
#define ACC_CHUNK_MIN 32u   // fewer than this and fewer lanes are launched (small or sharded MSMs)
// Every lane takes the same share of the E entries actually present (E <= max_entries is known on the device
// only): the host fixes the number of lanes, the chunk follows.
__device__ __forceinline__ uint32_t accum_chunk_dev(uint32_t E, uint32_t nlanes, uint32_t chunk_min) {
    const uint32_t c = (uint32_t)(((uint64_t)E + nlanes - 1) / nlanes);
    return c < chunk_min ? chunk_min : c;
}
#ifdef ZK_PROBES
#define ZK_GATHER_ROW(i) ((i) & batch.gather_mask)
#else
#define ZK_GATHER_ROW(i) (i)
#endif
#define ACC_CHUNK_N 32u     // slots per unit at levels >= 2 used to SIZE the workspace (a G2 wave takes 32, a G1 wave 64)
// Threads per workgroup of the level-1 kernels.  They use no LDS and no barrier, so the workgroup is only the unit in which the
// dispatcher hands waves to a CU (ZK_L1_BLOCK=64 in a measurement build: one wave per workgroup).
#ifndef ZK_L1_BLOCK
#define ZK_L1_BLOCK 256
#endif
#ifndef ZK_L1_PREFETCH
#define ZK_L1_PREFETCH 1      // raw points in flight per lane in the G1 level-1 kernel
#endif
#define SLOT_EMPTY 0xffffffffu
#define FLAG_STARTS 1u
#define FLAG_ENDS 2u

// Occupancy is what the compiler picks: G1 132 VGPRs (3 waves/SIMD); the G2 kernel (Fq2 split over lane pairs, below) 212 VGPRs
// (2 waves/SIMD).  Forcing 4 G1 waves (128 VGPRs, 20 B scratch) was measured slower for the whole proof.
// blockIdx.y selects one of up to three MSMs over the SAME sorted entry list (A, B1, C share sort(w)):
// small circuits launch them together — a level-1 launch there is latency-bound (a lane's chain of 32
// adds, a grid that does not fill the chip) and three of them cost what one costs.
#ifdef ZK_G1_FOUR_WAVES
#define ZK_G1_L1_WAVES __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define ZK_G1_L1_WAVES
#endif
template <class F>
__global__ __launch_bounds__(ZK_L1_BLOCK) ZK_G1_L1_WAVES void k_msm_accum_l1(G1Acc *buckets0, const uint32_t *offsets, const uint32_t *entries,
                                                      AccumBatch batch, uint32_t nbuckets_total, G1Acc *out_part0, uint32_t *out_key0,
                                                      uint32_t *out_flag0, uint32_t nlanes, uint32_t chunk_min) {
    static_assert(sizeof(F) == sizeof(Fq), "G1 only: the G2 level-1 kernel is k_msm_accum_l1_g2s");
    const uint32_t m = blockIdx.y;
    G1Acc *buckets = buckets0 + (uint64_t)m * batch.bucket_stride;
    G1Acc *out_part = out_part0 + (uint64_t)m * batch.ws_stride;
    uint32_t *out_key = out_key0 + (uint64_t)m * batch.ws_stride, *out_flag = out_flag0 + (uint64_t)m * batch.ws_stride;
    const Affine<F> *points = reinterpret_cast<const Affine<F> *>(batch.points[m]);
    const uint32_t idx_min = batch.idx_min[m], idx_sub = batch.idx_sub[m];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlanes) return;
    const uint32_t E = offsets[nbuckets_total];
    const uint32_t ACC_CHUNK = accum_chunk_dev(E, nlanes, chunk_min);
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
        // The end of the bucket AFTER the current one, re-loaded at the top of every iteration (one cached dword next to
        // the prefetch): a dependent load at the run's end — a run ends in ~46 % of a wave's iterations — cost a memory
        // latency every time, and so does a load issued AT the run's end (the loop-top wait for the next entry is
        // in-order: it would wait for that load too).
        uint32_t bend2 = 0;
        bool started_before = offsets[b] < lo;
        typedef REGF FR;
        XYZZ<FR> acc = XYZZ<FR>::inf();
        Affine<F> nextP;                 // raw words: the next point is in flight while this one is added
        bool nextNeg = false, nextSkip = false;
        // two loads deep: the ENTRY of position e+2 is in flight while the POINT of e+1 is, so the address of a point
        // load never waits for its entry (a wave's three resident siblings run in phase with it — same work, same
        // start — and do not cover that wait)
        // The loads are UNCONDITIONAL (positions past the chunk's end are clamped to its last entry and the result is
        // never used): a load under `if (e < hi)` makes the compiler merge its result with the old value right behind
        // the branch — an s_waitcnt vmcnt(0) straight after the issue, i.e. no prefetch at all.
        uint32_t entNext = entries[lo];
        auto fetch = [&](uint32_t pos) {          // point of position pos (its entry is in entNext), entry of pos + 1
            const uint32_t ent = entNext;
            const uint32_t idx = ent & 0x7fffffffu;
            nextNeg = (ent >> 31) != 0;
            nextSkip = idx < idx_min;
            bend2 = offsets[b + 2 < nbuckets_total ? b + 2 : nbuckets_total];
            const Affine<F> *src = points + (nextSkip ? 0 : ZK_GATHER_ROW(idx - idx_sub));
            nextP.x = load_row_el(&src->x);
            nextP.y = load_row_el(&src->y);
            entNext = entries[pos + 1 < hi ? pos + 1 : hi - 1];
        };
        uint32_t e = lo;
        fetch(e);
#if ZK_L1_PREFETCH >= 2
        // TWO points in flight: the table rows are random 64-byte gathers from a 3 GiB table, and with the rows confined to an
        // L2-resident range (probe: ZKHIP_GATHER_MASK) the launch is 10 % shorter — more than the 8 % the s_waitcnt counters
        // showed: one addition (~4.5 us) is not always enough to cover a gather under this load.  The kernel has 36 VGPRs to
        // spare below the three-waves-per-SIMD limit; the second raw point costs 16 of them and 18 moves per addition.
        Affine<F> curP = nextP;
        bool curNeg = nextNeg, curSkip = nextSkip;
        fetch(lo + 1 < hi ? lo + 1 : hi - 1);
#endif
        while (e < hi) {
#if ZK_L1_PREFETCH >= 2
            Affine<F> Pw = curP;
            bool ng = curNeg, skip = curSkip;
            curP = nextP; curNeg = nextNeg; curSkip = nextSkip;      // the point of position e + 1 (still on its way, normally)
            e++;
            fetch(e + 1 < hi ? e + 1 : hi - 1);                        // point of e + 1 (entry prefetched), entry of e + 2
#else
            Affine<F> Pw = nextP;
            bool ng = nextNeg, skip = nextSkip;
            e++;
            fetch(e < hi ? e : hi - 1);
#endif
            if (!skip) {
                Affine<FR> P = to_reg_affine<F>(Pw);
                if (ng) negate_y(P);
                madd(acc, P);          // curve29.hpp: bound-tracked specialisation
            }
            if (e == bend || e == hi) {              // the run of bucket b ends here (or is cut)
                const bool ends = (e == bend);
                // ONE store sequence whatever the case (a wave's lanes are in different ones): complete run -> its bucket,
                // run cut on the left -> HEAD slot, cut on the right only -> TAIL slot
                G1Acc *dst = (!started_before && ends) ? buckets + b : out_part + 2 * (uint64_t)t + (started_before ? 0u : 1u);
                if (started_before) {
                    hkey = b;
                    hflag = ends ? FLAG_ENDS : 0u;
                } else if (!ends) {
                    tkey = b;
                    tflag = FLAG_STARTS;
                }
                LaneModel<Fq>::store(dst, acc);
                if (e < hi) {                        // next non-empty bucket
                    b++;
                    bend = bend2;
                    while (bend == e) { b++; bend = offsets[b + 1]; }        // empty buckets: rare on uniform scalars
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

// G2 level 1 with the accumulator split across lane pairs (curve29.hpp, Fq2s): two lanes per chunk,
// lane parity = Fq2 component.  Both lanes of a pair walk the same entries, so the loop and every
// branch are uniform inside the pair (the DPP exchanges need both lanes active).
#ifdef ZK_G2_THREE_WAVES
#define ZK_G2_L1_WAVES __attribute__((amdgpu_waves_per_eu(3, 3)))
#else
#define ZK_G2_L1_WAVES
#endif
__global__ __launch_bounds__(ZK_L1_BLOCK) ZK_G2_L1_WAVES void k_msm_accum_l1_g2s(G2Acc *buckets, const uint32_t *offsets, const uint32_t *entries,
                                                          const G2Affine *points, uint32_t idx_min, uint32_t idx_sub,
                                                          uint32_t nbuckets_total, G2Acc *out_part, uint32_t *out_key,
                                                          uint32_t *out_flag, uint32_t nlanes, uint32_t chunk_min) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gt >> 1, comp = gt & 1u;          // chunk, component (blockDim is even: comp == threadIdx.x & 1)
    if (t >= nlanes) return;
    const uint32_t E = offsets[nbuckets_total];
    const uint32_t ACC_CHUNK = accum_chunk_dev(E, nlanes, chunk_min);
    const uint64_t lo64 = (uint64_t)t * ACC_CHUNK;
    uint32_t hkey = SLOT_EMPTY, tkey = SLOT_EMPTY, hflag = 0, tflag = 0;
    auto store_comp = [&](G2Acc *dst, const XYZZ<Fq2s> &v) { LaneModel<Fq2>::store(dst, v); };   // this lane's component (comp == threadIdx.x & 1)
    if (lo64 < E) {
        const uint32_t lo = (uint32_t)lo64;
        const uint32_t hi = (E - lo > ACC_CHUNK) ? lo + ACC_CHUNK : E;
        uint32_t bl = 0, br = nbuckets_total - 1;
        while (bl < br) {
            uint32_t mid = (bl + br + 1) >> 1;
            if (offsets[mid] <= lo) bl = mid; else br = mid - 1;
        }
        uint32_t b = bl;
        uint32_t bend = offsets[b + 1];
        uint32_t bend2 = 0;              // end of the bucket after the current one, as in the G1 kernel
        bool started_before = offsets[b] < lo;
        XYZZ<Fq2s> acc = XYZZ<Fq2s>::inf();
        Fq nextX, nextY;                 // raw words of this lane's component of the next point
        bool nextNeg = false, nextSkip = false;
        uint32_t entNext = entries[lo];               // two loads deep and unconditional, as in the G1 kernel
        auto fetch = [&](uint32_t pos) {
            const uint32_t ent = entNext;
            const uint32_t idx = ent & 0x7fffffffu;
            nextNeg = (ent >> 31) != 0;
            nextSkip = idx < idx_min;
            bend2 = offsets[b + 2 < nbuckets_total ? b + 2 : nbuckets_total];
            const Fq *src = reinterpret_cast<const Fq *>(points + (nextSkip ? 0 : idx - idx_sub)) + comp;
            nextX = load_row_el(src);
            nextY = load_row_el(src + 2);
            entNext = entries[pos + 1 < hi ? pos + 1 : hi - 1];
        };
        uint32_t e = lo;
        fetch(e);
        while (e < hi) {
            Fq Xw = nextX, Yw = nextY;
            bool ng = nextNeg, skip = nextSkip;
            e++;
            fetch(e < hi ? e : hi - 1);
            if (!skip) {
                Affine<Fq2s> P{Fq2s{Fq29::load(Xw)}, Fq2s{Fq29::load(Yw)}};
                if (ng) P.y.v = Fq29::neg_lazy(P.y.v);
                madd(acc, P);
            }
            if (e == bend || e == hi) {
                const bool ends = (e == bend);
                G2Acc *dst = (!started_before && ends) ? buckets + b : out_part + 2 * (uint64_t)t + (started_before ? 0u : 1u);
                if (started_before) {
                    hkey = b;
                    hflag = ends ? FLAG_ENDS : 0u;
                } else if (!ends) {
                    tkey = b;
                    tflag = FLAG_STARTS;
                }
                store_comp(dst, acc);
                if (e < hi) {
                    b++;
                    bend = bend2;
                    while (bend == e) { b++; bend = offsets[b + 1]; }
                    started_before = false;
                    acc = XYZZ<Fq2s>::inf();
                }
            }
        }
    }
    if (comp == 0) {
        out_key[2 * (uint64_t)t] = hkey;
        out_flag[2 * (uint64_t)t] = hflag;
        out_key[2 * (uint64_t)t + 1] = tkey;
        out_flag[2 * (uint64_t)t + 1] = tflag;
    }
}

// Pairwise merge between level 1 and the generic levels: a bucket cut by exactly ONE chunk edge
// (the overwhelmingly common case — mean run length ~ chunk length) is the TAIL slot of lane t
// plus the HEAD slot of lane t+1 that also ENDS.  One general add per lane, full occupancy,
// and both slots are retired; what is left for the serial-ish generic levels is only the
// buckets spanning three or more chunks (top window, skewed witnesses).
template <class F>
__global__ __launch_bounds__(256) void k_msm_accum_pair(ACCMEM *buckets, const ACCMEM *part, uint32_t *key, const uint32_t *flag,
                                                        uint32_t nlanes, uint64_t bucket_stride, uint64_t ws_stride) {
    ZK_TAIL_PRIO();
    typedef LaneModel<F> LM;
    buckets += (uint64_t)blockIdx.y * bucket_stride;      // blockIdx.y: MSM of a batch (see k_msm_accum_l1)
    part += (uint64_t)blockIdx.y * ws_stride;
    key += (uint64_t)blockIdx.y * ws_stride;
    flag += (uint64_t)blockIdx.y * ws_stride;
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gt / LM::LPE;
    if (t + 1 >= nlanes) return;
    const uint32_t kt = key[2 * (uint64_t)t + 1], kh = key[2 * (uint64_t)t + 2];
    if (kt == SLOT_EMPTY || kt != kh) return;
    if (!(flag[2 * (uint64_t)t + 2] & FLAG_ENDS)) return;       // continues further: generic levels
    XYZZ<typename LM::R> a = LM::load(part + 2 * (uint64_t)t + 1);
    add(a, LM::load(part + 2 * (uint64_t)t + 2));
    LM::store(buckets + kt, a);
    // (both lanes of a G2 pair have read the keys before either of them retires the slots: the pair
    // is in one wave and the loads above precede these stores in program order)
    if (gt % LM::LPE == 0) {
        key[2 * (uint64_t)t + 1] = SLOT_EMPTY;
        key[2 * (uint64_t)t + 2] = SLOT_EMPTY;
    }
}

// Levels >= 2, wave-parallel: a WAVE takes 64 consecutive slots (32 for G2: a lane pair per element),
// compacts the non-empty ones to its low lanes (ds_permute), and runs a segmented inclusive scan by
// bucket key — log2(64) steps of one general add per lane instead of up to ACC_CHUNK_N sequential adds
// in one lane.  What made the sequential version slow is exactly the data that reaches these levels: a
// bucket spread over thousands of level-1 chunks (the top window of uniform scalars has ~12 non-empty
// buckets of n/12 points each; bucket "1" of real witnesses) is a run of thousands of partials, i.e.
// serial chains of 32 general adds per level (measured 1.3-1.7 ms per level at 2^22, exposed after the
// last MSM of a proof).  A wave that finds all its slots empty (the common case after the pairwise
// merge) leaves at once; the scan stops at the first distance no lane has a partner at.  Every wave
// emits at most two partials (its first run if that does not START there, its last run if it does not
// END there), so a level shrinks the list 32x (16x for G2).
template <class R> __device__ __forceinline__ R wave_up(const R &v, uint32_t d);
template <> __device__ __forceinline__ Fq29 wave_up<Fq29>(const Fq29 &v, uint32_t d) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = __shfl_up(v.l[i], d);
    return r;
}
template <> __device__ __forceinline__ Fq2s wave_up<Fq2s>(const Fq2s &v, uint32_t d) { return Fq2s{wave_up<Fq29>(v.v, d)}; }
template <class R> __device__ __forceinline__ R wave_push(const R &v, uint32_t dst_lane);       // lane L's value -> lane dst_lane (a permutation)
template <> __device__ __forceinline__ Fq29 wave_push<Fq29>(const Fq29 &v, uint32_t dst) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = __builtin_amdgcn_ds_permute((int)(dst << 2), v.l[i]);
    return r;
}
template <> __device__ __forceinline__ Fq2s wave_push<Fq2s>(const Fq2s &v, uint32_t dst) { return Fq2s{wave_push<Fq29>(v.v, dst)}; }

template <class F>
__global__ __launch_bounds__(256) void k_msm_accum_wave(ACCMEM *buckets, const ACCMEM *in_part, const uint32_t *in_key,
                                                        const uint32_t *in_flag, uint32_t nitems, ACCMEM *out_part,
                                                        uint32_t *out_key, uint32_t *out_flag, uint32_t nwaves,
                                                        uint64_t bucket_stride, uint64_t ws_stride) {
    ZK_TAIL_PRIO();
    typedef LaneModel<F> LM;
    typedef typename LM::R FR;
    {
        const uint64_t bo = (uint64_t)blockIdx.y * bucket_stride, wo = (uint64_t)blockIdx.y * ws_stride;   // MSM of a batch
        buckets += bo;
        in_part += wo; in_key += wo; in_flag += wo;
        out_part += wo; out_key += wo; out_flag += wo;
    }
    constexpr uint32_t LPE = LM::LPE, EPW = 64u / LPE;          // lanes per element, elements per wave
    const uint32_t lane = threadIdx.x & 63u, w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= nwaves) return;                                    // wave-uniform
    const uint32_t e = lane / LPE, comp = lane % LPE;           // this lane's element slot, component
    const uint32_t i = w * EPW + e;
    uint32_t key = i < nitems ? in_key[i] : SLOT_EMPTY;
    uint32_t flag = key != SLOT_EMPTY ? in_flag[i] : 0u;
    if (lane == 0) {                                            // ordered before the partial stores below (same wave)
        out_key[2 * (uint64_t)w] = SLOT_EMPTY;
        out_key[2 * (uint64_t)w + 1] = SLOT_EMPTY;
        out_flag[2 * (uint64_t)w] = 0;
        out_flag[2 * (uint64_t)w + 1] = 0;
    }
    const uint64_t vmask = __ballot(key != SLOT_EMPTY);
    if (vmask == 0) return;
    XYZZ<FR> v = XYZZ<FR>::inf();
    if (key != SLOT_EMPTY) v = LM::load(in_part + i);
    // compaction: valid element with `rank` valid elements before it -> element slot rank; the invalid ones fill the rest
    const uint64_t below = vmask & ((1ull << (e * LPE)) - 1ull);
    const uint32_t rank = (uint32_t)__popcll(below) / LPE, cnt = (uint32_t)__popcll(vmask) / LPE;
    const uint32_t dst_e = key != SLOT_EMPTY ? rank : cnt + (e - rank);
    const uint32_t dst = dst_e * LPE + comp;
    key = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)key);
    flag = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)flag);
    v.x = wave_push(v.x, dst); v.y = wave_push(v.y, dst); v.zz = wave_push(v.zz, dst); v.zzz = wave_push(v.zzz, dst);
    // segmented inclusive scan over the equal-key runs of elements 0 .. cnt-1
    for (uint32_t d = 1; d < EPW; d <<= 1) {
        const uint32_t kprev = __shfl_up(key, d * LPE);
        const bool take = e >= d && e < cnt && kprev == key;
        if (!__any(take)) break;                                // runs are contiguous: nothing at 2d either
        XYZZ<FR> o{wave_up(v.x, d * LPE), wave_up(v.y, d * LPE), wave_up(v.zz, d * LPE), wave_up(v.zzz, d * LPE)};
        if (take) add(v, o);
    }
    if (e >= cnt) return;
    const uint32_t knext = __shfl_down(key, LPE);
    const uint32_t kbefore = __shfl_up(key, LPE);
    const bool last = e + 1 == cnt || knext != key;
    const bool first = e == 0 || kbefore != key;
    // STARTS comes from the first element of the run, ENDS from the last one (this lane when `last`)
    const uint64_t fmask = __ballot(first);
    const uint32_t start_lane = 63u - (uint32_t)__clzll(fmask & ((2ull << lane) - 1ull));     // highest run start at or below this lane (same comp parity not needed: flags are replicated)
    const uint32_t sflag = (uint32_t)__shfl((int)flag, (int)start_lane);
    if (!last) return;
    const bool starts = (sflag & FLAG_STARTS) != 0, ends = (flag & FLAG_ENDS) != 0;
    if (starts && ends) {
        LM::store(buckets + key, v);
    } else if (!starts) {
        LM::store(out_part + 2 * (uint64_t)w, v);
        if (comp == 0) {
            out_key[2 * (uint64_t)w] = key;
            out_flag[2 * (uint64_t)w] = ends ? FLAG_ENDS : 0u;
        }
    } else {
        LM::store(out_part + 2 * (uint64_t)w + 1, v);
        if (comp == 0) {
            out_key[2 * (uint64_t)w + 1] = key;
            out_flag[2 * (uint64_t)w + 1] = FLAG_STARTS;
        }
    }
}

// Large bucket sets, first level: lane t of a set takes the buckets t*chunk .. t*chunk + chunk - 1, so that
// (k+1) = t*chunk + (j+1) splits  sum_k (k+1) B_k  =  chunk * sum_t t*T_t  +  sum_t A_t   with T_t = sum_j B[t*chunk+j] and
// A_t = sum_j (j+1)*B[t*chunk+j]: two additions per bucket and NO per-lane scalar multiplication (the chunked form below
// pays ~28 point operations per lane for lo*T on top of its 32).  sum_t t*T_t is the same problem on the L - 1 points
// T_1 .. T_(L-1) (stored shifted by one; the chunked form finishes it), the A_t are a plain tree sum.
template <class F>
__global__ __launch_bounds__(128) void k_msm_reduce_split(ACCMEM *A, uint32_t a_stride, ACCMEM *T, const ACCMEM *buckets, uint32_t nbuckets, uint32_t chunk, uint32_t L) {
    ZK_TAIL_PRIO();
    typedef LaneModel<F> LM;
    typedef typename LM::R FR;
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / LM::LPE;
    if (t >= L) return;
    const ACCMEM *B = buckets + (uint64_t)blockIdx.y * nbuckets + (uint64_t)t * chunk;
    XYZZ<FR> run = XYZZ<FR>::inf(), sum = XYZZ<FR>::inf();
    for (int j = (int)chunk - 1; j >= 0; j--) {
        add(run, LM::load(B + j));
        add(sum, run);
    }
    LM::store(A + (uint64_t)blockIdx.y * a_stride + t, sum);
    if (t == 0) run = XYZZ<FR>::inf();                                     // weight 0; its slot is the (empty) last one
    LM::store(T + (uint64_t)blockIdx.y * L + (t ? t - 1 : L - 1), run);
}

// Lane per chunk of REDUCE_CHUNK buckets: running sums give A = sum (j+1)*B[lo+j], T = sum B;
// X = A + lo*T is the chunk's share of sum_k (k+1)*B_k.
template <class F>
__global__ __launch_bounds__(128) void k_msm_reduce_chunks(ACCMEM *scratch, uint32_t out_stride, const ACCMEM *buckets, uint32_t nbuckets,
                                                           uint32_t chunk, uint32_t total_chunks) {
    ZK_TAIL_PRIO();
    typedef LaneModel<F> LM;
    typedef typename LM::R FR;
    uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / LM::LPE;
    if (t >= total_chunks) return;
    const uint32_t chunks_per_window = nbuckets / chunk;
    const uint32_t cw = t % chunks_per_window;          // chunk index inside its window
    const ACCMEM *B = buckets + (uint64_t)t * chunk;    // windows (and MSMs) are laid back to back
    XYZZ<FR> run = XYZZ<FR>::inf(), sum = XYZZ<FR>::inf();
    for (int j = (int)chunk - 1; j >= 0; j--) {
        add(run, LM::load(B + j));
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
    LM::store(scratch + (uint64_t)(t / chunks_per_window) * out_stride + cw, sum);      // (out_stride = chunks per set: back to back)
}

// Tree sum of `count` consecutive points per group, 2 inputs per lane + an LDS tree per workgroup:
// grid (blocks_per_group, groups) -> one point per workgroup.  Launched repeatedly until one point
// per (msm, window) is left; the last launch stores in the zkey's 2^256 Montgomery form.
template <class F>
static constexpr uint32_t tree_in() { return 2u * REDUCE_THREADS / LaneModel<F>::LPE; }    // inputs per workgroup
#define TREE_IN_MIN REDUCE_THREADS      // the smaller fan-in (G2): sizes the shared scratch formula
template <class F>
__global__ __launch_bounds__(REDUCE_THREADS) void k_msm_reduce_tree(ACCMEM *out, XYZZ<F> *out_final, const ACCMEM *in, uint32_t count, uint32_t last) {
    ZK_TAIL_PRIO();
    extern __shared__ uint32_t lds_raw[];
    typedef LaneModel<F> LM;
    typedef typename LM::R FR;
    constexpr uint32_t NE = REDUCE_THREADS / LM::LPE;           // elements per workgroup pass
    XYZZ<FR> *lds = reinterpret_cast<XYZZ<FR> *>(lds_raw);      // one entry per lane (its component(s))
    const ACCMEM *X = in + (uint64_t)blockIdx.y * count;
    const uint32_t e = threadIdx.x / LM::LPE;
    const uint32_t i0 = blockIdx.x * (2u * NE) + e, i1 = i0 + NE;
    XYZZ<FR> acc = XYZZ<FR>::inf();
    if (i0 < count) acc = LM::load(X + i0);
    if (i1 < count) add(acc, LM::load(X + i1));
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = NE / 2; s > 0; s >>= 1) {
        if (e < s) {
            XYZZ<FR> o = lds[threadIdx.x + s * LM::LPE];
            add(acc, o);
            lds[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (e == 0) {
        const uint64_t at = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
        if (last) LM::store256(out_final + at, acc);    // window sums: canonical words, back in the zkey's 2^256 form
        else LM::store(out + at, acc);
    }
}

// Last launch of the split form: per set, the partial sums of the shares X (cnt_x points at P) and of the A level (cnt_a
// points behind them) are summed in the two halves of ONE LDS tree;  window sum = 2^scale_log * sum X + sum A.
template <class F>
__global__ __launch_bounds__(REDUCE_THREADS) void k_msm_reduce_final2(XYZZ<F> *out_final, const ACCMEM *P, uint32_t cnt_x, uint32_t cnt_a, uint32_t scale_log) {
    ZK_TAIL_PRIO();
    extern __shared__ uint32_t lds_raw[];
    typedef LaneModel<F> LM;
    typedef typename LM::R FR;
    constexpr uint32_t H = REDUCE_THREADS / LM::LPE / 2;        // elements per half: 2H inputs each
    XYZZ<FR> *lds = reinterpret_cast<XYZZ<FR> *>(lds_raw);
    const uint32_t e = threadIdx.x / LM::LPE;
    const bool xs = e >= H;                                      // upper half: the shares X;  lower half: A
    const uint32_t eh = xs ? e - H : e, cnt = xs ? cnt_x : cnt_a;
    const ACCMEM *in = P + (uint64_t)blockIdx.x * (cnt_x + cnt_a) + (xs ? 0u : cnt_x);
    XYZZ<FR> acc = XYZZ<FR>::inf();
    if (eh < cnt) acc = LM::load(in + eh);
    if (eh + H < cnt) add(acc, LM::load(in + eh + H));
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = H / 2; s > 0; s >>= 1) {
        if (eh < s) {
            XYZZ<FR> o = lds[threadIdx.x + s * LM::LPE];
            add(acc, o);
            lds[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (xs && eh == 0) {
        for (uint32_t i = 0; i < scale_log; i++) acc = dbl(acc);
        lds[threadIdx.x] = acc;
    }
    __syncthreads();
    if (e == 0) {
        XYZZ<FR> o = lds[threadIdx.x + H * LM::LPE];
        add(acc, o);
        LM::store256(out_final + blockIdx.x, acc);
    }
}

// ---- bit-sum reduction: small bucket sets ---------------------------------------------------------------
// sum_k (k+1) B_k = T + sum_j 2^j S_j with T = sum_k B_k and S_j = sum of the buckets whose index has bit j
// set.  The c sums (T, S_0 .. S_{c-2}) come out of ONE binary tree: a block of 2^m buckets carries m+1 sums;
// joining siblings L (bit m clear) and R (bit m set) is  T = T_L + T_R,  S_j = S_j^L + S_j^R (j < m),
// S_m = T_R  — m+1 INDEPENDENT additions per join, so every level is one addition deep and the whole
// reduction is c-1 additions deep (≈ 0.1 ms), where the chunked form above is a serial chain of 2*chunk
// additions plus a ~c-step double-and-add per lane and then an LDS tree (0.35-0.5 ms per launch pair, a third
// of a small proof's kernel time).  The c sums go to the host, whose serial Horner over c-1 bits costs
// microseconds.  Used while a launch reduces at most 2^16 buckets (circuits up to 2^18 constraints); the chunked
// form stays for the large sets, where work, not depth, is what counts.  In place: a block of size 2^m keeps T in its slot 0 and S_j in slot 1+j.
#define BITS_RS 18u         // slots per block record in global memory (>= c)
template <class F>
__global__ __launch_bounds__(REDUCE_THREADS) void k_msm_reduce_bits_block(ACCMEM *rec, XYZZ<F> *final_out, const ACCMEM *buckets,
                                                                         uint32_t nbuckets, uint32_t c, uint32_t nblk) {
    ZK_TAIL_PRIO();
    extern __shared__ uint32_t lds_raw[];
    typedef LaneModel<F> LM;
    typedef typename LM::R FR;
    constexpr uint32_t NE = REDUCE_THREADS / LM::LPE;           // buckets per workgroup: 256 (G1), 128 (G2)
    constexpr uint32_t LB = LM::LPE == 1 ? 8u : 7u;
    XYZZ<FR> *lds = reinterpret_cast<XYZZ<FR> *>(lds_raw);      // slot s of the block: lds[s * LPE + component]
    const uint32_t e = threadIdx.x / LM::LPE, comp = threadIdx.x % LM::LPE;
    const uint32_t group = blockIdx.y, blk = blockIdx.x;
    const uint32_t k = blk * NE + e;
    lds[threadIdx.x] = k < nbuckets ? LM::load(buckets + (uint64_t)group * nbuckets + k) : XYZZ<FR>::inf();
    __syncthreads();
#pragma unroll 1
    for (uint32_t m = 0; m < LB; m++) {
        const uint32_t per = m + 2, ng = NE >> (m + 1);
        const uint32_t g = e / per, i = e % per;
        if (g < ng) {
            const uint32_t L = g << (m + 1), R = L + (1u << m);
            if (i <= m) {                                   // i = 0: T; i = 1 + j: S_j
                XYZZ<FR> a = lds[(L + i) * LM::LPE + comp];
                add(a, lds[(R + i) * LM::LPE + comp]);
                lds[(L + i) * LM::LPE + comp] = a;
            } else if (m >= 2) {                            // S_m = T_R (for m < 2 that slot IS R's slot 0)
                lds[(L + m + 1) * LM::LPE + comp] = lds[R * LM::LPE + comp];
            }
        }
        __syncthreads();
    }
    if (e <= LB && e < c) {
        const XYZZ<FR> v = lds[threadIdx.x];
        if (nblk == 1) LM::store256(final_out + (uint64_t)group * c + e, v);
        else LM::store(rec + ((uint64_t)group * nblk + blk) * BITS_RS + e, v);
    }
}
// the levels above the blocks, one workgroup per bucket set, on the block records in global memory
template <class F>
__global__ __launch_bounds__(REDUCE_THREADS) void k_msm_reduce_bits_top(XYZZ<F> *final_out, ACCMEM *rec, uint32_t nblk, uint32_t c) {
    ZK_TAIL_PRIO();
    typedef LaneModel<F> LM;
    typedef typename LM::R FR;
    constexpr uint32_t NE = REDUCE_THREADS / LM::LPE;
    constexpr uint32_t LB = LM::LPE == 1 ? 8u : 7u;
    const uint32_t e = threadIdx.x / LM::LPE;
    ACCMEM *R0 = rec + (uint64_t)blockIdx.x * nblk * BITS_RS;
#pragma unroll 1
    for (uint32_t m = LB; m + 1 < c; m++) {
        const uint32_t span = 1u << (m - LB), per = m + 2, ntasks = (nblk >> (m - LB + 1)) * per;
#pragma unroll 1
        for (uint32_t t = e; t < ntasks; t += NE) {
            const uint32_t g = t / per, i = t % per;
            ACCMEM *L = R0 + (uint64_t)g * 2u * span * BITS_RS, *R = L + (uint64_t)span * BITS_RS;
            if (i <= m) {
                XYZZ<FR> a = LM::load(L + i);
                add(a, LM::load(R + i));
                LM::store(L + i, a);
            } else {
                LM::store(L + m + 1, LM::load(R));
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (e < c) LM::store256(final_out + (uint64_t)blockIdx.x * c + e, LM::load(R0 + e));
}
static inline bool reduce_bits_for(MsmPlan p) {
    static const int forced = [] { const char *e = probe_env("ZKHIP_REDUCE_BITS"); return e ? atoi(e) : -1; }();
    if (p.c > BITS_RS || (uint64_t)p.sets * p.nbuckets > (1u << 16)) return false;      // large sets: work, not depth, counts (2^22 with plain tables: +0.6 %)
    return forced < 0 ? true : forced != 0;
}
uint32_t msm_wsum_rc(MsmPlan p) { return reduce_bits_for(p) ? p.c : 1u; }

// Buckets per lane of k_msm_reduce_chunks.  16 is the cheapest in total work (2 adds per bucket + one ~20-op
// scalar multiplication per chunk); but a lane's chain is serial (32 + ~20 general adds of ~7 us each), and
// with few buckets (small circuits, shards) the launch is a handful of waves whose latency — not work — is
// what the proof waits for (2^16: 321 us per launch, the largest single item of a proof): smaller chunks
// there (ZKHIP_REDUCE_CHUNK overrides, tuning aid).
static inline uint32_t reduce_chunk_for(MsmPlan p) {
    static const uint32_t forced = [] { const char *e = probe_env("ZKHIP_REDUCE_CHUNK"); return e ? (uint32_t)atoi(e) : 0u; }();
    uint32_t chunk = REDUCE_CHUNK;
    if (forced) chunk = forced;
    else if ((uint64_t)p.sets * p.nbuckets <= (1u << 16)) chunk = 4;
    return p.nbuckets < chunk ? p.nbuckets : chunk;
}

// Buckets per lane of k_msm_reduce_split (0: the chunked form alone).  Sets of 2^14 buckets and more: below that a launch is
// a handful of waves and the chunked form's depth is what counts.  ZKHIP_REDUCE_SPLIT overrides (tuning aid; 0 = off).
#define REDUCE_SPLIT 16u
// chunk of the chunked form on the T level (L points per set): 8 where that leaves whole tree workgroups of shares, else 4
// (2^22 / 2^20, periods with 8 against 4: -0.6 % / -0.5 %, profiles/r05zd_split_parameters.txt)
static inline uint32_t reduce_split_top(uint32_t L) {
    static const uint32_t forced = [] { const char *e = probe_env("ZKHIP_REDUCE_SPLIT_TOP"); const uint32_t t = e ? (uint32_t)atoi(e) : 0u; return t >= 2u && (t & (t - 1)) == 0 ? t : 0u; }();
    if (forced) return forced;
    return (L / 8u) % (2u * REDUCE_THREADS) == 0 ? 8u : 4u;
}
static inline uint32_t reduce_split_for(MsmPlan p) {
    static const int forced = [] { const char *e = probe_env("ZKHIP_REDUCE_SPLIT"); return e ? atoi(e) : -1; }();
    if (reduce_bits_for(p) || p.nbuckets < (1u << 14)) return 0;
    // (sets of more than 2^19 buckets: longer lanes, so that the T level stays the 2^15 points the last two launches are sized for)
    const uint32_t ch = forced >= 0 ? (uint32_t)forced : (p.nbuckets > (REDUCE_SPLIT << 15) ? p.nbuckets >> 15 : REDUCE_SPLIT);
    if (ch < 2 || (ch & (ch - 1)) != 0 || ch > p.nbuckets) return 0;
    // the last two launches take the shares and the A level together (k_msm_reduce_final2): whole tree workgroups of
    // shares, and at most one final half of partial sums, for both fan-ins
    const uint32_t L = p.nbuckets / ch;
    return L >= 8u && (L / reduce_split_top(L)) % (2u * REDUCE_THREADS) == 0 && L / TREE_IN_MIN <= REDUCE_THREADS / 2u ? ch : 0u;
}

// points of a tree's input and of every level above it (a geometric tail; upper bound for both fan-ins)
static inline uint64_t tree_points(uint64_t groups, uint64_t cnt) {
    uint64_t total = 0;
    for (;;) {
        total += groups * cnt;
        if (cnt == 1) break;
        cnt = (cnt + TREE_IN_MIN - 1) / TREE_IN_MIN;
    }
    return total + groups;      // (a one-point input still gets its output slot)
}
// scratch: chunk sums + the intermediate levels of the tree; the split form: T, its chunk sums and tree, A and its tree
uint64_t msm_reduce_scratch_points(uint32_t n_msm, MsmPlan p) {
    if (reduce_bits_for(p)) return (uint64_t)n_msm * p.sets * (p.nbuckets / 128u + 1u) * BITS_RS;      // block records (G2 blocks are the smaller)
    const uint64_t groups = (uint64_t)n_msm * p.sets;
    if (const uint32_t ch = reduce_split_for(p)) {
        const uint64_t L = p.nbuckets / ch;
        return groups * L + tree_points(groups, L / reduce_split_top((uint32_t)L) + L);      // T; shares and A side by side, the partial sums behind them
    }
    return tree_points(groups, p.nbuckets / reduce_chunk_for(p));
}

// exclusive scan of counts[0..total) -> out[0..total], out[total] = grand total;
// out must hold total + 1 + msm_scan_extra_words(total) words (block sums live past the end)
static void launch_scan(uint32_t *out, const uint32_t *counts, uint32_t total, hipStream_t s) {
    uint32_t nblocks = (total + SCAN_ELEMS - 1) / SCAN_ELEMS;
    uint32_t *block_sums = out + total + 1;
    ZK_LAUNCH(k_scan_local, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, out, block_sums, counts, total);
    ZK_LAUNCH(k_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, block_sums, nblocks, out + total);
    ZK_LAUNCH(k_scan_add, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, out, (const uint32_t *)block_sums, total);
}
uint32_t msm_scan_extra_words(uint32_t total) { return (total + SCAN_ELEMS - 1) / SCAN_ELEMS; }
void launch_exclusive_scan_u32(uint32_t *out, const uint32_t *counts, uint32_t total, hipStream_t s) { launch_scan(out, counts, total, s); }

// geometry of the two sort levels
#define BIN_SHIFT 11u          // buckets per bin = 2^11 (see above)
#define BIN_SLICES 32u         // second-level workgroups per bin
static inline uint32_t plan_total_buckets(MsmPlan p) { return p.sets * p.nbuckets; }
static inline uint32_t plan_bin_shift(MsmPlan p) {      // at most BIN_MAX bins; the low key bits travel as 16 bits
    uint32_t sh = BIN_SHIFT;
    while (((plan_total_buckets(p) + (1u << sh) - 1) >> sh) > BIN_MAX) sh++;
    return sh;
}
static inline uint32_t plan_nbins(MsmPlan p) { uint32_t sh = plan_bin_shift(p); return (plan_total_buckets(p) + (1u << sh) - 1) >> sh; }
static inline uint32_t bin_span() { return BIN_ITEMS * SORT_THREADS; }       // items per first-level workgroup
static inline uint32_t plan_bin_blocks(uint64_t n, MsmPlan p) { return (uint32_t)(((n ? n : 1) * p.W + bin_span() - 1) / bin_span()); }

MsmSortSizes msm_sort_sizes(uint64_t n, MsmPlan p) {
    MsmSortSizes z;
    memset(&z, 0, sizeof z);
    const uint64_t tb = plan_total_buckets(p), items = (n ? n : 1) * p.W;
    const uint64_t bb = (uint64_t)plan_nbins(p) * plan_bin_blocks(n, p);
    z.counts_u32 = tb * BIN_SLICES;
    z.starts_u32 = tb * BIN_SLICES + 1 + msm_scan_extra_words((uint32_t)(tb * BIN_SLICES));
    z.offsets_u32 = tb + 1;
    z.entries_u32 = items;
    z.codes_u32 = items;
    z.lo_u16 = items;
    z.val_u32 = items;
    z.bin_counts_u32 = bb;
    z.bin_starts_u32 = bb + 1 + msm_scan_extra_words((uint32_t)bb);
    return z;
}

static inline size_t bin_scatter_lds_bytes() { return (size_t)(2 * BIN_MAX + 2 * bin_span() + bin_span() / 2 + 1) * 4; }
static void sort_lds_attr() {
    static PerDeviceOnce attr;
    if (!attr.need()) return;   // > 64 KiB of dynamic LDS needs the opt-in (160 KiB per CU on gfx950), on every device
    ZK_HIP(hipFuncSetAttribute((const void *)k_bin_count_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZK_HIP(hipFuncSetAttribute((const void *)k_bin_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZK_HIP(hipFuncSetAttribute((const void *)k_bin_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZK_HIP(hipFuncSetAttribute((const void *)k_bin_scatter_staged, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr.done();
}

// digits -> bin partition -> per-bin LDS histograms -> scan -> LDS-ranked scatter -> compact bucket offsets
void launch_msm_sort(const MsmSortBufs &b, const Fr *scalars, uint64_t n, MsmPlan p, hipStream_t s) {
    const uint32_t tb = plan_total_buckets(p);
    sort_lds_attr();
    uint64_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    const uint32_t sh = plan_bin_shift(p), nbins = plan_nbins(p), nblocks = plan_bin_blocks(n, p);
    // second-level workgroups per bin: about one staging tile of entries each (longer runs per bucket = fewer, wider stores), at most
    // BIN_SLICES (what the count / start arrays are sized for), at least 4
    uint32_t slices = BIN_SLICES;
    {
        const uint64_t per_bin = (n * p.W) / (nbins ? nbins : 1);
        while (slices > 4u && per_bin / slices < STAGE_CAP * 3u / 4u) slices >>= 1;
    }
    const uint32_t bpb = tb < (1u << sh) ? tb : (1u << sh);
    const uint64_t total = n * p.W;
    const size_t lds = (size_t)bpb * 4;
    const uint32_t grid2 = ((nbins + 7u) / 8u) * 8u * slices;
    const uint32_t set_shift = p.precomp ? 32u : p.c - 1u;
    if (n) ZK_LAUNCH(k_msm_digits, dim3((uint32_t)g), dim3(256), 0, s, b.codes, scalars, n, p);
    ZK_LAUNCH(k_bin_count, dim3(nblocks), dim3(SORT_THREADS), 0, s, b.bin_counts, (const uint32_t *)b.codes, total, nbins, nblocks, sh, bin_span());
    launch_scan(b.bin_starts, b.bin_counts, nbins * nblocks, s);
    ZK_LAUNCH(k_bin_scatter, dim3(nblocks), dim3(SORT_THREADS), bin_scatter_lds_bytes(), s, b.lo, b.val, (const uint32_t *)b.bin_starts,
                       (const uint32_t *)b.codes, total, nbins, nblocks, sh, bin_span(), n, set_shift, p.batch > 1 ? p.batch_n : 0u, p.precomp);
    ZK_LAUNCH(k_bin_count_lds, dim3(grid2), dim3(SORT_THREADS), lds, s, b.counts, (const uint16_t *)b.lo,
                       (const uint32_t *)b.bin_starts, nblocks, bpb, nbins, slices, tb);
    launch_scan(b.starts, b.counts, tb * slices, s);
    static const bool direct = probe_env("ZKHIP_SORT_DIRECT") != nullptr;      // (-DZK_PROBES builds: the unstaged second-level scatter, for A/Bs)
    if (bpb <= 2048u && SORT_THREADS == SCAN_BLOCK && !direct)
        ZK_LAUNCH(k_bin_scatter_staged, dim3(grid2), dim3(SCAN_BLOCK), (size_t)(4096 + 16 + STAGE_CAP) * 4 + (size_t)STAGE_CAP * 2, s, b.entries,
                           (const uint32_t *)b.starts, (const uint16_t *)b.lo, (const uint32_t *)b.val, (const uint32_t *)b.bin_starts, nblocks, bpb, nbins,
                           slices, tb);
    else
        ZK_LAUNCH(k_bin_scatter_lds, dim3(grid2), dim3(SORT_THREADS), lds, s, b.entries, (const uint32_t *)b.starts,
                           (const uint16_t *)b.lo, (const uint32_t *)b.val, (const uint32_t *)b.bin_starts, nblocks, bpb, nbins, slices, tb);
    ZK_LAUNCH(k_msm_compact_offsets, dim3((tb + 256) / 256), dim3(256), 0, s, b.offsets, (const uint32_t *)b.starts, tb, slices);
    ZK_LAUNCH_OK("msm sort");
}

// ---------------------------------------------------------------- window pre-computation
// T[j*n + i] = 2^(c*j) * P_i for j < W, affine, resident in HBM (x W table memory — what 288 GB
// are for).  Every window then adds into the SAME bucket set, so the bucket reduction is paid
// once and the window can grow to c = 20: 13 instead of 16 additions per point.
template <class F>
__global__ __launch_bounds__(128) void k_precomp_walk(XYZZ<F> *tmp, const Affine<F> *pts, uint64_t n, uint32_t c, uint32_t W) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef REGF FR;
    XYZZ<FR> acc = XYZZ<FR>::from_affine(load_affine(pts + i));
    for (uint32_t j = 1; j < W; j++) {
        for (uint32_t k = 0; k < c; k++) acc = dbl(acc);
        store_xyzz(tmp + (uint64_t)(j - 1) * n + i, acc);
    }
}
// XYZZ -> affine over segments of 64 points with one Fermat inversion per segment
template <class F>
__global__ __launch_bounds__(64) void k_precomp_normalize(Affine<F> *out, const XYZZ<F> *tmp, F *pref, uint64_t total) {
    const uint64_t lo = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 64;
    if (lo >= total) return;
    const uint64_t hi = lo + 64 < total ? lo + 64 : total;
    typedef REGF FR;
    FR acc = FR::one();
    for (uint64_t i = lo; i < hi; i++) {
        Reg<F>::store(pref + i, acc);
        FR t = FR::mul(Reg<F>::load(&tmp[i].zz), Reg<F>::load(&tmp[i].zzz));
        if (!t.is_zero()) acc = FR::mul(acc, t);          // infinity (zz = 0): skipped
    }
    FR inv = FR::inv(acc);
    for (uint64_t i = hi; i-- > lo;) {
        FR zz = Reg<F>::load(&tmp[i].zz), zzz = Reg<F>::load(&tmp[i].zzz);
        FR t = FR::mul(zz, zzz);
        if (t.is_zero()) {
            Reg<F>::store(&out[i].x, FR::zero());
            Reg<F>::store(&out[i].y, FR::zero());
            continue;
        }
        FR ii = FR::mul(inv, Reg<F>::load(pref + i));     // 1/(zz*zzz)
        inv = FR::mul(inv, t);
        Reg<F>::store(&out[i].x, FR::mul(Reg<F>::load(&tmp[i].x), FR::mul(ii, zzz)));   // X/ZZ
        Reg<F>::store(&out[i].y, FR::mul(Reg<F>::load(&tmp[i].y), FR::mul(ii, zz)));    // Y/ZZZ
    }
}
template <class F>
static void precomp_table(Affine<F> *table, XYZZ<F> *tmp, F *pref, uint64_t n, MsmPlan p, hipStream_t s) {
    if (!n || p.W < 2) return;
    ZK_LAUNCH(k_precomp_walk<F>, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, s, tmp, (const Affine<F> *)table, n, p.c, p.W);
    const uint64_t total = (uint64_t)(p.W - 1) * n, segs = (total + 63) / 64;
    ZK_LAUNCH(k_precomp_normalize<F>, dim3((uint32_t)((segs + 63) / 64)), dim3(64), 0, s, table + n, (const XYZZ<F> *)tmp, pref, total);
    ZK_LAUNCH_OK("window pre-computation");
}
// (a table with a row per second window is a table with a row per window of twice the width)
static MsmPlan table_plan(MsmPlan p) {
    if (p.precomp > 1) { p.W = msm_table_rows(p); p.c *= p.precomp; }
    return p;
}
void launch_msm_precomp_g1(G1Affine *table, G1XYZZ *tmp, Fq *pref, uint64_t n, MsmPlan p, hipStream_t s) { precomp_table<Fq>(table, tmp, pref, n, table_plan(p), s); }
void launch_msm_precomp_g2(G2Affine *table, G2XYZZ *tmp, Fq2 *pref, uint64_t n, MsmPlan p, hipStream_t s) { precomp_table<Fq2>(table, tmp, pref, n, table_plan(p), s); }

// workspace: level-1 slots (2 per lane) + level-2 slots + ... (geometric: < 2.2x level 1)
static inline uint32_t accum_chunk_min() {
    static const uint32_t cmin = [] { const char *e = probe_env("ZKHIP_ACC_CHUNK_MIN"); uint32_t v = e ? (uint32_t)atoi(e) : ACC_CHUNK_MIN; return v < 4u ? 4u : v; }();
    return cmin;
}
static inline uint32_t accum_chunk_max() {
    static const uint32_t cmax = [] { const char *e = probe_env("ZKHIP_ACC_CHUNK_MAX"); uint32_t v = e ? (uint32_t)atoi(e) : ACC_CHUNK_MAX; return v < 8u ? 8u : v; }();
    return cmax;
}
// chunks one round holds: what the occupancy calculator says fits on the device at once (per MSM of a batch)
template <class F>
static uint64_t accum_round_lanes(uint32_t n_msm) {
    static const uint64_t threads = [] {
        int dev = 0, cus = 256, wgs = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess) {
            if constexpr (sizeof(F) == sizeof(Fq2)) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs, k_msm_accum_l1_g2s, ZK_L1_BLOCK, 0);
            else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs, k_msm_accum_l1<F>, ZK_L1_BLOCK, 0);
        }
        if (e != hipSuccess || wgs < 1) { (void)hipGetLastError(); wgs = (sizeof(F) == sizeof(Fq2) ? 2 : 3) * (256 / ZK_L1_BLOCK); }
        if (const char *o = probe_env("ZKHIP_ACC_ROUND_WGS")) wgs = atoi(o) > 0 ? atoi(o) : wgs;   // workgroups per CU (probe)
        return (uint64_t)wgs * (uint64_t)ZK_L1_BLOCK * (uint64_t)cus;
    }();
    uint64_t lanes = threads / LaneModel<F>::LPE / (n_msm ? n_msm : 1);
    return lanes ? lanes : 1;
}
template <class F>
static inline uint64_t accum_lanes_for(uint64_t max_entries, uint32_t n_msm, uint32_t chunk_min = 0, uint32_t chunk_max = 0) {
    if (!max_entries) max_entries = 1;
    const uint64_t R = accum_round_lanes<F>(n_msm), cmin = chunk_min ? chunk_min : accum_chunk_min(), cmax = chunk_max ? chunk_max : accum_chunk_max();
    if (max_entries <= R * cmin) return (max_entries + cmin - 1) / cmin;       // one partial round of minimum chunks
    const uint64_t rounds = (max_entries + R * cmax - 1) / (R * cmax);
    return rounds * R;
}
uint64_t msm_accum_workspace_slots(uint64_t max_entries) {
    // one size for whichever MSM uses the workspace: single G1, one of a G1 batch, G2
    uint64_t lanes = accum_lanes_for<Fq>(max_entries, 1);
    for (uint32_t nb = 2; nb <= 3; nb++) lanes = std::max(lanes, accum_lanes_for<Fq>(max_entries, nb));
    lanes = std::max(lanes, accum_lanes_for<Fq2>(max_entries, 1));
    uint64_t total = 0;
    for (;;) {
        uint64_t slots = 2 * lanes;
        total += slots;
        if (lanes == 1) break;
        lanes = (slots + ACC_CHUNK_N - 1) / ACC_CHUNK_N;
    }
    return total;
}

static bool skip_followups_probe() {      // ZKHIP_PROBE_SKIP_FOLLOWUPS (-DZK_PROBES builds only, WRONG results): no partial merges, no bucket reductions
    static const bool skip = probe_env("ZKHIP_PROBE_SKIP_FOLLOWUPS") != nullptr;
    return skip;
}

template <class F>
static void launch_accum(ACCMEM *buckets, const uint32_t *offsets, const uint32_t *entries, AccumBatch batch,
                         uint32_t total_buckets, uint64_t max_entries,
                         ACCMEM *ws_part, uint32_t *ws_key, uint32_t *ws_flag, hipStream_t s, hipEvent_t *ev, AccumTail tail) {
    const uint32_t nb = batch.n ? batch.n : 1;
    // empty buckets are never written by the kernels: infinity is the all-zero pattern
    if (!tail.buckets_zeroed) ZK_HIP(hipMemsetAsync(buckets, 0, ((size_t)(nb - 1) * batch.bucket_stride + total_buckets) * sizeof(ACCMEM), s));
    const uint32_t cmin = tail.chunk_min > accum_chunk_min() ? tail.chunk_min : accum_chunk_min();      // (never more lanes than the workspace was sized for)
    uint64_t lanes = accum_lanes_for<F>(max_entries, nb, cmin, tail.chunk_max > accum_chunk_max() ? tail.chunk_max : 0u);
    if (ev) ZK_HIP(hipEventRecord(ev[0], s));          // tight bracket around the level-1 kernel (roofline timing)
    if constexpr (sizeof(F) == sizeof(Fq2)) {
        ZK_LAUNCH(k_msm_accum_l1_g2s, dim3((uint32_t)((2 * lanes + ZK_L1_BLOCK - 1) / ZK_L1_BLOCK)), dim3(ZK_L1_BLOCK), 0, s, buckets, offsets, entries,
                           reinterpret_cast<const G2Affine *>(batch.points[0]), batch.idx_min[0], batch.idx_sub[0], total_buckets, ws_part, ws_key, ws_flag,
                           (uint32_t)lanes, cmin);
    } else {
        ZK_LAUNCH(k_msm_accum_l1<F>, dim3((uint32_t)((lanes + ZK_L1_BLOCK - 1) / ZK_L1_BLOCK), nb), dim3(ZK_L1_BLOCK), 0, s, buckets, offsets, entries,
                           batch, total_buckets, ws_part, ws_key, ws_flag, (uint32_t)lanes, cmin);
    }
    if (ev) ZK_HIP(hipEventRecord(ev[1], s));
    if (skip_followups_probe()) return;
    if (tail.stream && tail.stream != s) {            // partial merges continue on the caller's follow-up stream
        ZK_HIP(hipEventRecord(tail.l1_done, s));
        ZK_HIP(hipStreamWaitEvent(tail.stream, tail.l1_done, 0));
        s = tail.stream;
    }
    if (lanes > 1)
        ZK_LAUNCH(k_msm_accum_pair<F>, dim3((uint32_t)((lanes * LaneModel<F>::LPE + 255) / 256), nb), dim3(256), 0, s, buckets,
                           (const ACCMEM *)ws_part, ws_key, (const uint32_t *)ws_flag, (uint32_t)lanes, batch.bucket_stride, batch.ws_stride);
    uint64_t off = 0;
    while (lanes > 1) {          // a single unit has no cut runs: everything it saw was complete
        uint64_t items = 2 * lanes;
        uint64_t noff = off + items;
        const uint64_t epw = 64u / LaneModel<F>::LPE;
        const uint64_t nl = (items + epw - 1) / epw;             // waves; each emits two slots
        ZK_LAUNCH(k_msm_accum_wave<F>, dim3((uint32_t)((nl + 3) / 4), nb), dim3(256), 0, s, buckets, ws_part + off,
                           ws_key + off, ws_flag + off, (uint32_t)items, ws_part + noff, ws_key + noff, ws_flag + noff, (uint32_t)nl,
                           batch.bucket_stride, batch.ws_stride);
        off = noff;
        lanes = nl;
    }
    ZK_LAUNCH_OK("msm bucket accumulation");
}

static uint32_t gather_mask_env() {      // ZKHIP_GATHER_MASK (-DZK_PROBES builds only, WRONG results): confine the G1 table gathers to the low rows
    static const uint32_t m = [] { const char *e = probe_env("ZKHIP_GATHER_MASK"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 0xffffffffu; }();
    return m;
}
static AccumBatch single(const void *points, uint32_t idx_min, uint32_t idx_sub) {
    AccumBatch b;
    memset(&b, 0, sizeof b);
    b.n = 1;
    b.gather_mask = gather_mask_env();
    b.points[0] = points;
    b.idx_min[0] = idx_min;
    b.idx_sub[0] = idx_sub;
    return b;
}

void launch_msm_accum_g1(G1Acc *buckets, const uint32_t *offsets, const uint32_t *entries, const G1Affine *points,
                         uint32_t idx_min, uint32_t idx_sub, uint32_t total, uint64_t max_entries, G1Acc *ws_part,
                         uint32_t *ws_key, uint32_t *ws_flag, hipStream_t s, hipEvent_t *ev, AccumTail tail) {
    launch_accum<Fq>(buckets, offsets, entries, single(points, idx_min, idx_sub), total, max_entries, ws_part, ws_key, ws_flag, s, ev, tail);
}
void launch_msm_accum_g1_batch(G1Acc *buckets, const uint32_t *offsets, const uint32_t *entries, const AccumBatch &batch, uint32_t total,
                               uint64_t max_entries, G1Acc *ws_part, uint32_t *ws_key, uint32_t *ws_flag, hipStream_t s, hipEvent_t *ev,
                               AccumTail tail) {
    AccumBatch b = batch;
    b.gather_mask = gather_mask_env();
    launch_accum<Fq>(buckets, offsets, entries, b, total, max_entries, ws_part, ws_key, ws_flag, s, ev, tail);
}
void launch_msm_accum_g2(G2Acc *buckets, const uint32_t *offsets, const uint32_t *entries, const G2Affine *points,
                         uint32_t idx_min, uint32_t idx_sub, uint32_t total, uint64_t max_entries, G2Acc *ws_part,
                         uint32_t *ws_key, uint32_t *ws_flag, hipStream_t s, hipEvent_t *ev, AccumTail tail) {
    launch_accum<Fq2>(buckets, offsets, entries, single(points, idx_min, idx_sub), total, max_entries, ws_part, ws_key, ws_flag, s, ev, tail);
}

template <class F>
static void launch_reduce(XYZZ<F> *window_sums, ACCMEM *scratch, const ACCMEM *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s) {
    if (skip_followups_probe()) return;
    if (reduce_bits_for(p)) {          // c sums per bucket set (msm_wsum_rc), the host finishes
        const uint32_t NE = REDUCE_THREADS / LaneModel<F>::LPE, groups = n_msm * p.sets;
        const uint32_t nblk = p.nbuckets > NE ? p.nbuckets / NE : 1u;
        const size_t lds = REDUCE_THREADS * sizeof(XYZZ<typename LaneModel<F>::R>);
        ZK_LAUNCH(k_msm_reduce_bits_block<F>, dim3(nblk, groups), dim3(REDUCE_THREADS), lds, s, scratch, window_sums, buckets, p.nbuckets, p.c, nblk);
        if (nblk > 1) ZK_LAUNCH(k_msm_reduce_bits_top<F>, dim3(groups), dim3(REDUCE_THREADS), 0, s, window_sums, scratch, nblk, p.c);
        ZK_LAUNCH_OK("msm bucket reduction (bit sums)");
        return;
    }
    const uint32_t groups = n_msm * p.sets;
    const size_t lds = REDUCE_THREADS * sizeof(XYZZ<typename LaneModel<F>::R>);
    const uint32_t TREE_IN = tree_in<F>();
    auto chunks = [&](ACCMEM *out, uint32_t out_stride, const ACCMEM *in, uint32_t n, uint32_t chunk) {      // sum_k (k+1) in[k] over n points per set -> n / chunk shares
        const uint32_t total_chunks = groups * (n / chunk);
        ZK_LAUNCH(k_msm_reduce_chunks<F>, dim3((total_chunks * LaneModel<F>::LPE + 127) / 128), dim3(128), 0, s, out, out_stride, in, n, chunk, total_chunks);
    };
    if (const uint32_t ch = reduce_split_for(p)) {
        const uint32_t L = p.nbuckets / ch, top = reduce_split_top(L), nx = L / top, per = nx + L;      // per set: nx shares, then the L points of the A level
        ACCMEM *T = scratch, *Y = T + (uint64_t)groups * L, *P = Y + (uint64_t)groups * per;
        ZK_LAUNCH(k_msm_reduce_split<F>, dim3((L * LaneModel<F>::LPE + 127) / 128, groups), dim3(128), 0, s, Y + nx, per, T, buckets, p.nbuckets, ch, L);
        chunks(Y, per, T, L, top);
        const uint32_t bx = nx / TREE_IN, ba = L / TREE_IN;        // whole workgroups of each kind (reduce_split_for)
        ZK_LAUNCH(k_msm_reduce_tree<F>, dim3(bx + ba, groups), dim3(REDUCE_THREADS), lds, s, P, window_sums, (const ACCMEM *)Y, per, 0u);
        ZK_LAUNCH(k_msm_reduce_final2<F>, dim3(groups), dim3(REDUCE_THREADS), lds, s, window_sums, (const ACCMEM *)P, bx, ba, (uint32_t)__builtin_ctz(ch));
        ZK_LAUNCH_OK("msm bucket reduction (split)");
        return;
    }
    const uint32_t chunk = reduce_chunk_for(p);
    uint32_t cnt = p.nbuckets / chunk;
    chunks(scratch, cnt, buckets, p.nbuckets, chunk);
    ACCMEM *in = scratch;
    for (;;) {
        const uint32_t blocks = (cnt + TREE_IN - 1) / TREE_IN;
        const bool last = blocks == 1;
        ACCMEM *out = in + (uint64_t)groups * cnt;
        ZK_LAUNCH(k_msm_reduce_tree<F>, dim3(blocks, groups), dim3(REDUCE_THREADS), lds, s, out, window_sums, (const ACCMEM *)in, cnt, last ? 1u : 0u);
        if (last) break;
        in = out;
        cnt = blocks;
    }
    ZK_LAUNCH_OK("msm bucket reduction");
}
void launch_msm_reduce_g1(G1XYZZ *ws, G1Acc *scratch, const G1Acc *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s) {
    launch_reduce<Fq>(ws, scratch, buckets, n_msm, p, s);
}
void launch_msm_reduce_g2(G2XYZZ *ws, G2Acc *scratch, const G2Acc *buckets, uint32_t n_msm, MsmPlan p, hipStream_t s) {
    launch_reduce<Fq2>(ws, scratch, buckets, n_msm, p, s);
}

}   // namespace zk
