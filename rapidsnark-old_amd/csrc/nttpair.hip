// The three coset evaluations of a proof (src/groth16.cpp:98-155: ifft, x[i] *= w_2n^i, fft — for a, b and c) as ONE
// pipeline per polynomial batch, the way north_star describes the transform: radix-8 butterflies in REGISTERS between
// LDS exchanges, small root tables instead of one 48-byte global twiddle per butterfly, and as few trips through HBM as
// the 160 KiB of LDS allow.  Replaces k_ntt_pass (ntt.hip: radix-2, one LDS round trip per stage, three launches per
// transform at 2^22) on the proof path; ntt.hip keeps the stand-alone operators and the cross-GPU stages.
//
// Decomposition (four-step, n = 2^L = N2 * N1, N1 = 2^m <= 2^11 the contiguous "tile", N2 = 2^t the strided rest):
//   inverse = DIF (natural -> bit-reversed), forward = DIT (bit-reversed -> natural): no permutation pass.
//     k_ntt_outer<DIF> : clean size-N2 transforms along the high index bits (rows 2^m apart, 2^q contiguous columns)
//     k_ntt_mid        : element * Tinv'  ->  clean size-N1 DIF on the tile  ->  * D  ->  clean size-N1 DIT  ->  * T
//     k_ntt_outer<DIT> : clean size-N2 transforms along the high bits
//   "clean" = the twiddles of a stage are roots of the SUB-transform's own order: one table of w_4096^k serves every
//   size (2048 entries, L1/L2-resident); what the sub-transforms owe each other is collected in per-element tables that
//   stream beside the data:  T[p1][j0] = w_n^(j0 * brev_t(p1)),  Tinv'[p1][i0] = w_n^(-i0 * brev_t(p1)) * w_2n^(brev_t(p1)) * kappa
//   (the tile-constant part of the coset shift and the 1/n ride along), D[p0] = w_(2 N1)^(brev_m(p0)) (the rest of the
//   coset shift: a 2^m-entry table).  The last DIF pass and the first DIT pass work on the same contiguous tile, so they are
//   ONE kernel: a transform pair is 3 launches and 3 trips through HBM (192 B per element; before: 6 and 384 B).
//   t > 11 (n > 2^22): the high bits are two groups with a 2^t-entry table T2 between them (5 launches per pair).
//   One more multiplication per element and transform than radix-2 with full twiddles (T / Tinv'), paid back by the
//   lowest window of every clean sub-transform, whose twiddles are 1, w_4, w_8^k: constants, and 1 is skipped.
// A workgroup holds 2^11 (2^12 for an 11-bit outer pass) elements, eight per thread as nine 29-bit limbs each (72
// VGPRs); a "window" is the three index bits a thread's eight elements differ in: three stages run in registers,
// then the elements go through LDS (limb planes, XOR-swizzled: conflict-free) into the next window's layout; only the
// exchange at the top of a pass crosses waves (two barriers), the others stay inside a wave.
// A partitioned chain (zk_multi_prover / zk_shard_*) runs the same pipeline on its block with kappa = w_2n^(brev_g(r)) / n
// (ntt.hip's cross stages handle the log2(G) top bits).
#include <stdlib.h>
#include <string.h>
#include "kernels.hpp"
#include "hipcheck.hpp"
#include "common.hpp"
#include "field29.hpp"

namespace zk {

template <class F>
__device__ __forceinline__ F pload_el(const F *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 lo = q[0], hi = q[1];
    F r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
template <class F>
__device__ __forceinline__ void pstore_el(F *p, const F &r) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
__device__ __forceinline__ Fr29 pload_tw(const TwEntry *e) {
    const uint4 *q = reinterpret_cast<const uint4 *>(e);
    const uint4 a = q[0], b = q[1];
    const uint32_t c = e->l[8];
    Fr29 w;
    w.l[0] = (int32_t)a.x; w.l[1] = (int32_t)a.y; w.l[2] = (int32_t)a.z; w.l[3] = (int32_t)a.w;
    w.l[4] = (int32_t)b.x; w.l[5] = (int32_t)b.y; w.l[6] = (int32_t)b.z; w.l[7] = (int32_t)b.w;
    w.l[8] = (int32_t)c;
    return w;
}

// ---------------------------------------------------------------- windows
// v = index of an element inside its workgroup (VB bits).  In the window at bit `wlo` thread T holds the eight elements
// v = T's low wlo bits | k << wlo | T's other bits << (wlo + 3), k = 0..7.
__device__ __forceinline__ uint32_t v_of(uint32_t T, uint32_t k, uint32_t wlo) {
    return (T & ((1u << wlo) - 1u)) | (k << wlo) | ((T >> wlo) << (wlo + 3u));
}
__device__ __forceinline__ uint32_t pbrev_bits(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }
// LDS: nine limb planes of 8*NT words, element v at word v ^ (v[5..7] << 2): with this XOR every window layout from
// bit 2 up is bank-conflict free (32 lanes of a pass differ in v bits {0..wlo-1} and {wlo+3..7}: all of them reach distinct
// bank bits), and so is the bottom window (wlo = 0), whose eight elements per thread are contiguous and move as two
// 16-byte accesses per plane.  wlo = 1 keeps a two-way conflict.  No padding: 72 KiB per 2^11 elements, two workgroups per CU.
__device__ __forceinline__ uint32_t lds_at(uint32_t v) { return v ^ (((v >> 5) & 7u) << 2); }

// position of v-bit i among the bits of the thread index in the window at wlo
__device__ __forceinline__ uint32_t tbit_of(uint32_t i, uint32_t wlo) { return i < wlo ? i : i - 3u; }
// The threads that trade elements in the exchange wa -> wb differ, in layout wa, in the bits of window wb outside window
// wa (and vice versa): when those are lane bits (< 6) in BOTH layouts every wave only reads what it wrote itself, and no
// workgroup barrier is needed (LDS operations of a wave execute in order).
__device__ __forceinline__ bool exchange_is_intra_wave(uint32_t wa, uint32_t wb) {
    bool intra = true;
    for (uint32_t i = wb; i < wb + 3u; i++)
        if (i < wa || i >= wa + 3u) intra = intra && tbit_of(i, wa) < 6u;
    for (uint32_t i = wa; i < wa + 3u; i++)
        if (i < wb || i >= wb + 3u) intra = intra && tbit_of(i, wb) < 6u;
    return intra;
}

template <int NT>
__device__ __forceinline__ void exchange(int32_t *lds, Fr29 (&x)[8], uint32_t T, uint32_t wa, uint32_t wb) {
    constexpr uint32_t PL = 8u * NT;            // words per limb plane
    const bool intra = exchange_is_intra_wave(wa, wb);
    if (!intra) __syncthreads();                // everybody has read what the previous exchange left here
    if (wa == 0) {
        const uint32_t a0 = lds_at(v_of(T, 0, 0)), a1 = lds_at(v_of(T, 4, 0));
#pragma unroll
        for (int l = 0; l < 9; l++) {
            *reinterpret_cast<int4 *>(lds + l * PL + a0) = make_int4(x[0].l[l], x[1].l[l], x[2].l[l], x[3].l[l]);
            *reinterpret_cast<int4 *>(lds + l * PL + a1) = make_int4(x[4].l[l], x[5].l[l], x[6].l[l], x[7].l[l]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t a = lds_at(v_of(T, k, wa));
#pragma unroll
            for (int l = 0; l < 9; l++) lds[l * PL + a] = x[k].l[l];
        }
    }
    if (intra) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
    if (wb == 0) {
        const uint32_t a0 = lds_at(v_of(T, 0, 0)), a1 = lds_at(v_of(T, 4, 0));
#pragma unroll
        for (int l = 0; l < 9; l++) {
            const int4 lo = *reinterpret_cast<const int4 *>(lds + l * PL + a0), hi = *reinterpret_cast<const int4 *>(lds + l * PL + a1);
            x[0].l[l] = lo.x; x[1].l[l] = lo.y; x[2].l[l] = lo.z; x[3].l[l] = lo.w;
            x[4].l[l] = hi.x; x[5].l[l] = hi.y; x[6].l[l] = hi.z; x[7].l[l] = hi.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t a = lds_at(v_of(T, k, wb));
#pragma unroll
            for (int l = 0; l < 9; l++) x[k].l[l] = lds[l * PL + a];
        }
    }
    if (intra) __builtin_amdgcn_wave_barrier();
}

// One stage on register bit RB of the window: four butterflies.  level = index of the stage inside its clean
// sub-transform (bit `level` of the sub-transform's own index e = v >> plo); twiddle of the butterfly whose lower element has
// in-transform index e: w_(2^(level+1))^(e mod 2^level) = roots[(e mod 2^level) << (11 - level)].  The four lower elements
// differ in the register bits below RB (candidate c): at most 2^RB distinct twiddles.  When no thread bit lies between the
// bottom of the processed range and the window (uniform) e mod 2^level depends on c only: the twiddles are wave-uniform
// constants (1, w_4, w_8^k), and the butterflies whose twiddle is 1 skip the multiplication.
// Butterflies are multiplied in PAIRS (field29.hpp run2), one pair after the other (few values live at a time).
// (Measured and dropped: three waves per SIMD — 168 VGPRs, 44-256 B of scratch, the limb planes exchanged in two rounds
// through 40 KiB of LDS: mid pass 1.68 -> 1.65 ms, forward outer pass 0.91 -> 1.08 ms, pipelined proof unchanged; and
// loading a stage's four twiddles one stage ahead — 36 more live registers, 4 % more instructions,
// mid pass 1.64 -> 1.81 ms: the kernels are bound by VALU issue at ~4.7 cycles per instruction like the bucket kernels,
// not by the L2 trips of the twiddle loads.)
// Bounds (field29.hpp): DIF keeps every value carried ("tight": a difference of two tight values is a valid factor);
// DIT adds lazily and carries where the caller says (`carry_out`).
template <int RB, bool DIF>
__device__ __forceinline__ void stage(Fr29 (&x)[8], uint32_t level, uint32_t v0, uint32_t wlo, uint32_t plo, bool uniform, const TwEntry *roots,
                                      bool carry_out) {
    typedef Fr29 F;
    constexpr int NW = 1 << RB;                 // candidates
    // lower elements (bit RB clear), in pairs: (lk[0], lk[1]) and (lk[2], lk[3]) share their candidate when there are <= 2
    constexpr int lk[4] = {0, RB == 0 ? 2 : (RB == 1 ? 4 : 2), RB == 0 ? 4 : 1, RB == 0 ? 6 : (RB == 1 ? 5 : 3)};
    constexpr int hb = 1 << RB;
    if (level == 0) {                           // w = 1 for every butterfly
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const F u = x[lk[i]], v = x[lk[i] | hb];
            x[lk[i]] = F::add(u, v);
            x[lk[i] | hb] = F::sub(u, v);
        }
        return;
    }
    const uint32_t mask = (1u << level) - 1u, tshift = 11u - level;
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        const int ka = lk[i], kb = lk[i + 1];
        const uint32_t ca = ((uint32_t)(ka & (NW - 1)) << wlo) >> plo, cb = ((uint32_t)(kb & (NW - 1)) << wlo) >> plo;     // wave-uniform
        const uint32_t ja = ((v0 >> plo) | ca) & mask, jb = ((v0 >> plo) | cb) & mask;
        const bool onea = uniform && (ca & mask) == 0, oneb = uniform && (cb & mask) == 0;     // (only the first of a pair can be 1 beside a different one)
        F fa, fb, ra, rb;                       // f = the factor that meets the twiddle, r = the product
        if (DIF) {
            fa = F::sub_nc(x[ka], x[ka | hb]);
            x[ka] = F::add(x[ka], x[ka | hb]);
            fb = F::sub_nc(x[kb], x[kb | hb]);
            x[kb] = F::add(x[kb], x[kb | hb]);
        } else {
            fa = x[ka | hb];
            fb = x[kb | hb];
        }
        if (onea && oneb) {
            ra = DIF ? F::carry(fa) : fa;
            rb = DIF ? F::carry(fb) : fb;
        } else if (onea) {
            ra = DIF ? F::carry(fa) : fa;
            rb = F::mul(fb, pload_tw(roots + ((size_t)jb << tshift)));
        } else {
            const F wa = pload_tw(roots + ((size_t)ja << tshift)), wb = pload_tw(roots + ((size_t)jb << tshift));
            F::mul2(ra, fa, wa, rb, fb, wb);
        }
        if (DIF) {
            x[ka | hb] = ra;
            x[kb | hb] = rb;
        } else {
            const F ua = x[ka], ub = x[kb];
            x[ka] = carry_out ? F::add(ua, ra) : F::add_nc(ua, ra);
            x[ka | hb] = carry_out ? F::sub(ua, ra) : F::sub_nc(ua, ra);
            x[kb] = carry_out ? F::add(ub, rb) : F::add_nc(ub, rb);
            x[kb | hb] = carry_out ? F::sub(ub, rb) : F::sub_nc(ub, rb);
        }
    }
}

// The stages of one window: bits [sb, sb + ns) of v, ascending for DIT, descending for DIF.  plo = lowest bit of the
// processed range (the clean sub-transform's bit 0).
template <bool DIF>
__device__ __forceinline__ void run_window(Fr29 (&x)[8], uint32_t T, uint32_t wlo, uint32_t sb, uint32_t ns, uint32_t plo, const TwEntry *roots) {
    const uint32_t v0 = v_of(T, 0, wlo);
    const bool uniform = wlo <= plo;                                 // no thread bits between plo and the window
    for (uint32_t s = 0; s < ns; s++) {
        const uint32_t vb = DIF ? sb + ns - 1 - s : sb + s;
        const uint32_t rb = vb - wlo, level = vb - plo;
        // DIT: values are added lazily in the first stage of a window and carried after the second and the last one
        const bool carry_out = DIF || (s & 1u) || s + 1 == ns;
        if (rb == 0) stage<0, DIF>(x, level, v0, wlo, plo, uniform, roots, carry_out);
        else if (rb == 1) stage<1, DIF>(x, level, v0, wlo, plo, uniform, roots, carry_out);
        else stage<2, DIF>(x, level, v0, wlo, plo, uniform, roots, carry_out);
    }
}

struct PhasePlan {          // windows in DIT (ascending) order; DIF walks them backwards
    uint32_t nph;
    uint8_t wlo[6], sb[6], ns[6];
};
static PhasePlan plan_windows(uint32_t plo, uint32_t phi, uint32_t VB) {
    PhasePlan p;
    memset(&p, 0, sizeof p);
    const uint32_t len = phi - plo, nfull = len / 3, rem = len % 3;
    uint32_t b = plo;
    if (rem) {
        p.wlo[p.nph] = (uint8_t)(plo < VB - 3 ? plo : VB - 3);
        p.sb[p.nph] = (uint8_t)plo;
        p.ns[p.nph] = (uint8_t)rem;
        p.nph++;
        b += rem;
    }
    for (uint32_t i = 0; i < nfull; i++) {
        p.wlo[p.nph] = (uint8_t)b;
        p.sb[p.nph] = (uint8_t)b;
        p.ns[p.nph] = 3;
        p.nph++;
        b += 3;
    }
    return p;
}

// DIF: the all-sums output of a window (element k = 0) has grown 8-fold: back to (-p, p).  Everything else has passed
// through a product inside the window and is below 8p.
__device__ __forceinline__ void dif_settle(Fr29 (&x)[8]) { x[0] = Fr29::reduce_near_zero(x[0]); }

template <bool DIF, int NT>
__device__ __forceinline__ void run_plan(Fr29 (&x)[8], int32_t *lds, uint32_t T, const PhasePlan &pl, uint32_t plo, const TwEntry *roots,
                                         uint32_t &wcur) {
    for (uint32_t i = 0; i < pl.nph; i++) {
        const uint32_t ph = DIF ? pl.nph - 1 - i : i;
        const uint32_t wlo = pl.wlo[ph];
        if (wlo != wcur) {
            exchange<NT>(lds, x, T, wcur, wlo);
            wcur = wlo;
        }
        run_window<DIF>(x, T, wlo, pl.sb[ph], pl.ns[ph], plo, roots);
        if (DIF) dif_settle(x);
    }
}

struct PairDev {
    uint32_t m, tbits;           // tile bits, strided bits (L - m)
    uint32_t batch, tiles;       // vectors per launch, workgroups per vector
    uint64_t nloc;               // elements per vector
    const TwEntry *rfwd, *rinv;  // w_4096^k, w_4096^-k, k < 2048
    const Fr *tinv, *tfwd;       // per-element tables (nullptr when the transform is a single tile)
    const Fr *dtab;              // 2^m entries
    PhasePlan plan;
};

// ---------------------------------------------------------------- the middle kernel: * Tinv', DIF, * D, DIT, * T on contiguous tiles
// MODE 0: the proof's pair, as described above.  MODE 1 / 2: ONE transform in natural order on both sides (the stand-alone
// operator zk_fr_ntt, the counterpart of FFT::ifft / FFT::fft, src/groth16.cpp:102,115): 1 = the inverse's last pass — * Tinv
// (1/n inside), DIF, and the results STORED at their natural positions; 2 = the forward's first pass — elements GATHERED
// from their natural positions, DIT, * T.  Position p = tile << m | p0 of the bit-reversed order holds index
// brev_m(p0) << t | brev_t(tile): the permutation costs no pass, only the coalescing of one side of this kernel (32-byte
// accesses 2^t elements apart).
// MODE 1 / 2 permute across workgroups: they read `src` and write `data`, two DIFFERENT buffers (MODE 0: the same, in place).
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_ntt_mid(Fr *data, const Fr *src, uint64_t stride_elems, PairDev t) {
    ZK_CHAIN_PRIO();
    extern __shared__ int32_t lds[];
    typedef Fr29 F;
    // The vectors (a, b, c) of a tile run next to each other ON THE SAME XCD (workgroup i goes to XCD i mod 8, each XCD has
    // its own L2): the per-element tables Tinv' / T of the tile are fetched from HBM once and found in L2 by the other two.
    // id = (g, x) with x = id mod 8: tile = 8 * (g / batch) + x, vector = g mod batch (tile counts that are multiples of 8).
    uint32_t vec, tile;
    if ((t.tiles & 7u) == 0) {
        const uint32_t g = blockIdx.x >> 3, x = blockIdx.x & 7u;
        vec = g % t.batch;
        tile = ((g / t.batch) << 3) | x;
    } else {
        vec = blockIdx.x % t.batch;
        tile = blockIdx.x / t.batch;
    }
    Fr *xg = data + (uint64_t)vec * stride_elems;
    const Fr *xs = MODE == 0 ? xg : src + (uint64_t)vec * stride_elems;
    const uint32_t T = threadIdx.x;
    const uint64_t wg_base = (uint64_t)tile << 11;
    const uint32_t wtop = t.plan.wlo[t.plan.nph - 1];            // DIF starts (and DIT ends) in the top window: coalesced
    const uint32_t wfirst = MODE == 2 ? t.plan.wlo[0] : wtop;
    F x[8];
    // nloc < 2^11: the tail of the only workgroup idles (but meets the barriers); the eight elements of a thread differ in
    // bits below m, so they are active together
    const bool active = wg_base + v_of(T, 0, wfirst) < t.nloc && wg_base + v_of(T, 7, wfirst) < t.nloc;
    auto natural = [&](uint32_t v) -> uint64_t {                 // natural index of the element at tile position v (MODE 1, 2)
        return ((uint64_t)pbrev_bits(v & ((1u << t.m) - 1u), t.m) << t.tbits) | pbrev_bits(tile, t.tbits);
    };
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t v = v_of(T, k, wfirst);
        const uint64_t pos = MODE == 2 ? natural(v) : wg_base + v;
        x[k] = active ? F::load(pload_el(xs + pos)) : F::zero();
    }
    if (MODE != 2 && t.tinv) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const uint64_t p0 = wg_base + v_of(T, k, wtop), p1 = wg_base + v_of(T, k + 1, wtop);
            const F a = F::load(pload_el(t.tinv + (active ? p0 : 0))), b = F::load(pload_el(t.tinv + (active ? p1 : 0)));
            F::mul2(x[k], x[k], a, x[k + 1], x[k + 1], b);
        }
    }
    uint32_t wcur = wfirst;
    if (MODE != 2) run_plan<true, 256>(x, lds, T, t.plan, 0, t.rinv, wcur);
    if (MODE == 0 || (MODE == 1 && !t.tinv)) {   // coset shift inside the tile (and kappa when there is no Tinv'): position p0 of the tile
        const uint32_t tmask = (1u << t.m) - 1u;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const uint32_t q0 = v_of(T, k, wcur) & tmask, q1 = v_of(T, k + 1, wcur) & tmask;
            const F a = F::load(pload_el(t.dtab + q0)), b = F::load(pload_el(t.dtab + q1));
            F::mul2(x[k], x[k], a, x[k + 1], x[k + 1], b);
        }
    }
    if (MODE != 1) run_plan<false, 256>(x, lds, T, t.plan, 0, t.rfwd, wcur);
    if (MODE != 1 && t.tfwd) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const uint64_t p0 = wg_base + v_of(T, k, wcur), p1 = wg_base + v_of(T, k + 1, wcur);
            const F a = F::load(pload_el(t.tfwd + (active ? p0 : 0))), b = F::load(pload_el(t.tfwd + (active ? p1 : 0)));
            F::mul2(x[k], x[k], a, x[k + 1], x[k + 1], b);
        }
    }
    if (active) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t v = v_of(T, k, wcur);
            pstore_el(xg + (MODE == 1 ? natural(v) : wg_base + v), F::store(x[k]));
        }
    }
}

// ---------------------------------------------------------------- the outer kernels: clean transforms along bits [lo, lo + tt)
struct OuterDev {
    uint32_t lo, tt, q;          // rows 2^lo apart, 2^tt rows, 2^q contiguous columns: 2^(tt+q) = 8 * NT elements per workgroup
    const TwEntry *roots;
    const Fr *tab;               // optional 2^tmask_bits-entry table indexed by (position >> tab_shift) & tab_mask: pre (DIF) / post (DIT)
    uint32_t tab_shift, tab_mask;
    PhasePlan plan;
};
template <bool DIF, int NT>
__global__ __launch_bounds__(NT, 2) void k_ntt_outer(Fr *data, uint64_t stride_elems, OuterDev o) {
    ZK_CHAIN_PRIO();
    extern __shared__ int32_t lds[];
    typedef Fr29 F;
    Fr *xg = data + (uint64_t)blockIdx.y * stride_elems;
    const uint32_t T = threadIdx.x;
    const uint32_t midw = o.lo - o.q;
    const uint64_t tile = blockIdx.x;
    const uint64_t base = ((tile >> midw) << (o.lo + o.tt)) | ((tile & ((1ull << midw) - 1ull)) << o.q);
    const uint32_t cmask = (1u << o.q) - 1u;
    auto pos_of = [&](uint32_t v) { return base | ((uint64_t)(v >> o.q) << o.lo) | (v & cmask); };
    const uint32_t wfirst = DIF ? o.plan.wlo[o.plan.nph - 1] : o.plan.wlo[0];
    F x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = F::load(pload_el(xg + pos_of(v_of(T, k, wfirst))));
    if (DIF && o.tab) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const F a = F::load(pload_el(o.tab + ((pos_of(v_of(T, k, wfirst)) >> o.tab_shift) & o.tab_mask)));
            const F b = F::load(pload_el(o.tab + ((pos_of(v_of(T, k + 1, wfirst)) >> o.tab_shift) & o.tab_mask)));
            F::mul2(x[k], x[k], a, x[k + 1], x[k + 1], b);
        }
    }
    uint32_t wcur = wfirst;
    run_plan<DIF, NT>(x, lds, T, o.plan, o.q, o.roots, wcur);
    if (!DIF && o.tab) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const F a = F::load(pload_el(o.tab + ((pos_of(v_of(T, k, wcur)) >> o.tab_shift) & o.tab_mask)));
            const F b = F::load(pload_el(o.tab + ((pos_of(v_of(T, k + 1, wcur)) >> o.tab_shift) & o.tab_mask)));
            F::mul2(x[k], x[k], a, x[k + 1], x[k + 1], b);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) pstore_el(xg + pos_of(v_of(T, k, wcur)), F::store(x[k]));
}

// ---------------------------------------------------------------- tables
__device__ __constant__ const uint32_t PROOT_2_28_STD[8] = {0x725b19f0u, 0x9bd61b6eu, 0x41112ed4u, 0x402d111eu,
                                                            0x8ef62abcu, 0x00e0a7ebu, 0xa58a7e85u, 0x2a3c09f0u};
__device__ Fr pfr_pow(Fr base, uint64_t e) {
    Fr r = Fr::one();
    while (e) {
        if (e & 1) r = Fr::mul(r, base);
        base = Fr::sqr(base);
        e >>= 1;
    }
    return r;
}
__device__ Fr pfr_root(uint32_t k) {          // w_(2^k), Montgomery (2^256) form
    Fr w;
#pragma unroll
    for (int i = 0; i < 8; i++) w.v[i] = PROOT_2_28_STD[i];
    w = Fr::to_mont(w);
    for (uint32_t i = k; i < 28; i++) w = Fr::sqr(w);
    return w;
}
__device__ __forceinline__ uint32_t pbrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }
__device__ __forceinline__ void pstore_tw(TwEntry *e, const Fr29 &v) {
    uint4 *q = reinterpret_cast<uint4 *>(e);
    q[0] = make_uint4((uint32_t)v.l[0], (uint32_t)v.l[1], (uint32_t)v.l[2], (uint32_t)v.l[3]);
    q[1] = make_uint4((uint32_t)v.l[4], (uint32_t)v.l[5], (uint32_t)v.l[6], (uint32_t)v.l[7]);
    q[2] = make_uint4((uint32_t)v.l[8], 0u, 0u, 0u);
}

// L = bits of the local transform, Lg = bits of the whole domain (L < Lg: one block of a partitioned chain, rho = brev of
// the block index), m / t = tile / strided bits, ta = low group of the strided bits when they are split (else t)
// plain: tables of the stand-alone transforms (no coset shift: Tinv = w_n^(-i0 brev(p1)) / n, D = 1 — or 1/n when there is no Tinv)
__global__ __launch_bounds__(256) void k_pair_tables(TwEntry *rfwd, TwEntry *rinv, Fr *tinv, Fr *tfwd, Fr *dtab, Fr *t2inv, Fr *t2fwd,
                                                     uint32_t L, uint32_t Lg, uint32_t rho, uint32_t m, uint32_t ta, uint32_t plain) {
    const uint32_t t = L - m;
    __shared__ Fr s_w12, s_w12i, s_wn, s_wni, s_w2n, s_kappa, s_w2m, s_wt, s_wti;
    if (threadIdx.x == 0) {
        s_w12 = pfr_root(12);
        s_w12i = Fr::inv(s_w12);
        s_wn = pfr_root(L);
        s_wni = Fr::inv(s_wn);
        s_w2n = pfr_root(L + 1);                                  // root of order 2 n_loc
        // kappa = w_(2 n_global)^rho / n_global
        Fr nn = Fr::zero();
        nn.v[0] = (uint32_t)(1ull << Lg);
        nn.v[1] = (uint32_t)((1ull << Lg) >> 32);
        s_kappa = Fr::mul(pfr_pow(pfr_root(Lg + 1), rho), Fr::inv(Fr::to_mont(nn)));
        s_w2m = pfr_root(m + 1);
        s_wt = pfr_root(t);
        s_wti = Fr::inv(s_wt);
    }
    __syncthreads();
    const uint64_t n = 1ull << L, N1 = 1ull << m;
    const uint64_t st = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n || i < 2048; i += st) {
        if (i < 2048) {
            pstore_tw(rfwd + i, Fr29::canonical(Fr29::from_mont256(pfr_pow(s_w12, i))));
            pstore_tw(rinv + i, Fr29::canonical(Fr29::from_mont256(pfr_pow(s_w12i, i))));
        }
        if (i < N1) {
            Fr d = plain ? Fr::one() : pfr_pow(s_w2m, pbrev((uint32_t)i, m));
            if (t == 0) d = Fr::mul(d, s_kappa);
            pstore_el(dtab + i, Fr29::store(Fr29::from_mont256(d)));
        }
        if (t && i < n) {
            const uint64_t i0 = i & (N1 - 1), k1 = pbrev((uint32_t)(i >> m), t);
            const uint64_t e = (i0 * k1) & (n - 1);
            pstore_el(tfwd + i, Fr29::store(Fr29::from_mont256(pfr_pow(s_wn, e))));
            pstore_el(tinv + i, Fr29::store(Fr29::from_mont256(Fr::mul(plain ? pfr_pow(s_wni, e) : Fr::mul(pfr_pow(s_wni, e), pfr_pow(s_w2n, k1)), s_kappa))));
        }
        if (t2fwd && i < (1ull << t)) {
            const uint32_t tb = t - ta;
            const uint64_t ja = i & ((1ull << ta) - 1), kb = pbrev((uint32_t)(i >> ta), tb);
            const uint64_t e = (ja * kb) & ((1ull << t) - 1);
            pstore_el(t2fwd + i, Fr29::store(Fr29::from_mont256(pfr_pow(s_wt, e))));
            pstore_el(t2inv + i, Fr29::store(Fr29::from_mont256(pfr_pow(s_wti, e))));
        }
    }
}

// ---------------------------------------------------------------- host
bool ntt_pair_supported(uint32_t local_logn) { return local_logn >= 3 && local_logn <= 27; }

void NttPair::build(uint32_t logn_global, uint32_t logn_local, uint32_t block_index, hipStream_t s, bool plain) {
    L = logn_local;
    Lg = logn_global;
    m = L < 11 ? L : 11;
    const uint32_t t = L - m;
    // strided bits: one pass up to 11 bits (10: 256 threads and >= 64-byte runs; 11: 512 threads, 64-byte runs), else two
    ngroups = t == 0 ? 0 : (t <= 11 ? 1 : 2);
    if (const char *e = probe_env("ZKHIP_NTT_SPLIT")) {
        if (atoi(e) == 2 && t >= 2) ngroups = 2;
    }
    g[0] = ngroups == 2 ? (t + 1) / 2 : t;
    g[1] = t - g[0];
    const uint64_t n = 1ull << L;
    auto dalloc = [&](void **p, size_t bytes) { ZK_HIP(hipMalloc(p, bytes ? bytes : 16)); };
    release();
    dalloc((void **)&rfwd, 2048 * sizeof(TwEntry));
    dalloc((void **)&rinv, 2048 * sizeof(TwEntry));
    dalloc((void **)&dtab, ((size_t)1 << m) * sizeof(Fr));
    if (t) {
        dalloc((void **)&tinv, n * sizeof(Fr));
        dalloc((void **)&tfwd, n * sizeof(Fr));
    }
    if (ngroups == 2) {
        dalloc((void **)&t2inv, ((size_t)1 << t) * sizeof(Fr));
        dalloc((void **)&t2fwd, ((size_t)1 << t) * sizeof(Fr));
    }
    const uint32_t lg = Lg - L;                       // log2(blocks)
    uint32_t rho = 0;
    for (uint32_t i = 0; i < lg; i++) rho |= ((block_index >> i) & 1u) << (lg - 1 - i);
    uint64_t grid = ((n > 2048 ? n : 2048) + 255) / 256;
    if (grid > 4096) grid = 4096;
    ZK_LAUNCH(k_pair_tables, dim3((uint32_t)grid), dim3(256), 0, s, rfwd, rinv, tinv, tfwd, dtab, t2inv, t2fwd, L, Lg, rho, m, g[0], plain ? 1u : 0u);
    ZK_LAUNCH_OK("ntt pair tables");
}
void NttPair::release() {
    for (void *p : {(void *)rfwd, (void *)rinv, (void *)tinv, (void *)tfwd, (void *)dtab, (void *)t2inv, (void *)t2fwd})
        if (p) (void)hipFree(p);
    rfwd = rinv = nullptr;
    tinv = tfwd = dtab = t2inv = t2fwd = nullptr;
}

// the passes use 72 KiB (2048-element tiles) or 144 KiB (4096) of dynamic LDS: above 64 KiB a kernel needs the opt-in
static void lds_opt_in() {
    static PerDeviceOnce attr;
    if (!attr.need()) return;
    ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_outer<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_outer<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_outer<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_outer<false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_mid<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_mid<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_mid<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr.done();
}

template <bool DIF>
static void run_outer(Fr *data, uint64_t stride, uint32_t batch, const NttPair &tb, uint32_t lo, uint32_t tt, const Fr *tab, uint32_t tab_bits,
                      hipStream_t s) {
    OuterDev o;
    memset(&o, 0, sizeof o);
    const uint32_t VB = tt == 11 ? 12 : 11;
    o.lo = lo;
    o.tt = tt;
    o.q = VB - tt;
    o.roots = DIF ? tb.rinv : tb.rfwd;
    o.tab = tab;
    o.tab_shift = tb.m;
    o.tab_mask = (1u << tab_bits) - 1u;
    o.plan = plan_windows(o.q, VB, VB);
    const uint32_t tiles = (uint32_t)((1ull << tb.L) >> VB);
    lds_opt_in();
    if (VB == 12) {
        const size_t shmem = (size_t)9 * 4096 * 4;
        ZK_LAUNCH((k_ntt_outer<DIF, 512>), dim3(tiles, batch), dim3(512), shmem, s, data, stride, o);
    } else {
        const size_t shmem = (size_t)9 * 2048 * 4;
        ZK_LAUNCH((k_ntt_outer<DIF, 256>), dim3(tiles, batch), dim3(256), shmem, s, data, stride, o);
    }
    ZK_LAUNCH_OK("ntt outer pass");
}

template <int MODE>
static void run_mid(Fr *data, const Fr *src, uint64_t stride, uint32_t batch, const NttPair &tb, hipStream_t s) {
    const uint32_t t = tb.L - tb.m;
    lds_opt_in();
    PairDev d;
    memset(&d, 0, sizeof d);
    d.m = tb.m;
    d.tbits = t;
    d.batch = batch;
    d.nloc = 1ull << tb.L;
    d.rfwd = tb.rfwd;
    d.rinv = tb.rinv;
    d.tinv = t ? tb.tinv : nullptr;
    d.tfwd = t ? tb.tfwd : nullptr;
    d.dtab = tb.dtab;
    d.plan = plan_windows(0, tb.m, 11);
    const uint32_t wgs = (uint32_t)(((1ull << tb.L) + 2047) >> 11);
    d.tiles = wgs;
    const size_t shmem = (size_t)9 * 2048 * 4;
    ZK_LAUNCH(k_ntt_mid<MODE>, dim3(wgs * batch), dim3(256), shmem, s, data, src, stride, d);
    ZK_LAUNCH_OK("ntt middle pass");
}
template <bool DIF>
static void run_outer_passes(Fr *data, uint64_t stride, uint32_t batch, const NttPair &tb, hipStream_t s) {
    const uint32_t t = tb.L - tb.m;
    if (DIF) {
        if (tb.ngroups == 2) {
            run_outer<true>(data, stride, batch, tb, tb.m + tb.g[0], tb.g[1], nullptr, 0, s);
            run_outer<true>(data, stride, batch, tb, tb.m, tb.g[0], tb.t2inv, t, s);
        } else if (tb.ngroups == 1) {
            run_outer<true>(data, stride, batch, tb, tb.m, t, nullptr, 0, s);
        }
    } else {
        if (tb.ngroups == 2) {
            run_outer<false>(data, stride, batch, tb, tb.m, tb.g[0], tb.t2fwd, t, s);
            run_outer<false>(data, stride, batch, tb, tb.m + tb.g[0], tb.g[1], nullptr, 0, s);
        } else if (tb.ngroups == 1) {
            run_outer<false>(data, stride, batch, tb, tb.m, t, nullptr, 0, s);
        }
    }
}

// a | b | c (batch vectors `stride` elements apart, nloc = 2^L elements each): in place, natural order in and out
void launch_ntt_coset_pair(Fr *data, uint64_t stride, uint32_t batch, const NttPair &tb, hipStream_t s) {
    run_outer_passes<true>(data, stride, batch, tb, s);
    run_mid<0>(data, data, stride, batch, tb, s);
    run_outer_passes<false>(data, stride, batch, tb, s);
}

// ONE transform per vector, natural order in and out, on the same passes (tables built with plain = true): the inverse
// (1/n included) is the outer DIF passes + the middle pass storing at natural positions, the forward the middle pass
// gathering from natural positions + the outer DIT passes.  No permutation pass, no per-call table build.
// The permutation crosses workgroups, so the middle pass works out of place: the result is left in `out` (as large as `data`,
// which is used as scratch).
void launch_ntt_plain(Fr *out, Fr *data, uint64_t stride, uint32_t batch, const NttPair &tb, bool inverse, hipStream_t s) {
    if (inverse) {
        run_outer_passes<true>(data, stride, batch, tb, s);
        run_mid<1>(out, data, stride, batch, tb, s);
    } else {
        run_mid<2>(out, data, stride, batch, tb, s);
        run_outer_passes<false>(out, stride, batch, tb, s);
    }
}

}   // namespace zk
