// BN254 Fq in 9 x 29-bit SIGNED unsaturated limbs — the register representation of the MSM
// kernels (replaces ffiasm's RawFq ADX assembly on the hot loop).
//
// Why: measured on gfx950 (profiles/r01_ubench_valu.txt) v_addc_co_u32 costs as much as
// v_mad_u64_u32 (~4 cycles per wave64), so the saturated 8x32 CIOS product (128 MADs + ~250
// carry-chain adds, field.hpp) spends more time on carries than on multiplies.  With 29-bit
// limbs a whole column of the product accumulates in one 64-bit register through
// v_mad_i64_i32 with NO carry instruction: 81 + 81 MADs per Montgomery product, one shift per
// column.  Limbs are signed so a subtraction is 9 plain subtracts (no borrow chain, no +kp).
//
// Conventions
//   value  = sum l[i] * 2^(29 i); limbs 0..7 kept within [-8, 2^29 + 8] by a one-step parallel
//            carry after every add/sub; limb 8 is signed and absorbs the excess.
//   range  : any value in (-16p, 16p) is a valid operand.  The Montgomery radix is
//            R' = 2^261 (9 limbs), ~169 p, so a product of two such operands lands in (-p, 2p).
//   memory : coordinates live in HBM as canonical 256-bit integers in [0, p) holding x * 2^261
//            (tables are converted from the zkey's x * 2^256 once, at zk_prover_create).
//   zero   : is_zero() is the exact test (value = 0 mod p); is_zero_raw() tests the limbs and
//            is valid for canonical values and for the explicit infinity encoding.
#pragma once
#include "field.hpp"

namespace zk {

struct Fq29Params {
    typedef Fq Words;   // the 8x32-bit container of the same field (HBM format)
    static constexpr int32_t P[9] = {410844487, 17064118, 477274959, 47522512, 361093496, 47923392, 10936641, 240920116, 3171406};
    static constexpr uint32_t N0INV = 75916169u;   // -p^-1 mod 2^29
    static constexpr int32_t ONE[9] = {360500257, 337389400, 408039635, 21759001, 178483129, 490881230, 299191303, 86689704, 903222};      // 2^261 mod p
    static constexpr int32_t K_IN[9] = {322215073, 442336424, 171859116, 268585440, 314135016, 244503300, 348886451, 68918589, 360451};    // 2^266 mod p
    static constexpr int32_t K_OUT[9] = {93261213, 451550318, 297979764, 299258347, 342016167, 297253948, 482187706, 406012155, 920183};   // 2^256 mod p
};
struct Fr29Params {
    typedef Fr Words;
    static constexpr int32_t P[9] = {268435457, 521120927, 240919632, 131109107, 361091715, 47923392, 10936641, 240920116, 3171406};
    static constexpr uint32_t N0INV = 268435455u;  // -r^-1 mod 2^29
    static constexpr int32_t ONE[9] = {268435287, 514263732, 86771339, 391139145, 178784091, 490881230, 299191303, 86689704, 903222};       // 2^261 mod r
    static constexpr int32_t K_IN[9] = {268430039, 492061940, 71535269, 62181526, 323781850, 244503300, 348886451, 68918589, 360451};      // 2^266 mod r
    static constexpr int32_t K_OUT[9] = {268435451, 78749922, 406014571, 418196286, 342025071, 297253948, 482187706, 406012155, 920183};   // 2^256 mod r
};

// ---- The multiply-accumulate of the column chains: three builds of the same arithmetic.
// Left as C (acc += (int64_t)a * b) hipcc (i) splits every column into independent chains and merges them with a
// 64-bit add per column (17 extra 4-cycle instructions per product) and (ii) — whenever known-bits analysis proves one
// factor non-negative (a limb just masked out of a word) — rewrites sext x sext as sext x zext, which no longer matches
// v_mad_i64_i32 and is expanded into v_mad_u64_u32 plus a correction MAD for the sign word (seen: 386 extra MADs per G1
// addition after an unrelated change to the loop around it; the empty-asm barriers in from_words / out_limb hide the sign).
//   default        : PAIRS of independent products (run2) as blocks of inline-asm MADs with the two chains alternating
//                    (mad_blocks.inc; hipcc puts an s_nop behind every asm statement whose result a later asm reads, so
//                    the unit is a block, not a MAD); lone products (run1) as C column sums
//   ZK_STMT_MAD    : one asm statement per MAD everywhere (measured slower: profiles/r03a_mul_rate.txt)
//   ZK_COMPILER_MAD: C column sums everywhere
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_COMPILER_MAD)
#if defined(ZK_STMT_MAD)
#define ZK_ASM_MAD 1
#else
#define ZK_BLOCK_MAD 1
#endif
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_STMT_MAD)
#define ZK_SIGN_BARRIER(x) asm("" : "+v"(x))
#else
#define ZK_SIGN_BARRIER(x) ((void)0)
#endif
#if defined(ZK_ASM_MAD)
__device__ __forceinline__ int64_t zk_mad(int64_t acc, int32_t a, int32_t b) {
    uint64_t cy;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "v"(b));
    return acc;
}
__device__ __forceinline__ int64_t zk_mad0(int32_t a, int32_t b) {        // first term of a chain: addend = inline constant 0
    int64_t acc;
    uint64_t cy;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0" : "=v"(acc), "=s"(cy) : "v"(a), "v"(b));
    return acc;
}
__device__ __forceinline__ int64_t zk_mad_s(int64_t acc, int32_t a, int32_t s) {      // s: wave-uniform (modulus limb) in an SGPR
    uint64_t cy;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "s"(s));
    return acc;
}
#elif defined(ZK_CHECK_COLUMNS) && !defined(__HIP_DEVICE_COMPILE__)
// host-only checking build (tools/field29_bounds_test.cpp): every MAD in 128 bits, overflows of the 64-bit column counted
inline long &zk_column_overflows() { static long n = 0; return n; }
inline __int128 &zk_column_peak() { static __int128 v = 0; return v; }
inline int64_t zk_mad(int64_t acc, int32_t a, int32_t b) {
    const __int128 v = (__int128)acc + (__int128)a * b, m = v < 0 ? -v : v;
    if (m > zk_column_peak()) zk_column_peak() = m;
    if (v > (__int128)INT64_MAX || v < (__int128)INT64_MIN) zk_column_overflows()++;
    return (int64_t)v;
}
inline int64_t zk_mad0(int32_t a, int32_t b) { return zk_mad(0, a, b); }
inline int64_t zk_mad_s(int64_t acc, int32_t a, int32_t s) { return zk_mad(acc, a, s); }
#else
ZK_HD int64_t zk_mad(int64_t acc, int32_t a, int32_t b) { return acc + (int64_t)a * b; }
ZK_HD int64_t zk_mad0(int32_t a, int32_t b) { return (int64_t)a * b; }
ZK_HD int64_t zk_mad_s(int64_t acc, int32_t a, int32_t s) { return acc + (int64_t)a * s; }
#endif
#if defined(ZK_BLOCK_MAD)
#include "mad_blocks.inc"
#endif

template <class PR>
struct Fp29 {
    int32_t l[9];
    typedef typename PR::Words Words;
    typedef Fp29 Fq29;   // (keeps the member bodies below readable: "Fq29" = this instantiation)

    static constexpr int32_t MASK = (1 << 29) - 1;
    static constexpr const int32_t (&P)[9] = PR::P;
    static constexpr uint32_t N0INV = PR::N0INV;
    static constexpr const int32_t (&ONE)[9] = PR::ONE;
    static constexpr const int32_t (&K_IN)[9] = PR::K_IN;
    static constexpr const int32_t (&K_OUT)[9] = PR::K_OUT;

    // Modulus limb k as a LITERAL.  P[k] read through the constexpr table is a load on the device (hipcc emits
    // constexpr class statics as externally initialised constant memory and does not fold reads of them): the
    // bucket kernels then re-load the nine limbs with three vector loads inside every loop iteration, and the
    // in-order vmcnt wait for them also waits for the prefetch of the next point issued just before.  The
    // local constexpr copies below are folded by the front end.
    template <const int32_t (&A)[9]>
    ZK_HD static constexpr int32_t LIT(int k) {
        constexpr int32_t c0 = A[0], c1 = A[1], c2 = A[2], c3 = A[3], c4 = A[4], c5 = A[5], c6 = A[6], c7 = A[7], c8 = A[8];
        return k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : k == 3 ? c3 : k == 4 ? c4 : k == 5 ? c5 : k == 6 ? c6 : k == 7 ? c7 : c8;
    }
    ZK_HD static constexpr int32_t PL(int k) { return LIT<PR::P>(k); }
    // Modulus limb for the reduction MADs, pinned in an SGPR on the device through an (empty, CSE-able) asm:
    // left to itself the compiler parks p[1..8] in 8 VGPRs.
    ZK_HD static int32_t PS(int k) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_SGPR_MODULUS)
        int32_t r;
        asm("" : "=s"(r) : "0"(PL(k)));
        return r;
#else
        return PL(k);
#endif
    }
    ZK_HD static Fq29 zero() {
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = 0;
        return r;
    }
    ZK_HD static Fq29 one() {
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = LIT<PR::ONE>(i);
        return r;
    }
    ZK_HD static Fq29 k_in() {
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = LIT<PR::K_IN>(i);
        return r;
    }
    ZK_HD static Fq29 k_out() {
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = LIT<PR::K_OUT>(i);
        return r;
    }
    ZK_HD bool is_zero_raw() const {
        int32_t o = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) o |= l[i];
        return o == 0;
    }

    // one-step parallel carry: limbs 0..7 back into [-8, 2^29 + 8) for inputs |t_i| < 2^31
    ZK_HD static Fq29 carry(const Fq29 &t) {
        Fq29 r;
        r.l[0] = t.l[0] & MASK;
#pragma unroll
        for (int i = 1; i < 8; i++) r.l[i] = (t.l[i] & MASK) + (t.l[i - 1] >> 29);
        r.l[8] = t.l[8] + (t.l[7] >> 29);
        return r;
    }
    // full sequential carry: limbs 0..7 in [0, 2^29), limb 8 carries the sign
    ZK_HD static Fq29 carry_full(const Fq29 &t) {
        Fq29 r = t;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            r.l[i + 1] += r.l[i] >> 29;
            r.l[i] &= MASK;
        }
        return r;
    }
    ZK_HD static Fq29 add(const Fq29 &a, const Fq29 &b) {
        Fq29 t;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = a.l[i] + b.l[i];
        return carry(t);
    }
    ZK_HD static Fq29 sub(const Fq29 &a, const Fq29 &b) {
        Fq29 t;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = a.l[i] - b.l[i];
        return carry(t);
    }
    ZK_HD static Fq29 neg(const Fq29 &a) {
        Fq29 t;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = -a.l[i];
        return carry(t);
    }
    ZK_HD static Fq29 dbl(const Fq29 &a) {
        Fq29 t;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = a.l[i] << 1;
        return carry(t);
    }

    // ---- Montgomery products a*b*2^-261 (mod p), product-scanning over 17 columns of v_mad_i64_i32.
    // A column is a CHAIN of MADs into one 64-bit accumulator; the carry of column k (acc >> 29) is the addend of column
    // k+1's first MAD, so there is no 64-bit add per column.  Budget of a column (2^63 = 32 * 2^58): the reduction terms take
    // 2^31 * (sum of the modulus limbs) = 12.05 * 2^58 (Fq) / 13.6 * 2^58 (Fr) now that the lower eight reduction digits are
    // unmasked (mont_m), so the OPERAND terms of a job must stay within 18 * 2^58 of one sign — 18 products of tight limbs, or
    // 9 of tight x wide (one lazily added operand), see the job shapes below; Fr with a lazily added operand has ~1 % left.
    // tools/field29_bounds_test.cpp (-DZK_CHECK_COLUMNS: every MAD checked in 128 bits) runs each job shape the kernels use
    // over extreme and random operands of the callers' limb ranges.  A "job" lists the operand products of a column; the engine below runs
    //   run2: TWO independent jobs with their chains interleaved MAD by MAD — the dependent-issue latency of one
    //         chain is covered by the other (what hipcc's split-and-merge bought, without its 17 merge adds), and no
    //         result is consumed by the very next instruction (hipcc puts an s_nop behind an inline asm whose
    //         result the next instruction reads);
    //   run1: one job, its operand products and its reduction products as two interleaved chains merged by one
    //         64-bit add per column (a product that has no independent sibling).
    // Per product: 162 MADs + 17 shifts + 9 v_mul_lo + 10 masks (+ 17 adds in run1).
    // masked limb of a result.  C column sums only: the mask tells the optimiser the limb is non-negative (see from_words)
    ZK_HD static int32_t out_limb(int64_t acc) {
        int32_t v = (int32_t)((uint32_t)acc & (uint32_t)MASK);
        ZK_SIGN_BARRIER(v);
        return v;
    }
    // The reduction digit of a column: any m = acc * (-p^-1) (mod 2^29) clears the column's low 29 bits.  Only the TOP digit
    // (column 8) is masked to 29 bits; the lower ones stay the signed 32-bit product as it comes out of v_mul_lo_u32: their three
    // extra bits act as a carry into the next digit (which is computed from the running sum and absorbs it), the value of
    // M = sum m_i 2^(29 i) — and with it the output range — is set by the masked top digit alone, and eight v_and per product go
    // away.  Column bound with |m_i| < 2^31: the reduction terms add up to at most 2^31 * (sum of the modulus limbs) = 3.47e18 (Fq) /
    // 3.92e18 (Fr) beside at most 18 operand terms of (2^29 + 16)^2, or 9 of (2^29 + 16)(2^30 + 32), = 5.19e18: below 2^63 = 9.22e18.
    ZK_HD static int32_t mont_m(int64_t acc, bool top) {
        const uint32_t m = (uint32_t)acc * N0INV;
        return top ? (int32_t)(m & (uint32_t)MASK) : (int32_t)m;
    }
    ZK_HD static constexpr int col_lo(int k) { return k < 9 ? 0 : k - 8; }
    ZK_HD static constexpr int col_n(int k) { return k < 9 ? k + 1 : 17 - k; }          // a_i * b_(k-i) terms of column k
    struct JMul {                       // a * b
        const Fp29 &a, &b;
        ZK_HD static constexpr int n(int k) { return col_n(k); }
        ZK_HD void term(int64_t &acc, int k, int t, bool first) const {
            const int i = col_lo(k) + t;
            acc = first ? zk_mad0(a.l[i], b.l[k - i]) : zk_mad(acc, a.l[i], b.l[k - i]);
        }
        ZK_HD void ops(int k, int t, int32_t &x, int32_t &y) const {
            const int i = col_lo(k) + t;
            x = a.l[i];
            y = b.l[k - i];
        }
    };
    struct JSqr {                       // a * a: symmetric terms once with a doubled operand — 45 product MADs instead of 81
        const Fp29 &a;
        int32_t a2[9];
        ZK_HD explicit JSqr(const Fp29 &x) : a(x) {
#pragma unroll
            for (int i = 0; i < 9; i++) a2[i] = x.l[i] << 1;
        }
        ZK_HD static constexpr int nsym(int k) { return (k + 1) / 2 - col_lo(k); }       // i in [lo, k/2)
        ZK_HD static constexpr int n(int k) { return nsym(k) + ((k & 1) == 0 ? 1 : 0); }
        ZK_HD void term(int64_t &acc, int k, int t, bool first) const {
            if (t < nsym(k)) {
                const int i = col_lo(k) + t;
                acc = first ? zk_mad0(a2[i], a.l[k - i]) : zk_mad(acc, a2[i], a.l[k - i]);
            } else {
                acc = first ? zk_mad0(a.l[k >> 1], a.l[k >> 1]) : zk_mad(acc, a.l[k >> 1], a.l[k >> 1]);
            }
        }
        ZK_HD void ops(int k, int t, int32_t &x, int32_t &y) const {
            if (t < nsym(k)) {
                const int i = col_lo(k) + t;
                x = a2[i];
                y = a.l[k - i];
            } else {
                x = y = a.l[k >> 1];
            }
        }
    };
    struct JMulAdd2 {                   // a * b + c * d with ONE reduction
        const Fp29 &a, &b, &c, &d;
        ZK_HD static constexpr int n(int k) { return 2 * col_n(k); }
        ZK_HD void term(int64_t &acc, int k, int t, bool first) const {
            const int i = col_lo(k) + (t >> 1);
            if (t & 1) acc = zk_mad(acc, c.l[i], d.l[k - i]);
            else acc = first ? zk_mad0(a.l[i], b.l[k - i]) : zk_mad(acc, a.l[i], b.l[k - i]);
        }
        ZK_HD void ops(int k, int t, int32_t &x, int32_t &y) const {
            const int i = col_lo(k) + (t >> 1);
            x = (t & 1) ? c.l[i] : a.l[i];
            y = (t & 1) ? d.l[k - i] : b.l[k - i];
        }
    };
    struct JMulAdd4 {                   // a * b + c * d + e * f + g * h with ONE reduction (the G2 lane pair's Y3 = R*D - Y1*PPP)
        // Column bound: 36 operand terms.  Callers keep every limb non-negative up to the carry slack ([-8, 2^29 + 8]) and
        // arrange the signs so that two of the four products enter negated: at most 18 terms of one sign (18 * 2^58) plus the
        // reduction terms (2^31 * sum of the modulus limbs = 12.05 * 2^58 for Fq, the only field this job runs in) stay below
        // 2^63 = 32 * 2^58.
        const Fp29 &a, &b, &c, &d, &e, &f, &g, &h;
        ZK_HD static constexpr int n(int k) { return 4 * col_n(k); }
        ZK_HD void term(int64_t &acc, int k, int t, bool first) const {
            int32_t x, y;
            ops(k, t, x, y);
            acc = first ? zk_mad0(x, y) : zk_mad(acc, x, y);
        }
        ZK_HD void ops(int k, int t, int32_t &x, int32_t &y) const {
            const int i = col_lo(k) + (t >> 2), s = t & 3;
            x = s == 0 ? a.l[i] : s == 1 ? c.l[i] : s == 2 ? e.l[i] : g.l[i];
            y = s == 0 ? b.l[k - i] : s == 1 ? d.l[k - i] : s == 2 ? f.l[k - i] : h.l[k - i];
        }
    };
#if defined(ZK_BLOCK_MAD)
    // Block form: the MADs of a column go out as a few multi-instruction asm statements (mad_blocks.inc) in which the
    // two chains alternate — one s_nop per block instead of one per MAD.  Columns are template instances (block sizes
    // must be constants where the statement is chosen).
    template <int OFF, int N>
    __device__ __forceinline__ static void blk_dual(int64_t &acc0, int64_t &acc1, const int32_t *x0, const int32_t *y0, const int32_t *x1, const int32_t *y1) {
        if constexpr (N > 0) {
            zk_blk_dvv(N < 6 ? N : 6, acc0, acc1, x0 + OFF, y0 + OFF, x1 + OFF, y1 + OFF);
            blk_dual<OFF + 6, N - 6>(acc0, acc1, x0, y0, x1, y1);
        }
    }
    template <int OFF, int N>
    __device__ __forceinline__ static void blk_single(int64_t &acc, const int32_t *x, const int32_t *y) {
        if constexpr (N > 0) {
            zk_blk_svv(N < 9 ? N : 9, acc, x + OFF, y + OFF);
            blk_single<OFF + 9, N - 9>(acc, x, y);
        }
    }
    template <int K, class J0, class J1>
    __device__ __forceinline__ static void col2(int64_t &acc0, int64_t &acc1, int32_t (&m0)[9], int32_t (&m1)[9], Fp29 &r0, const J0 &j0, Fp29 &r1,
                                                const J1 &j1) {
        constexpr int n0 = J0::n(K), n1 = J1::n(K), nmin = n0 < n1 ? n0 : n1;
        int32_t x0[36], y0[36], x1[36], y1[36];
#pragma unroll
        for (int t = 0; t < 36; t++) {
            if (t < n0) j0.ops(K, t, x0[t], y0[t]);
            if (t < n1) j1.ops(K, t, x1[t], y1[t]);
        }
        blk_dual<0, nmin>(acc0, acc1, x0, y0, x1, y1);                 // both chains alternating, six MADs of each per statement
        blk_single<nmin, n0 - nmin>(acc0, x0, y0);                     // what the longer job has left: one chain, nine per statement
        blk_single<nmin, n1 - nmin>(acc1, x1, y1);
        constexpr int rlo = K < 9 ? 0 : K - 8, rhi = K < 9 ? K : 9, nr = rhi - rlo;
        int32_t pr[9];
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (i < nr) pr[i] = PS(K - (rlo + i));
        if constexpr (nr > 0) zk_blk_dmp(nr < 8 ? nr : 8, acc0, acc1, m0 + rlo, m1 + rlo, pr);
        if constexpr (nr > 8) zk_blk_dmp(nr - 8, acc0, acc1, m0 + rlo + 8, m1 + rlo + 8, pr + 8);
        if constexpr (K < 9) {
            m0[K] = mont_m(acc0, K == 8);
            m1[K] = mont_m(acc1, K == 8);
            zk_blk_dmp1(acc0, acc1, m0[K], m1[K], PS(0));
        } else {
            r0.l[K - 9] = out_limb(acc0);
            r1.l[K - 9] = out_limb(acc1);
        }
        acc0 >>= 29;
        acc1 >>= 29;
        if constexpr (K < 16) col2<K + 1>(acc0, acc1, m0, m1, r0, j0, r1, j1);
    }
    template <class J0, class J1>
    __device__ __forceinline__ static void run2(Fp29 &r0, const J0 &j0, Fp29 &r1, const J1 &j1) {
        int64_t acc0 = 0, acc1 = 0;
        int32_t m0[9], m1[9];
        col2<0>(acc0, acc1, m0, m1, r0, j0, r1, j1);
        r0.l[8] = (int32_t)acc0;
        r1.l[8] = (int32_t)acc1;
    }
    // Three jobs at once, for a long job J2 with as many operand terms per column as J0 and J1 together (the G2 lane
    // pair's  zz' | zzz' | Y3 = R*D - Y1*PPP): J2's chain alternates first with J0's MADs, then with J1's, so no chain runs
    // alone for long (a column of J2 beside J0 only — 18 dependent MADs at its end — measured 2 % slower per addition).
    template <int K, class J0, class J1, class J2>
    __device__ __forceinline__ static void col3(int64_t &acc0, int64_t &acc1, int64_t &acc2, int32_t (&m0)[9], int32_t (&m1)[9], int32_t (&m2)[9],
                                                Fp29 &r0, const J0 &j0, Fp29 &r1, const J1 &j1, Fp29 &r2, const J2 &j2) {
        constexpr int n0 = J0::n(K), n1 = J1::n(K), n2 = J2::n(K);
        static_assert(n2 == n0 + n1, "run3: the long job carries as many terms as the two short ones");
        int32_t x0[18], y0[18], x1[18], y1[18], x2[36], y2[36];
#pragma unroll
        for (int t = 0; t < 36; t++) {
            if (t < n0) j0.ops(K, t, x0[t], y0[t]);
            if (t < n1) j1.ops(K, t, x1[t], y1[t]);
            if (t < n2) j2.ops(K, t, x2[t], y2[t]);
        }
        blk_dual<0, n0>(acc0, acc2, x0, y0, x2, y2);
        blk_dual<0, n1>(acc1, acc2, x1, y1, x2 + n0, y2 + n0);
        constexpr int rlo = K < 9 ? 0 : K - 8, rhi = K < 9 ? K : 9, nr = rhi - rlo;
        int32_t pr[9];
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (i < nr) pr[i] = PS(K - (rlo + i));
        if constexpr (nr > 0) zk_blk_tmp(nr < 6 ? nr : 6, acc0, acc1, acc2, m0 + rlo, m1 + rlo, m2 + rlo, pr);
        if constexpr (nr > 6) zk_blk_tmp(nr - 6, acc0, acc1, acc2, m0 + rlo + 6, m1 + rlo + 6, m2 + rlo + 6, pr + 6);
        if constexpr (K < 9) {
            m0[K] = mont_m(acc0, K == 8);
            m1[K] = mont_m(acc1, K == 8);
            m2[K] = mont_m(acc2, K == 8);
            zk_blk_tmp1(acc0, acc1, acc2, m0[K], m1[K], m2[K], PS(0));
        } else {
            r0.l[K - 9] = out_limb(acc0);
            r1.l[K - 9] = out_limb(acc1);
            r2.l[K - 9] = out_limb(acc2);
        }
        acc0 >>= 29;
        acc1 >>= 29;
        acc2 >>= 29;
        if constexpr (K < 16) col3<K + 1>(acc0, acc1, acc2, m0, m1, m2, r0, j0, r1, j1, r2, j2);
    }
    template <class J0, class J1, class J2>
    __device__ __forceinline__ static void run3(Fp29 &r0, const J0 &j0, Fp29 &r1, const J1 &j1, Fp29 &r2, const J2 &j2) {
        int64_t acc0 = 0, acc1 = 0, acc2 = 0;
        int32_t m0[9], m1[9], m2[9];
        col3<0>(acc0, acc1, acc2, m0, m1, m2, r0, j0, r1, j1, r2, j2);
        r0.l[8] = (int32_t)acc0;
        r1.l[8] = (int32_t)acc1;
        r2.l[8] = (int32_t)acc2;
    }
#else
    template <class J0, class J1>
    ZK_HD static void run2(Fp29 &r0, const J0 &j0, Fp29 &r1, const J1 &j1) {
        int64_t acc0 = 0, acc1 = 0;
        int32_t m0[9], m1[9];
#pragma unroll
        for (int k = 0; k < 17; k++) {
            const int n0 = J0::n(k), n1 = J1::n(k), nmax = n0 > n1 ? n0 : n1;
#pragma unroll
            for (int t = 0; t < nmax; t++) {
                if (t < n0) j0.term(acc0, k, t, k == 0 && t == 0);
                if (t < n1) j1.term(acc1, k, t, k == 0 && t == 0);
            }
#pragma unroll
            for (int i = (k < 9 ? 0 : k - 8); i < (k < 9 ? k : 9); i++) {
                acc0 = zk_mad_s(acc0, m0[i], PS(k - i));
                acc1 = zk_mad_s(acc1, m1[i], PS(k - i));
            }
            if (k < 9) {
                m0[k] = mont_m(acc0, k == 8);
                m1[k] = mont_m(acc1, k == 8);
                acc0 = zk_mad_s(acc0, m0[k], PS(0));
                acc1 = zk_mad_s(acc1, m1[k], PS(0));
            } else {
                r0.l[k - 9] = out_limb(acc0);
                r1.l[k - 9] = out_limb(acc1);
            }
            acc0 >>= 29;         // k < 9: exact, the low 29 bits are zero now
            acc1 >>= 29;
        }
        r0.l[8] = (int32_t)acc0;
        r1.l[8] = (int32_t)acc1;
    }
    template <class J0, class J1, class J2>
    ZK_HD static void run3(Fp29 &r0, const J0 &j0, Fp29 &r1, const J1 &j1, Fp29 &r2, const J2 &j2) {     // (host pass / C builds)
        Fp29 t2 = run1(j2);
        run2(r0, j0, r1, j1);
        r2 = t2;
    }
#endif
    template <class J>
    ZK_HD static Fp29 run1(const J &j) {
        int64_t acc = 0;
        int32_t m[9];
        Fp29 r;
#pragma unroll
        for (int k = 0; k < 17; k++) {
#if defined(ZK_ASM_MAD) && !defined(ZK_SERIAL_CHAIN)
            int64_t accp = 0;                              // operand products from zero; `acc` takes the reduction products
            const int np = J::n(k), nr = (k < 9 ? k : 17 - k), nmax = np > nr ? np : nr;
#pragma unroll
            for (int t = 0; t < nmax; t++) {
                if (t < np) j.term(accp, k, t, t == 0);
                if (t < nr) {
                    const int i = (k < 9 ? 0 : k - 8) + t;
                    acc = zk_mad_s(acc, m[i], PS(k - i));
                }
            }
            acc += accp;
#else
#pragma unroll
            for (int t = 0; t < J::n(k); t++) j.term(acc, k, t, k == 0 && t == 0);
#pragma unroll
            for (int i = (k < 9 ? 0 : k - 8); i < (k < 9 ? k : 9); i++) acc = zk_mad_s(acc, m[i], PS(k - i));
#endif
            if (k < 9) {
                m[k] = mont_m(acc, k == 8);
                acc = zk_mad_s(acc, m[k], PS(0));
            } else {
                r.l[k - 9] = out_limb(acc);
            }
            acc >>= 29;
        }
        r.l[8] = (int32_t)acc;
        return r;
    }
    ZK_HD static Fq29 mul(const Fq29 &a, const Fq29 &b) { return run1(JMul{a, b}); }
    ZK_HD static Fq29 sqr(const Fq29 &a) { return run1(JSqr(a)); }
    // (a*b + c*d) * 2^-261 with ONE reduction: both products share the column accumulators.  The Fq2 product is two
    // of these — same MAD count as Karatsuba but without its five add/sub + carry passes.
    static constexpr bool FUSED_MULADD = true;
    ZK_HD static Fq29 mul_add2(const Fq29 &a, const Fq29 &b, const Fq29 &c, const Fq29 &d) { return run1(JMulAdd2{a, b, c, d}); }
    // pairs of independent products (the mixed additions of curve29.hpp are written in these)
    ZK_HD static void mul2(Fq29 &r0, const Fq29 &a0, const Fq29 &b0, Fq29 &r1, const Fq29 &a1, const Fq29 &b1) {
        run2(r0, JMul{a0, b0}, r1, JMul{a1, b1});
    }
    ZK_HD static void sqr2(Fq29 &r0, const Fq29 &a0, Fq29 &r1, const Fq29 &a1) { run2(r0, JSqr(a0), r1, JSqr(a1)); }
    // limb-wise add / sub WITHOUT the carry pass.  Bound bookkeeping (see curve29.hpp): a product
    // tolerates |limb| <= 2^29+16 on both operands, or 2^30+32 on ONE of them; mul_add2 needs
    // <= 2^29+16 on all four.  The difference of two values with non-negative limbs (e.g. two
    // product outputs) already satisfies the tight bound.
    ZK_HD static Fq29 add_nc(const Fq29 &a, const Fq29 &b) {
        Fq29 t;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = a.l[i] + b.l[i];
        return t;
    }
    ZK_HD static Fq29 sub_nc(const Fq29 &a, const Fq29 &b) {
        Fq29 t;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = a.l[i] - b.l[i];
        return t;
    }
    // limb-wise negation / doubling WITHOUT the carry pass: valid as one operand of a product
    ZK_HD static Fq29 neg_lazy(const Fq29 &a) {
        Fq29 t;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = -a.l[i];
        return t;
    }
    ZK_HD static Fq29 dbl_lazy(const Fq29 &a) {
        Fq29 t;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = a.l[i] << 1;
        return t;
    }

    // value - k*p with k = round(value / p) estimated from the two top limbs: result in (-p, p)
    ZK_HD static Fq29 reduce_near_zero(const Fq29 &a) {
        // p >> 203 = 1702635872462388 for both BN254 primes (they share their top 128 bits);
        // (l8, l7) = value >> 203 up to the low limbs' slack
        float vt = (float)a.l[8] * 536870912.0f + (float)a.l[7];
        int32_t k = (int32_t)__builtin_rintf(vt * (1.0f / 1702635872462388.0f));
        Fq29 t;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = 0;
        // t = a - k*p, column-wise with 64-bit intermediates (|k| <= 16)
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (int64_t)a.l[i] - (int64_t)k * PL(i);
            t.l[i] = (int32_t)((uint32_t)c & (uint32_t)MASK);
            c >>= 29;
        }
        t.l[8] = (int32_t)(c + a.l[8] - (int64_t)k * PL(8));
        return t;    // limbs 0..7 in [0, 2^29)
    }
    // exact: value == 0 (mod p).  After reduce_near_zero the value is in (-p, p): zero iff all limbs are.
    // Four-instruction filter in front of it: value = k*p needs l[0] = k*p[0] (mod 2^29) — the lowest limb takes no carry
    // from anywhere, lazy or not — i.e. l[0] * p[0]^-1 = k, a SMALL number (|k| <= 32 for values in (-32p, 32p)).  A
    // non-zero value passes with probability 65 / 2^29; the exact test behind it then runs for (almost) no lane.
    ZK_HD bool may_be_zero() const {
        const uint32_t k = (uint32_t)l[0] * ((0u - N0INV) & (uint32_t)MASK);       // p[0]^-1 = -N0INV (mod 2^29)
        return ((k + 32u) & (uint32_t)MASK) < 65u;
    }
#if defined(ZK_NO_ZERO_FILTER)
    ZK_HD bool is_zero() const { return reduce_near_zero(*this).is_zero_raw(); }
#else
    ZK_HD bool is_zero() const { return may_be_zero() && reduce_near_zero(*this).is_zero_raw(); }
#endif

    // canonical representative in [0, p), limbs 0..8 all non-negative
    ZK_HD static Fq29 canonical(const Fq29 &a) {
        Fq29 t = reduce_near_zero(a);              // (-p, p), limbs 0..7 normalised
        if (t.l[8] < 0) {                          // negative: add p
            Fq29 u;
#pragma unroll
            for (int i = 0; i < 9; i++) u.l[i] = t.l[i] + PL(i);
            t = carry_full(u);
        }
        return t;
    }

    // ---- 256-bit words <-> limbs (value must be canonical for to_words)
    ZK_HD static Fq29 from_words(const u32 w[8]) {
        Fq29 r;
        r.l[0] = (int32_t)(w[0] & (u32)MASK);
        r.l[1] = (int32_t)(((w[0] >> 29) | (w[1] << 3)) & (u32)MASK);
        r.l[2] = (int32_t)(((w[1] >> 26) | (w[2] << 6)) & (u32)MASK);
        r.l[3] = (int32_t)(((w[2] >> 23) | (w[3] << 9)) & (u32)MASK);
        r.l[4] = (int32_t)(((w[3] >> 20) | (w[4] << 12)) & (u32)MASK);
        r.l[5] = (int32_t)(((w[4] >> 17) | (w[5] << 15)) & (u32)MASK);
        r.l[6] = (int32_t)(((w[5] >> 14) | (w[6] << 18)) & (u32)MASK);
        r.l[7] = (int32_t)(((w[6] >> 11) | (w[7] << 21)) & (u32)MASK);
        r.l[8] = (int32_t)(w[7] >> 8);
        // hide "this limb is non-negative" from the optimiser (see zk_mad above)
#pragma unroll
        for (int i = 0; i < 9; i++) ZK_SIGN_BARRIER(r.l[i]);
        return r;
    }
    ZK_HD static void to_words(u32 w[8], const Fq29 &c) {
        const u32 l0 = (u32)c.l[0], l1 = (u32)c.l[1], l2 = (u32)c.l[2], l3 = (u32)c.l[3], l4 = (u32)c.l[4];
        const u32 l5 = (u32)c.l[5], l6 = (u32)c.l[6], l7 = (u32)c.l[7], l8 = (u32)c.l[8];
        w[0] = l0 | (l1 << 29);
        w[1] = (l1 >> 3) | (l2 << 26);
        w[2] = (l2 >> 6) | (l3 << 23);
        w[3] = (l3 >> 9) | (l4 << 20);
        w[4] = (l4 >> 12) | (l5 << 17);
        w[5] = (l5 >> 15) | (l6 << 14);
        w[6] = (l6 >> 18) | (l7 << 11);
        w[7] = (l7 >> 21) | (l8 << 8);
    }
    // zkey form x*2^256 (canonical words)  <->  internal x*2^261
    ZK_HD static Fq29 from_mont256(const Words &x) { return mul(from_words(x.v), k_in()); }
    ZK_HD static Words to_mont256(const Fq29 &x) {
        Words r;
        to_words(r.v, canonical(mul(x, k_out())));
        return r;
    }
    // internal value <-> canonical words of the SAME Montgomery form (HBM residency format)
    ZK_HD static Fq29 load(const Words &x) { return from_words(x.v); }
    ZK_HD static Words store(const Fq29 &x) {
        Words r;
        to_words(r.v, canonical(x));
        return r;
    }
    // a^(p-2) (Fermat); table building only.  Exponent bits come from the 29-bit limbs of p.
    ZK_HD static Fq29 inv(const Fq29 &a) {
        Fq29 result = one(), base = a;
        for (int i = 0; i < 9 * 29; i++) {
            int32_t limb = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) limb = (i / 29 == k) ? PL(k) : limb;
            if (i < 29) limb -= 2;                      // p - 2 (P[0] >= 2)
            if ((limb >> (i % 29)) & 1) result = mul(result, base);
            base = sqr(base);
        }
        return result;
    }
    // value * 2^-261: leaves Montgomery form (standard-form result, e.g. MSM scalars)
    ZK_HD static Fq29 from_mont(const Fq29 &a) {
        Fq29 o = zero();
        o.l[0] = 1;
        return mul(a, o);
    }
};

typedef Fp29<Fq29Params> Fq29;
typedef Fp29<Fr29Params> Fr29;

}   // namespace zk
