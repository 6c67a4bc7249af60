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

    // Modulus limb for the reduction MADs.  On the device the value is pinned in an SGPR through an (empty,
    // CSE-able) asm: left to itself the compiler parks p[1..8] in 8 VGPRs that it re-loads from a constant
    // table inside every loop iteration of the bucket kernels (two global_load_dwordx4 + a wait per point).
    ZK_HD static int32_t PS(int k) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_SGPR_MODULUS)
        int32_t r;
        asm("" : "=s"(r) : "0"(__builtin_amdgcn_readfirstlane(P[k])));
        return r;
#else
        return P[k];
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
        for (int i = 0; i < 9; i++) r.l[i] = ONE[i];
        return r;
    }
    ZK_HD static Fq29 k_in() {
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = K_IN[i];
        return r;
    }
    ZK_HD static Fq29 k_out() {
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = K_OUT[i];
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

    // Montgomery product a*b*2^-261 (mod p), product-scanning: every column is a chain of
    // v_mad_i64_i32 into one 64-bit accumulator; |column| < 18 * 2^58.1 < 2^63.
    ZK_HD static Fq29 mul(const Fq29 &a, const Fq29 &b) {
        int64_t acc = 0;
        int32_t m[9];
        Fq29 r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (int64_t)m[i] * PS(k - i);
            m[k] = (int32_t)(((uint32_t)acc * N0INV) & (uint32_t)MASK);
            acc += (int64_t)m[k] * PS(0);
            acc >>= 29;          // exact: the low 29 bits are zero now
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (int64_t)m[i] * PS(k - i);
            r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)MASK);
            acc >>= 29;
        }
        r.l[8] = (int32_t)acc;
        return r;
    }
    // a^2 * 2^-261: symmetric terms once with a doubled operand — 45 product MADs instead of 81
    ZK_HD static Fq29 sqr(const Fq29 &a) {
        int32_t a2[9];
#pragma unroll
        for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
        int64_t acc = 0;
        int32_t m[9];
        Fq29 r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; 2 * i < k; i++) acc += (int64_t)a2[i] * a.l[k - i];
            if ((k & 1) == 0) acc += (int64_t)a.l[k >> 1] * a.l[k >> 1];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (int64_t)m[i] * PS(k - i);
            m[k] = (int32_t)(((uint32_t)acc * N0INV) & (uint32_t)MASK);
            acc += (int64_t)m[k] * PS(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; 2 * i < k; i++) acc += (int64_t)a2[i] * a.l[k - i];
            if ((k & 1) == 0) acc += (int64_t)a.l[k >> 1] * a.l[k >> 1];
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (int64_t)m[i] * PS(k - i);
            r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)MASK);
            acc >>= 29;
        }
        r.l[8] = (int32_t)acc;
        return r;
    }
    // (a*b + c*d) * 2^-261 with ONE reduction: both products share the column accumulators
    // (|column| < 27 * 2^58 < 2^63).  The Fq2 product is two of these — same MAD count as
    // Karatsuba but without its five add/sub + carry passes.
    static constexpr bool FUSED_MULADD = true;
    ZK_HD static Fq29 mul_add2(const Fq29 &a, const Fq29 &b, const Fq29 &c, const Fq29 &d) {
        int64_t acc = 0;
        int32_t m[9];
        Fq29 r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) {
                acc += (int64_t)a.l[i] * b.l[k - i];
                acc += (int64_t)c.l[i] * d.l[k - i];
            }
#pragma unroll
            for (int i = 0; i < k; i++) acc += (int64_t)m[i] * PS(k - i);
            m[k] = (int32_t)(((uint32_t)acc * N0INV) & (uint32_t)MASK);
            acc += (int64_t)m[k] * PS(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) {
                acc += (int64_t)a.l[i] * b.l[k - i];
                acc += (int64_t)c.l[i] * d.l[k - i];
            }
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (int64_t)m[i] * PS(k - i);
            r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)MASK);
            acc >>= 29;
        }
        r.l[8] = (int32_t)acc;
        return r;
    }
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
            c += (int64_t)a.l[i] - (int64_t)k * P[i];
            t.l[i] = (int32_t)((uint32_t)c & (uint32_t)MASK);
            c >>= 29;
        }
        t.l[8] = (int32_t)(c + a.l[8] - (int64_t)k * P[8]);
        return t;    // limbs 0..7 in [0, 2^29)
    }
    // exact: value == 0 (mod p).  After reduce_near_zero the value is in (-p, p): zero iff all limbs are.
    ZK_HD bool is_zero() const { return reduce_near_zero(*this).is_zero_raw(); }

    // canonical representative in [0, p), limbs 0..8 all non-negative
    ZK_HD static Fq29 canonical(const Fq29 &a) {
        Fq29 t = reduce_near_zero(a);              // (-p, p), limbs 0..7 normalised
        if (t.l[8] < 0) {                          // negative: add p
            Fq29 u;
#pragma unroll
            for (int i = 0; i < 9; i++) u.l[i] = t.l[i] + P[i];
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
            for (int k = 0; k < 9; k++) limb = (i / 29 == k) ? P[k] : limb;
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
