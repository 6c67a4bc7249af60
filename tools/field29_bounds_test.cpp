// Column-overflow check of field29.hpp's Montgomery products (host build, g++ -DZK_CHECK_COLUMNS).
// Every v_mad_i64_i32 of the kernels is `acc + a*b` in ONE signed 64-bit register; with the lower eight reduction digits
// unmasked the budget is 18 * 2^58 of operand terms (field29.hpp).  Here each MAD is evaluated in 128 bits and a column that
// leaves int64 is counted.  Shapes covered — every job the kernels run, with the limb ranges their callers produce:
//   JMul tight x tight, JMul tight x wide (one lazily added / doubled operand: DIT butterflies, f2_sqr), JSqr of a tight value,
//   JMulAdd2 of four tight operands, JMulAdd4 with two products of each sign (G2 lane pair's Y3), in Fq and (but JMulAdd4) Fr;
//   plus chains of G1 and G2 mixed additions (curve29.hpp) over random operands, which exercise sub_nc / neg_lazy / dbl_lazy as used.
// Prints the peak |column| as a multiple of 2^58 and exits 1 on any overflow.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "curve29.hpp"

using namespace zk;

static std::mt19937_64 rng(20260929);
static const int32_t TIGHT = (1 << 29) + 16, WIDE = (1 << 30) + 32;

// sign: +1 limbs in [0, bound], -1 limbs in [-bound, 0], 0 mixed signs; extreme: every limb at the bound
template <class F>
static F limbs(int32_t bound, int sign, bool extreme) {
    F r;
    for (int i = 0; i < 9; i++) {
        int32_t v = extreme ? bound : (int32_t)(rng() % ((uint64_t)bound + 1));
        if (sign < 0 || (sign == 0 && (rng() & 1))) v = -v;
        r.l[i] = v;
    }
    // limb 8 carries the value's magnitude: at most 16p -> |l8| < 16 * 2^22; keep it within the tight bound anyway
    return r;
}

template <class F>
static void job_shapes(const char *name, bool with_add4) {
    for (int round = 0; round < 20000; round++) {
        const bool ext = round < 64;
        const int sa = ext ? ((round & 1) ? 1 : -1) : 0, sb = ext ? ((round & 2) ? 1 : -1) : 0;
        F a = limbs<F>(TIGHT, sa, ext), b = limbs<F>(TIGHT, sb, ext), w = limbs<F>(WIDE, sb, ext);
        F c = limbs<F>(TIGHT, ext ? sa : 0, ext), d = limbs<F>(TIGHT, ext ? sb : 0, ext);
        (void)F::mul(a, b);
        (void)F::mul(a, w);                     // one lazily added operand
        (void)F::sqr(a);
        (void)F::mul_add2(a, b, c, d);          // extreme rounds: both products of one sign
        F r0, r1;
        F::mul2(r0, a, b, r1, c, w);
        F::sqr2(r0, a, r1, c);
        if (with_add4) {
            // two products >= 0 and two <= 0 (the caller's arrangement): non-negative operands, two of them negated
            F e = limbs<F>(TIGHT, 1, ext), f = limbs<F>(TIGHT, 1, ext), g = limbs<F>(TIGHT, 1, ext), h = limbs<F>(TIGHT, 1, ext);
            F p = limbs<F>(TIGHT, 1, ext), q = limbs<F>(TIGHT, 1, ext), u = limbs<F>(TIGHT, 1, ext), v = limbs<F>(TIGHT, 1, ext);
            (void)F::run1(typename F::JMulAdd4{e, f, g, h, F::neg_lazy(p), q, F::neg_lazy(u), v});
        }
    }
    printf("%-4s job shapes : peak |column| = %.3f * 2^58, overflows %ld\n", name, (double)zk_column_peak() / 288230376151711744.0, zk_column_overflows());
}

template <class F>
static F random_canonical() {
    u32 w[8];
    for (int i = 0; i < 8; i++) w[i] = (u32)rng();
    w[7] &= 0x1fffffffu;                        // < 2^253 < p
    return F::from_words(w);
}

int main() {
    job_shapes<Fq29>("Fq", true);
    const long fq_over = zk_column_overflows();
    zk_column_peak() = 0;
    job_shapes<Fr29>("Fr", false);
    zk_column_peak() = 0;
    // chains of mixed additions over random field elements (no curve equation needed for bounds: the formulas see the same ranges)
    XYZZ<Fq29> acc = XYZZ<Fq29>::inf();
    for (int i = 0; i < 20000; i++) {
        Affine<Fq29> p{random_canonical<Fq29>(), random_canonical<Fq29>()};
        if (i & 1) negate_y(p);
        madd(acc, p);
    }
    XYZZ<Fq2r> acc2 = XYZZ<Fq2r>::inf();
    for (int i = 0; i < 10000; i++) {
        Affine<Fq2r> p{Fq2r{random_canonical<Fq29>(), random_canonical<Fq29>()}, Fq2r{random_canonical<Fq29>(), random_canonical<Fq29>()}};
        if (i & 1) negate_y(p);
        madd(acc2, p);
    }
    printf("G1 / G2 mixed-addition chains: peak |column| = %.3f * 2^58, overflows %ld\n", (double)zk_column_peak() / 288230376151711744.0,
           zk_column_overflows());
    (void)fq_over;
    if (zk_column_overflows()) {
        printf("FAIL: %ld column overflows\n", zk_column_overflows());
        return 1;
    }
    printf("OK: no column left int64\n");
    return 0;
}
