// Mixed addition acc += P specialised for the 29-bit-limb fields (field29.hpp) — the innermost
// operation of the MSM (one per sorted entry).  Same formulas as the generic madd (curve.hpp,
// madd-2008-s: 8M + 2S), but with the limb bounds tracked by hand so that carry passes are spent
// only where a bound would otherwise break:
//   * a - b of two values whose limbs are non-negative (product outputs) has |limb| < 2^29:
//     no carry;
//   * X3 = R^2 - PPP - 2Q is formed limb-wise and carried ONCE (generic: sub, dbl, sub = 3);
//   * G1: Y3 = R(Q - X3) - Y1*PPP is one fused double product (one Montgomery reduction).
// Invariants on the stored accumulator: x carried (limbs in [-8, 2^29+8]), y / zz / zzz are
// product outputs (G1) or carried (G2 y): limbs >= -8 — which is what makes the next U2 - x and
// S2 - y differences tight.  Value ranges stay within (-5p, 5p), far inside the (-13p, 13p)
// the 2^261 radix tolerates.  Measured: G1 add 2956 -> ~2350 VALU instructions.
#pragma once
#include "curve.hpp"
#include "field29.hpp"

namespace zk {

typedef Fp2T<Fq29> Fq2r;

ZK_HD void madd(XYZZ<Fq29> &acc, const Affine<Fq29> &p) {
    typedef Fq29 F;
    if (p.is_inf()) return;
    if (acc.is_inf()) {
        acc = XYZZ<F>{p.x, F::carry(p.y), F::one(), F::one()};   // p.y may be a lazily negated value
        return;
    }
    F U2 = F::mul(p.x, acc.zz);
    F S2 = F::mul(p.y, acc.zzz);
    F P = F::sub_nc(U2, acc.x);
    F R = F::sub_nc(S2, acc.y);
    if (P.is_zero()) {
        if (R.is_zero()) acc = dbl_affine(Affine<F>{p.x, F::carry(p.y)});
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = F::sqr(P);
    F PPP = F::mul(P, PP);
    F Q = F::mul(acc.x, PP);
    F R2 = F::sqr(R);
    F X3;
#pragma unroll
    for (int i = 0; i < 9; i++) X3.l[i] = R2.l[i] - PPP.l[i] - (Q.l[i] << 1);
    X3 = F::carry(X3);
    F Y3 = F::mul_add2(R, F::sub_nc(Q, X3), F::neg_lazy(acc.y), PPP);
    acc.zz = F::mul(acc.zz, PP);
    acc.zzz = F::mul(acc.zzz, PPP);
    acc.x = X3;
    acc.y = Y3;
}

// Fq2 helpers with explicit bounds: inputs |limb| <= 2^29+16 ("tight")
ZK_HD Fq2r f2_mul_tight(const Fq2r &x, const Fq2r &y) {   // both tight -> product outputs (limbs >= 0)
    return Fq2r{Fq29::mul_add2(x.a, y.a, Fq29::neg_lazy(x.b), y.b), Fq29::mul_add2(x.a, y.b, x.b, y.a)};
}
ZK_HD Fq2r f2_sqr_tight(const Fq2r &x) {                  // (a+b)(a-b) + 2ab u ; one carry (on a-b)
    return Fq2r{Fq29::mul(Fq29::add_nc(x.a, x.b), Fq29::sub(x.a, x.b)), Fq29::mul(Fq29::dbl_lazy(x.a), x.b)};
}

ZK_HD void madd(XYZZ<Fq2r> &acc, const Affine<Fq2r> &p) {
    typedef Fq29 B;
    typedef Fq2r F;
    if (p.is_inf()) return;
    if (acc.is_inf()) {
        acc = XYZZ<F>{p.x, F{B::carry(p.y.a), B::carry(p.y.b)}, F::one(), F::one()};
        return;
    }
    F U2 = f2_mul_tight(p.x, acc.zz);
    F S2 = f2_mul_tight(p.y, acc.zzz);
    F P{B::sub_nc(U2.a, acc.x.a), B::sub_nc(U2.b, acc.x.b)};
    F R{B::sub_nc(S2.a, acc.y.a), B::sub_nc(S2.b, acc.y.b)};
    if (P.is_zero()) {
        if (R.is_zero()) acc = dbl_affine(Affine<F>{p.x, F{B::carry(p.y.a), B::carry(p.y.b)}});
        else acc = XYZZ<F>::inf();
        return;
    }
    // ordered so that every temporary dies as early as possible (the kernel lives at the edge of
    // the 256-VGPR file: fewer live values = fewer AGPR spill moves)
    F PP = f2_sqr_tight(P);
    acc.zz = f2_mul_tight(acc.zz, PP);            // zz' (old zz dead)
    F Q = f2_mul_tight(acc.x, PP);                // old x dead
    F PPP = f2_mul_tight(P, PP);                  // P, PP dead
    acc.zzz = f2_mul_tight(acc.zzz, PPP);         // zzz' (old zzz dead)
    F T2 = f2_mul_tight(acc.y, PPP);              // old y dead
    F R2 = f2_sqr_tight(R);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc.x.a.l[i] = R2.a.l[i] - PPP.a.l[i] - (Q.a.l[i] << 1);
        acc.x.b.l[i] = R2.b.l[i] - PPP.b.l[i] - (Q.b.l[i] << 1);
    }
    acc.x.a = B::carry(acc.x.a);                  // X3
    acc.x.b = B::carry(acc.x.b);
    F D{B::sub_nc(Q.a, acc.x.a), B::sub_nc(Q.b, acc.x.b)};
    F T1 = f2_mul_tight(R, D);
    acc.y = F{B::sub(T1.a, T2.a), B::sub(T1.b, T2.b)};        // carried: keeps the next S2 - y tight
}

// lazily negated y for a negative digit (limbs <= 0, magnitude unchanged: still a tight operand)
ZK_HD void negate_y(Affine<Fq29> &p) { p.y = Fq29::neg_lazy(p.y); }
ZK_HD void negate_y(Affine<Fq2r> &p) {
    p.y.a = Fq29::neg_lazy(p.y.a);
    p.y.b = Fq29::neg_lazy(p.y.b);
}

}   // namespace zk
