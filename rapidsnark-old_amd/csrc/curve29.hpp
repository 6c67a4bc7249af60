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
    // products in PAIRS of independent chains (field29.hpp run2): (U2, S2), (PP, R2), (PPP, Q), then ZZ3 | ZZZ3 | Y3 together
    F U2, S2;
    F::mul2(U2, p.x, acc.zz, S2, p.y, acc.zzz);
    F P = F::sub_nc(U2, acc.x);
    F R = F::sub_nc(S2, acc.y);
    if (P.is_zero()) {
        if (R.is_zero()) acc = dbl_affine(Affine<F>{p.x, F::carry(p.y)});
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP, R2;
    F::sqr2(PP, P, R2, R);
    F PPP, Q;
    F::mul2(PPP, P, PP, Q, acc.x, PP);
    F X3;
#pragma unroll
    for (int i = 0; i < 9; i++) X3.l[i] = R2.l[i] - PPP.l[i] - (Q.l[i] << 1);
    X3 = F::carry(X3);
    const F D = F::sub_nc(Q, X3), NY = F::neg_lazy(acc.y);
    F Y3;
#if defined(ZK_G1_RUN2_TAIL)
    F::run2(Y3, typename F::JMulAdd2{R, D, NY, PPP}, acc.zz, typename F::JMul{acc.zz, PP});      // zz' in place: column k writes limb k-9, reads limbs >= k-8
    acc.zzz = F::mul(acc.zzz, PPP);
#else
    // zz' | zzz' | Y3 as three chains at once (field29.hpp run3): Y3's two products alternate with zz' first, then with zzz' —
    // no product is left alone.  zz', zzz' in place: column k writes limb k-9 and reads limbs >= k-8.
    F::run3(acc.zz, typename F::JMul{acc.zz, PP}, acc.zzz, typename F::JMul{acc.zzz, PPP}, Y3, typename F::JMulAdd2{R, D, NY, PPP});
#endif
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


#if defined(__HIPCC__)
// ---- Fq2 split across a lane pair (device only): lane 2k holds the real component, lane 2k+1 the
// imaginary one, and each computes ONE fused double product per Fq2 multiplication after a DPP
// quad-permute exchange of the operands.  The arithmetic is operand-for-operand the one of
// Fp2T<Fq29> (same products, same carries, same results); what changes is the register file: a
// G2 accumulator needs the registers of a G1 one (3 waves/SIMD instead of 1).  Control flow must be
// uniform inside a pair: every predicate below is the AND over both lanes.
struct Fq2s {
    Fq29 v;
    static __device__ __forceinline__ bool odd() { return (threadIdx.x & 1u) != 0; }
    static __device__ __forceinline__ int xchg(int x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true); }   // quad_perm [1,0,3,2]
    static __device__ __forceinline__ Fq29 other(const Fq29 &a) {
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = xchg(a.l[i]);
        return r;
    }
    static __device__ __forceinline__ Fq29 sel(bool c, const Fq29 &a, const Fq29 &b) {   // c ? a : b
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
        return r;
    }
    static __device__ __forceinline__ bool both(bool mine) { return mine && xchg(mine ? 1 : 0) != 0; }
    static __device__ __forceinline__ Fq2s zero() { return Fq2s{Fq29::zero()}; }
    static __device__ __forceinline__ Fq2s one() { return Fq2s{sel(odd(), Fq29::zero(), Fq29::one())}; }
    __device__ __forceinline__ bool is_zero() const { return both(v.is_zero()); }
    __device__ __forceinline__ bool is_zero_raw() const { return both(v.is_zero_raw()); }
    static __device__ __forceinline__ Fq2s add(const Fq2s &x, const Fq2s &y) { return Fq2s{Fq29::add(x.v, y.v)}; }
    static __device__ __forceinline__ Fq2s sub(const Fq2s &x, const Fq2s &y) { return Fq2s{Fq29::sub(x.v, y.v)}; }
    static __device__ __forceinline__ Fq2s neg(const Fq2s &x) { return Fq2s{Fq29::neg(x.v)}; }
    static __device__ __forceinline__ Fq2s dbl(const Fq2s &x) { return Fq2s{Fq29::dbl(x.v)}; }
    static __device__ __forceinline__ Fq29 from_even(const Fq29 &a) {   // both lanes: the even lane's value (quad_perm [0,0,2,2])
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = __builtin_amdgcn_mov_dpp(a.l[i], 0xA0, 0xF, 0xF, true);
        return r;
    }
    static __device__ __forceinline__ Fq29 from_odd(const Fq29 &a) {    // both lanes: the odd lane's value (quad_perm [1,1,3,3])
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = __builtin_amdgcn_mov_dpp(a.l[i], 0xF5, 0xF, 0xF, true);
        return r;
    }
    // re = a0 b0 - a1 b1 (even lane), im = a0 b1 + a1 b0 (odd lane): with y = own component of b and
    // yo = the partner's, both are  a0 * y + (+-a1) * yo  — one instruction stream, no operand selects
    static __device__ __forceinline__ Fq2s mul(const Fq2s &x, const Fq2s &y) {
        const Fq29 a0 = from_even(x.v), a1 = from_odd(x.v), yo = other(y.v);
        return Fq2s{Fq29::mul_add2(a0, y.v, sel(odd(), a1, Fq29::neg_lazy(a1)), yo)};
    }
    // An operand used in several products is exchanged once: Pre holds (a0, +-a1) of it, the other
    // factor only needs its partner component.
    struct Pre {
        Fq29 a0, a1s;
    };
    static __device__ __forceinline__ Pre prepare(const Fq2s &x) {
        const Fq29 a1 = from_odd(x.v);
        return Pre{from_even(x.v), sel(odd(), a1, Fq29::neg_lazy(a1))};
    }
    static __device__ __forceinline__ Fq2s mul(const Pre &x, const Fq2s &y) {
        return Fq2s{Fq29::mul_add2(x.a0, y.v, x.a1s, other(y.v))};
    }
    // re = (a0 + a1)(a0 - a1), im = 2 a0 a1 = (a0 + a0) * a1:  u = a0 + partner, w = own - (even ? a1 : 0)
    static __device__ __forceinline__ void sqr_operands(Fq29 &u, Fq29 &w, const Fq2s &x) {
        const Fq29 a0 = from_even(x.v), xo = other(x.v);
        const int keep = odd() ? 0 : -1;
        Fq29 t;
#pragma unroll
        for (int i = 0; i < 9; i++) t.l[i] = xo.l[i] & keep;
        u = Fq29::add_nc(a0, xo);
        w = Fq29::sub(x.v, t);
    }
    static __device__ __forceinline__ Fq2s sqr(const Fq2s &x) {
        Fq29 u, w;
        sqr_operands(u, w, x);
        return Fq2s{Fq29::mul(u, w)};
    }
    // pairs of independent products: the two chains interleaved (field29.hpp run2)
    static __device__ __forceinline__ void mul2(Fq2s &r0, const Pre &x0, const Fq2s &y0, Fq2s &r1, const Pre &x1, const Fq2s &y1) {
        const Fq29 yo0 = other(y0.v), yo1 = other(y1.v);
        Fq29::run2(r0.v, Fq29::JMulAdd2{x0.a0, y0.v, x0.a1s, yo0}, r1.v, Fq29::JMulAdd2{x1.a0, y1.v, x1.a1s, yo1});
    }
    static __device__ __forceinline__ void mul2(Fq2s &r0, const Fq2s &x0, const Fq2s &y0, Fq2s &r1, const Fq2s &x1, const Fq2s &y1) {
        mul2(r0, prepare(x0), y0, r1, prepare(x1), y1);
    }
    static __device__ __forceinline__ void sqr2(Fq2s &r0, const Fq2s &x0, Fq2s &r1, const Fq2s &x1) {
        Fq29 u0, w0, u1, w1;
        sqr_operands(u0, w0, x0);
        sqr_operands(u1, w1, x1);
        Fq29::mul2(r0.v, u0, w0, r1.v, u1, w1);
    }
};

// the bound-tracked mixed add of XYZZ<Fq2r> above, one component per lane
__device__ __forceinline__ void madd(XYZZ<Fq2s> &acc, const Affine<Fq2s> &p) {
    typedef Fq29 B;
    typedef Fq2s F;
    if (p.is_inf()) return;
    if (acc.is_inf()) {
        acc = XYZZ<F>{p.x, F{B::carry(p.y.v)}, F::one(), F::one()};
        return;
    }
    // products in pairs of independent chains: (U2, S2), (PP, R2), (PPP, Q), then zz' | zzz' | Y3 as three chains at once.
    // Y3 = R*D - Y1*PPP is ONE job per lane — four operand products, one Montgomery reduction (G1's fused Y3, line 48,
    // in the lane-pair form): re = R0 D0 - R1 D1 - y0 PPP0 + y1 PPP1, im = R0 D1 + R1 D0 - y0 PPP1 - y1 PPP0.  R and D
    // are carried first (limbs >= -1) so that in either lane two of the four products are non-negative and two
    // non-positive: a column holds at most 18 terms of one sign (field29.hpp JMulAdd4).
    F U2, S2;
    F::mul2(U2, p.x, acc.zz, S2, p.y, acc.zzz);
    F P{B::sub_nc(U2.v, acc.x.v)};
    F R{B::sub(S2.v, acc.y.v)};
    if (P.is_zero()) {
        if (R.is_zero()) acc = dbl_affine(Affine<F>{p.x, F{B::carry(p.y.v)}});
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP, R2;
    F::sqr2(PP, P, R2, R);
    const F::Pre pp = F::prepare(PP);             // PP and PPP enter three and two products: exchanged once each
    F PPP, Q;
    F::mul2(PPP, pp, P, Q, pp, acc.x);
    const F::Pre ppp = F::prepare(PPP);
#pragma unroll
    for (int i = 0; i < 9; i++) acc.x.v.l[i] = R2.v.l[i] - PPP.v.l[i] - (Q.v.l[i] << 1);
    acc.x.v = B::carry(acc.x.v);                  // X3
    const B D = B::sub(Q.v, acc.x.v), Do = F::other(D);
    const B ny = B::neg_lazy(acc.y.v), nyo = F::other(ny);
    const F::Pre rp = F::prepare(R);
    const B zzo = F::other(acc.zz.v), zzzo = F::other(acc.zzz.v);
    B Y3;
    B::run3(acc.zz.v, B::JMulAdd2{pp.a0, acc.zz.v, pp.a1s, zzo},                            // zz', zzz' in place (see the G1 madd)
            acc.zzz.v, B::JMulAdd2{ppp.a0, acc.zzz.v, ppp.a1s, zzzo},
            Y3, B::JMulAdd4{rp.a0, D, rp.a1s, Do, ppp.a0, ny, ppp.a1s, nyo});
    acc.y = F{Y3};                                // a product output: limbs >= 0, the next S2 - y stays tight
}
#endif

}   // namespace zk
