// BN254 G1 / G2 group arithmetic, generic over the coordinate field (Fq or Fq2).
//
// Replaces ffiasm's Curve<> (absent submodule; call sites src/groth16.cpp:173-251).
// Accumulators are extended-Jacobian "XYZZ" (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// a mixed add costs 8M+2S with no inversion — the shape bucket accumulation wants.
// Affine encoding follows the zkey: Montgomery coordinates, all-zero = infinity.
// Every special case (acc = O, P = O, P = Q, P = -Q) is handled explicitly: real zkeys
// contain zero points and repeated points.
#pragma once
#include "field.hpp"

namespace zk {

// Fp2 = B[u]/(u^2+1) over a base field B (device: 8x32-bit Fq; host: 4x64-bit Fq64);
// layout {a, b} = a + b*u, matching the reference's G2 bytes x.a|x.b|y.a|y.b
// (src/groth16.cpp:280-284).
template <class B>
struct Fp2T {
    B a, b;
    ZK_HD static Fp2T zero() { return Fp2T{B::zero(), B::zero()}; }
    ZK_HD static Fp2T one() { return Fp2T{B::one(), B::zero()}; }
    ZK_HD bool is_zero() const { return a.is_zero() && b.is_zero(); }
    ZK_HD bool is_zero_raw() const { return a.is_zero_raw() && b.is_zero_raw(); }
    ZK_HD bool operator==(const Fp2T &o) const { return a == o.a && b == o.b; }
    ZK_HD bool operator!=(const Fp2T &o) const { return !(*this == o); }
    ZK_HD static Fp2T add(const Fp2T &x, const Fp2T &y) { return Fp2T{B::add(x.a, y.a), B::add(x.b, y.b)}; }
    ZK_HD static Fp2T sub(const Fp2T &x, const Fp2T &y) { return Fp2T{B::sub(x.a, y.a), B::sub(x.b, y.b)}; }
    ZK_HD static Fp2T neg(const Fp2T &x) { return Fp2T{B::neg(x.a), B::neg(x.b)}; }
    ZK_HD static Fp2T dbl(const Fp2T &x) { return Fp2T{B::dbl(x.a), B::dbl(x.b)}; }
    ZK_HD static Fp2T mul(const Fp2T &x, const Fp2T &y) {
        if constexpr (B::FUSED_MULADD) {
            // (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u as two fused double products (field29.hpp)
            return Fp2T{B::mul_add2(x.a, y.a, B::neg_lazy(x.b), y.b), B::mul_add2(x.a, y.b, x.b, y.a)};
        } else {   // Karatsuba: 3 base mul
            B v0 = B::mul(x.a, y.a);
            B v1 = B::mul(x.b, y.b);
            B s = B::mul(B::add(x.a, x.b), B::add(y.a, y.b));
            return Fp2T{B::sub(v0, v1), B::sub(B::sub(s, v0), v1)};
        }
    }
    ZK_HD static Fp2T sqr(const Fp2T &x) {                  // 2 base mul
        if constexpr (B::FUSED_MULADD) {
            return Fp2T{B::mul(B::add(x.a, x.b), B::sub(x.a, x.b)), B::mul(B::dbl_lazy(x.a), x.b)};
        } else {
            B t = B::mul(x.a, x.b);
            B c0 = B::mul(B::add(x.a, x.b), B::sub(x.a, x.b));
            return Fp2T{c0, B::dbl(t)};
        }
    }
    ZK_HD static Fp2T inv(const Fp2T &x) {                  // host-side only (final affine)
        B d = B::inv(B::add(B::sqr(x.a), B::sqr(x.b)));
        return Fp2T{B::mul(x.a, d), B::neg(B::mul(x.b, d))};
    }
};
typedef Fp2T<Fq> Fq2;

template <class F>
struct Affine {
    F x, y;
    ZK_HD bool is_inf() const { return x.is_zero_raw() && y.is_zero_raw(); }   // all-zero encoding
    ZK_HD static Affine inf() { return Affine{F::zero(), F::zero()}; }
};

template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    ZK_HD bool is_inf() const { return zz.is_zero_raw(); }   // infinity is always ASSIGNED (zz = 0 limbs), never computed
    ZK_HD static XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
    ZK_HD static XYZZ from_affine(const Affine<F> &p) {
        if (p.is_inf()) return inf();
        return XYZZ{p.x, p.y, F::one(), F::one()};
    }
};

// 2*P for an affine P != O  (dbl-2008-s-1 with ZZ = ZZZ = 1, curve a = 0)
template <class F>
ZK_HD XYZZ<F> dbl_affine(const Affine<F> &p) {
    F U = F::dbl(p.y);
    F V = F::sqr(U);
    F W = F::mul(U, V);
    F S = F::mul(p.x, V);
    F X2 = F::sqr(p.x);
    F M = F::add(F::dbl(X2), X2);
    XYZZ<F> r;
    r.x = F::sub(F::sqr(M), F::dbl(S));
    r.y = F::sub(F::mul(M, F::sub(S, r.x)), F::mul(W, p.y));
    r.zz = V;
    r.zzz = W;
    return r;   // y = 0 cannot happen on a prime-order curve
}

template <class F>
ZK_HD XYZZ<F> dbl(const XYZZ<F> &p) {
    if (p.is_inf()) return p;
    F U = F::dbl(p.y);
    F V = F::sqr(U);
    F W = F::mul(U, V);
    F S = F::mul(p.x, V);
    F X2 = F::sqr(p.x);
    F M = F::add(F::dbl(X2), X2);
    XYZZ<F> r;
    r.x = F::sub(F::sqr(M), F::dbl(S));
    r.y = F::sub(F::mul(M, F::sub(S, r.x)), F::mul(W, p.y));
    r.zz = F::mul(V, p.zz);
    r.zzz = F::mul(W, p.zzz);
    return r;
}

// acc += p  (mixed add, madd-2008-s); p affine with y already sign-adjusted by the caller
template <class F>
ZK_HD void madd(XYZZ<F> &acc, const Affine<F> &p) {
    if (p.is_inf()) return;
    if (acc.is_inf()) {
        acc = XYZZ<F>{p.x, p.y, F::one(), F::one()};
        return;
    }
    F U2 = F::mul(p.x, acc.zz);
    F S2 = F::mul(p.y, acc.zzz);
    F P = F::sub(U2, acc.x);
    F R = F::sub(S2, acc.y);
    if (P.is_zero()) {
        if (R.is_zero()) acc = dbl_affine(p);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = F::sqr(P);
    F PPP = F::mul(P, PP);
    F Q = F::mul(acc.x, PP);
    F X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    F Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(acc.y, PPP));
    acc.zz = F::mul(acc.zz, PP);
    acc.zzz = F::mul(acc.zzz, PPP);
    acc.x = X3;
    acc.y = Y3;
}

// acc += q  (general add, add-2008-s)
template <class F>
ZK_HD void add(XYZZ<F> &acc, const XYZZ<F> &q) {
    if (q.is_inf()) return;
    if (acc.is_inf()) {
        acc = q;
        return;
    }
    F U1 = F::mul(acc.x, q.zz);
    F U2 = F::mul(q.x, acc.zz);
    F S1 = F::mul(acc.y, q.zzz);
    F S2 = F::mul(q.y, acc.zzz);
    F P = F::sub(U2, U1);
    F R = F::sub(S2, S1);
    if (P.is_zero()) {
        if (R.is_zero()) acc = dbl(acc);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = F::sqr(P);
    F PPP = F::mul(P, PP);
    F Q = F::mul(U1, PP);
    F X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    F Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(S1, PPP));
    acc.zz = F::mul(F::mul(acc.zz, q.zz), PP);
    acc.zzz = F::mul(F::mul(acc.zzz, q.zzz), PPP);
    acc.x = X3;
    acc.y = Y3;
}

template <class F>
ZK_HD XYZZ<F> neg(const XYZZ<F> &p) {
    return XYZZ<F>{p.x, F::neg(p.y), p.zz, p.zzz};
}

// ---- host-side helpers for the O(1) tail of the proof (src/groth16.cpp:219-251) ----
// scalar: 32 little-endian bytes as 8 u32 limbs, standard form (E.g1.mulByScalar).
template <class F>
inline XYZZ<F> scalar_mul(const XYZZ<F> &p, const u32 k[8]) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = 255; i >= 0; i--) {
        acc = dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1u) add(acc, p);
    }
    return acc;
}

// projective -> affine (E.g1.copy(Affine&, Point&), src/groth16.cpp:249-251)
template <class F>
inline Affine<F> to_affine(const XYZZ<F> &p) {
    if (p.is_inf()) return Affine<F>::inf();
    // x = X/ZZ, y = Y/ZZZ ; one inversion: 1/(ZZ*ZZZ)
    F t = F::inv(F::mul(p.zz, p.zzz));
    F izz = F::mul(t, p.zzz);
    F izzz = F::mul(t, p.zz);
    return Affine<F>{F::mul(p.x, izz), F::mul(p.y, izzz)};
}

typedef Affine<Fq> G1Affine;
typedef Affine<Fq2> G2Affine;
typedef XYZZ<Fq> G1XYZZ;
typedef XYZZ<Fq2> G2XYZZ;

}   // namespace zk
