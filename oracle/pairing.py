"""TEST INFRASTRUCTURE ONLY — optimal ate pairing on BN254 (alt_bn128) and the Groth16 verification equation, in plain
Python integers.  Nothing under rapidsnark-old_amd/ imports this.

Why it exists: the reference repository has no verifier — rapidsnark only PROVES
(/root/reference/src/groth16.cpp:48-254) and leaves verification to snarkjs — and its hot path cannot be built in this image,
so parity with it stays unpinned.  The pairing check is the one test of a proof that needs neither the
reference nor the toxic waste of the key: any (proof.json, public.json, verification key) triple — including a real
Semaphore / iden3-auth key whose trapdoor nobody knows — either satisfies
        e(A, B) = e(alpha1, beta2) * e(sum_i pub_i * IC_i, gamma2) * e(C, delta2)
or it does not.  tools/refcheck/verify.py is the command-line form.

Construction (textbook; restated from the published algorithm, nothing is taken from a dependency — none is vendored):
  Fq2 = Fq[u]/(u^2 + 1);  Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + u;  Fq12 = Fq6[w]/(w^2 - v)     (so w^6 = xi)
  G2 lives on the D-type sextic twist E': y^2 = x^3 + 3/xi, untwist (x', y') -> (x' w^2, y' w^3)
  Miller loop over 6x + 2, x = 4965661367192848881 (the BN parameter), affine twist coordinates, then the two
  Frobenius lines (Q1 = pi(Q), -Q2 = -pi^2(Q)); final exponentiation (q^12 - 1)/r as easy part f^(q^6 - 1) = conj(f)/f followed
  by ONE plain square-and-multiply with the exponent (q^6 + 1)/r — slow (~0.5 s) and obviously right.
Pinned by: bilinearity, non-degeneracy and order checks (tests/test_oracle_py.py), and by every golden proof of this
repository — whose correctness is known independently through the pairing-free trapdoor check — verifying, while a
proof with one coordinate changed does not.
"""
from . import bn254 as bn
from .bn254 import Q_MOD, R_MOD, f2_add, f2_sub, f2_mul, f2_neg, f2_inv, F2_ZERO, F2_ONE

BN_X = 4965661367192848881
ATE_LOOP = 6 * BN_X + 2
XI = (9, 1)


def f2_sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % Q_MOD, 2 * a[0] * a[1] % Q_MOD)


def f2_conj(a):
    return (a[0], (-a[1]) % Q_MOD)


def f2_mul_xi(a):                       # (a0 + a1 u)(9 + u)
    return ((9 * a[0] - a[1]) % Q_MOD, (9 * a[1] + a[0]) % Q_MOD)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


# ---- Fq6 = Fq2[v]/(v^3 - xi): triples (c0, c1, c2)
F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)


def f6_add(a, b):
    return (f2_add(a[0], b[0]), f2_add(a[1], b[1]), f2_add(a[2], b[2]))


def f6_sub(a, b):
    return (f2_sub(a[0], b[0]), f2_sub(a[1], b[1]), f2_sub(a[2], b[2]))


def f6_neg(a):
    return (f2_neg(a[0]), f2_neg(a[1]), f2_neg(a[2]))


def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0, t1, t2 = f2_mul(a0, b0), f2_mul(a1, b1), f2_mul(a2, b2)
    c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(a1, a2), f2_add(b1, b2)), t1), t2)))
    c1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a0, a1), f2_add(b0, b1)), t0), t1), f2_mul_xi(t2))
    c2 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a0, a2), f2_add(b0, b2)), t0), t2), t1)
    return (c0, c1, c2)


def f6_mul_v(a):                        # a * v
    return (f2_mul_xi(a[2]), a[0], a[1])


def f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_sqr(a0), f2_mul_xi(f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul_xi(f2_sqr(a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    d = f2_inv(f2_add(f2_mul(a0, t0), f2_mul_xi(f2_add(f2_mul(a2, t1), f2_mul(a1, t2)))))
    return (f2_mul(t0, d), f2_mul(t1, d), f2_mul(t2, d))


# ---- Fq12 = Fq6[w]/(w^2 - v): pairs (c0, c1)
F12_ONE = (F6_ONE, F6_ZERO)


def f12_mul(a, b):
    t0, t1 = f6_mul(a[0], b[0]), f6_mul(a[1], b[1])
    return (f6_add(t0, f6_mul_v(t1)), f6_sub(f6_sub(f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1])), t0), t1))


def f12_sqr(a):
    return f12_mul(a, a)


def f12_conj(a):                        # a^(q^6)
    return (a[0], f6_neg(a[1]))


def f12_inv(a):
    d = f6_inv(f6_sub(f6_mul(a[0], a[0]), f6_mul_v(f6_mul(a[1], a[1]))))
    return (f6_mul(a[0], d), f6_neg(f6_mul(a[1], d)))


def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


# ---- Miller loop (affine points; P in G1 as (x, y) ints, Q on the twist as ((x0, x1), (y0, y1)); None = infinity)
def _line(T, Q2, P):
    """line through T and Q2 (tangent when equal) on the twist, evaluated at P -> (sparse Fq12 value, T + Q2)"""
    (x1, y1), (x2, y2) = T, Q2
    if x1 == x2 and y1 == y2:
        lam = f2_mul(bn.f2_scalar(f2_sqr(x1), 3), f2_inv(bn.f2_scalar(y1, 2)))
    elif x1 == x2:
        raise ValueError("vertical line in the Miller loop (point of small order?)")
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), x1), x2)
    y3 = f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)
    xp, yp = P
    # l = yP - (lam xP) w + (lam x1 - y1) w^3      (w^3 = v w)
    c0 = ((yp % Q_MOD, 0), F2_ZERO, F2_ZERO)
    c1 = (f2_neg(bn.f2_scalar(lam, xp)), f2_sub(f2_mul(lam, x1), y1), F2_ZERO)
    return (c0, c1), (x3, y3)


def _frobenius_twist(Q, power):
    x, y = Q
    if power == 1:
        return (f2_mul(f2_conj(x), f2_pow(XI, (Q_MOD - 1) // 3)), f2_mul(f2_conj(y), f2_pow(XI, (Q_MOD - 1) // 2)))
    return (f2_mul(x, f2_pow(XI, (Q_MOD * Q_MOD - 1) // 3)), f2_mul(y, f2_pow(XI, (Q_MOD * Q_MOD - 1) // 2)))


def miller_loop(P, Q):
    if P is None or Q is None:
        return F12_ONE
    f, T = F12_ONE, Q
    for bit in bin(ATE_LOOP)[3:]:
        l, T = _line(T, T, P)
        f = f12_mul(f12_sqr(f), l)
        if bit == "1":
            l, T = _line(T, Q, P)
            f = f12_mul(f, l)
    Q1 = _frobenius_twist(Q, 1)
    Q2 = _frobenius_twist(Q, 2)
    l, T = _line(T, Q1, P)
    f = f12_mul(f, l)
    l, T = _line(T, (Q2[0], f2_neg(Q2[1])), P)
    return f12_mul(f, l)


def final_exponentiation(f):
    f = f12_mul(f12_conj(f), f12_inv(f))                      # f^(q^6 - 1)
    return f12_pow(f, (Q_MOD ** 6 + 1) // R_MOD)              # (q^6 + 1) / r is an integer: r | q^4 - q^2 + 1 | q^6 + 1


def pairing(P, Q):
    """e(P, Q) for P in G1, Q in G2 (affine tuples as everywhere in oracle/bn254.py)."""
    return final_exponentiation(miller_loop(P, Q))


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 with ONE final exponentiation (what the EIP-197 precompile computes)."""
    f = F12_ONE
    for P, Q in pairs:
        f = f12_mul(f, miller_loop(P, Q))
    return final_exponentiation(f) == F12_ONE


# ---- Groth16 verification (snarkjs groth16_verify; the reference has no counterpart)
def groth16_verify(vk, public_signals, proof):
    """vk: dict with alpha1 (G1), beta2, gamma2, delta2 (G2), IC (list of G1); public_signals: ints; proof: (A, B, C) affine.
    e(-A, B) * e(alpha1, beta2) * e(vk_x, gamma2) * e(C, delta2) == 1"""
    A, B, C = proof
    if len(vk["IC"]) != len(public_signals) + 1:
        raise ValueError("verification key has %d IC points for %d public signals" % (len(vk["IC"]), len(public_signals)))
    for pt, grp in ((A, bn.G1), (C, bn.G1), (B, bn.G2)):
        if pt is None or not grp.is_on_curve(pt):
            return False
    if bn.G2.mul(B, R_MOD) is not None:                       # G2 has a cofactor: B must lie in the order-r subgroup
        return False
    vk_x = vk["IC"][0]
    for s, ic in zip(public_signals, vk["IC"][1:]):
        if not 0 <= s < R_MOD:
            return False
        vk_x = bn.G1.add(vk_x, bn.G1.mul(ic, s))
    negA = (A[0], (-A[1]) % Q_MOD)
    return pairing_product_is_one([(negA, B), (vk["alpha1"], vk["beta2"]), (vk_x, vk["gamma2"]), (C, vk["delta2"])])
