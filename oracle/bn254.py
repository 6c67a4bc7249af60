"""BN254 (alt_bn128) big-integer arithmetic — TEST INFRASTRUCTURE ONLY.

This file is part of the ORACLE: a CPU restatement (Python big-int) of the
arithmetic that the reference obtains from its absent `depends/ffiasm`
submodule (reference `.gitmodules:7-9`; call sites listed in SURVEY.md §2.2).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import it.  The product path (rapidsnark-old_amd/) never does.

PARITY UNPINNED: the reference ships no tests/golden vectors
(`package.json:7`), and ffiasm cannot be built here (no sources, no nasm).
Correctness is pinned by mathematics instead: curve/field identities
(r*G == O, Montgomery constants of SURVEY §A.2) and the pairing-free trapdoor
check in `oracle/groth16_ref.py`.

Conventions (reference call sites in parentheses):
  * Fr / Fq elements are Python ints in [0, p).  "Montgomery form" of x is
    x*R mod p with R = 2^256 (`src/groth16.cpp:162` needs fromMontgomery before
    the MSM; zkey points are Montgomery).
  * G1 affine = (x, y) or None for infinity; G2 affine = ((xa, xb), (ya, yb))
    with Fq2 = a + b*u, u^2 = -1 (`src/groth16.cpp:261` uses .a/.b).
"""

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # main_prover.cpp:34
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # tasksfile.js:10
MONT_R = 1 << 256
TWO_ADICITY = 28
# snarkjs/ffjavascript convention: w_{2^28} = 5^((r-1)/2^28)  
ROOT_2_28 = pow(5, (R_MOD - 1) >> TWO_ADICITY, R_MOD)

G1_GEN = (1, 2)
# EIP-197 generator of G2 (x = xa + xb*u, y = ya + yb*u)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


def fr_root(log2n: int) -> int:
    """Primitive 2^log2n-th root of unity in Fr (ffiasm FFT::root convention)."""
    assert 0 <= log2n <= TWO_ADICITY
    return pow(ROOT_2_28, 1 << (TWO_ADICITY - log2n), R_MOD)


def to_mont(x: int, p: int) -> int:
    return (x * MONT_R) % p


def from_mont(x: int, p: int) -> int:
    return (x * pow(MONT_R, -1, p)) % p


def mont_mul(a: int, b: int, p: int) -> int:
    """a*b*R^-1 mod p  (what `E.fr.mul` computes)."""
    return (a * b * pow(MONT_R, -1, p)) % p


# ----------------------------------------------------------------- Fq2
def f2_add(a, b):
    return ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)


def f2_sub(a, b):
    return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)


def f2_neg(a):
    return ((-a[0]) % Q_MOD, (-a[1]) % Q_MOD)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q_MOD)
    return ((a[0] * d) % Q_MOD, (-a[1] * d) % Q_MOD)


def f2_scalar(a, k):
    return ((a[0] * k) % Q_MOD, (a[1] * k) % Q_MOD)


F2_ZERO = (0, 0)
F2_ONE = (1, 0)
# twist coefficient b' = 3 / (9 + u)
G2_B = f2_mul((3, 0), f2_inv((9, 1)))


# ----------------------------------------------------------------- generic affine group law
class _Fq:
    zero = 0
    one = 1
    add = staticmethod(lambda a, b: (a + b) % Q_MOD)
    sub = staticmethod(lambda a, b: (a - b) % Q_MOD)
    mul = staticmethod(lambda a, b: (a * b) % Q_MOD)
    neg = staticmethod(lambda a: (-a) % Q_MOD)
    inv = staticmethod(lambda a: pow(a, -1, Q_MOD))
    small = staticmethod(lambda a, k: (a * k) % Q_MOD)


class _Fq2:
    zero = F2_ZERO
    one = F2_ONE
    add = staticmethod(f2_add)
    sub = staticmethod(f2_sub)
    mul = staticmethod(f2_mul)
    neg = staticmethod(f2_neg)
    inv = staticmethod(f2_inv)
    small = staticmethod(f2_scalar)


class Curve:
    """Short Weierstrass y^2 = x^3 + b over field F; Jacobian internally, affine in/out."""

    def __init__(self, F, b, gen):
        self.F, self.b, self.gen = F, b, gen

    def is_on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), self.b)

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    # --- Jacobian helpers (X, Y, Z), Z==zero => infinity
    def _to_jac(self, P):
        F = self.F
        return (F.one, F.one, F.zero) if P is None else (P[0], P[1], F.one)

    def _from_jac(self, J):
        F = self.F
        X, Y, Z = J
        if Z == F.zero:
            return None
        zi = F.inv(Z)
        zi2 = F.mul(zi, zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def _jdbl(self, J):
        F = self.F
        X, Y, Z = J
        if Z == F.zero or Y == F.zero:
            return (F.one, F.one, F.zero)
        A = F.mul(X, X)
        B = F.mul(Y, Y)
        C = F.mul(B, B)
        t = F.add(X, B)
        D = F.small(F.sub(F.sub(F.mul(t, t), A), C), 2)
        E = F.small(A, 3)
        Fv = F.mul(E, E)
        X3 = F.sub(Fv, F.small(D, 2))
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.small(C, 8))
        Z3 = F.small(F.mul(Y, Z), 2)
        return (X3, Y3, Z3)

    def _jadd(self, J1, J2):
        F = self.F
        X1, Y1, Z1 = J1
        X2, Y2, Z2 = J2
        if Z1 == F.zero:
            return J2
        if Z2 == F.zero:
            return J1
        Z1Z1 = F.mul(Z1, Z1)
        Z2Z2 = F.mul(Z2, Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(Y1, F.mul(Z2, Z2Z2))
        S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
        if U1 == U2:
            if S1 == S2:
                return self._jdbl(J1)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        Rr = F.sub(S2, S1)
        HH = F.mul(H, H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.mul(Rr, Rr), HHH), F.small(V, 2))
        Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def add(self, P, Q):
        return self._from_jac(self._jadd(self._to_jac(P), self._to_jac(Q)))

    def sub(self, P, Q):
        return self.add(P, self.neg(Q))

    def dbl(self, P):
        return self._from_jac(self._jdbl(self._to_jac(P)))

    def mul(self, P, k: int):
        """k*P for any non-negative integer k (ffiasm mulByScalar takes raw LE bytes)."""
        assert k >= 0
        F = self.F
        acc = (F.one, F.one, F.zero)
        if P is None or k == 0:
            return None
        J = self._to_jac(P)
        for bit in bin(k)[2:]:
            acc = self._jdbl(acc)
            if bit == "1":
                acc = self._jadd(acc, J)
        return self._from_jac(acc)

    def msm(self, points, scalars):
        """Naive sum_i scalars[i]*points[i] (semantics of multiMulByScalar)."""
        F = self.F
        acc = (F.one, F.one, F.zero)
        for P, k in zip(points, scalars):
            if P is None or k == 0:
                continue
            acc = self._jadd(acc, self._to_jac(self.mul(P, k)))
        return self._from_jac(acc)

    def fixed_base_table(self, P, bits=256, w=8):
        """table[j][d] = d * 2^(w*j) * P in Jacobian, for fast many-scalar generation."""
        F = self.F
        tbl = []
        base = self._to_jac(P)
        for _ in range((bits + w - 1) // w):
            row = [(F.one, F.one, F.zero)]
            for d in range(1, 1 << w):
                row.append(self._jadd(row[-1], base))
            tbl.append(row)
            for _ in range(w):
                base = self._jdbl(base)
        return (tbl, w)

    def mul_fixed(self, table, k: int):
        tbl, w = table
        F = self.F
        acc = (F.one, F.one, F.zero)
        j = 0
        mask = (1 << w) - 1
        while k:
            d = k & mask
            if d:
                acc = self._jadd(acc, tbl[j][d])
            k >>= w
            j += 1
        return self._from_jac(acc)


G1 = Curve(_Fq, 3, G1_GEN)
G2 = Curve(_Fq2, G2_B, G2_GEN)


# ----------------------------------------------------------------- NTT over Fr (values are plain ints)
def ntt(vals, inverse=False):
    """Natural-order in/out radix-2 transform over the n-th roots of unity.

    Semantics of ffiasm `FFT::fft` / `FFT::ifft` as the reference uses them
    (`src/groth16.cpp:102,115`): fft: X[i] = sum_j x[j] w^(ij);
    ifft includes the 1/n scale.  O(n log n), iterative.
    """
    n = len(vals)
    assert n & (n - 1) == 0 and n > 0
    logn = n.bit_length() - 1
    a = list(vals)
    # bit reversal
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    w_n = fr_root(logn)
    if inverse:
        w_n = pow(w_n, -1, R_MOD)
    m = 1
    while m < n:
        w_m = pow(w_n, n // (2 * m), R_MOD)
        for k in range(0, n, 2 * m):
            w = 1
            for jj in range(m):
                u = a[k + jj]
                v = (a[k + jj + m] * w) % R_MOD
                a[k + jj] = (u + v) % R_MOD
                a[k + jj + m] = (u - v) % R_MOD
                w = (w * w_m) % R_MOD
        m *= 2
    if inverse:
        ninv = pow(n, -1, R_MOD)
        a = [(x * ninv) % R_MOD for x in a]
    return a


# ----------------------------------------------------------------- byte encodings
def int_to_le32(x: int) -> bytes:
    return int(x).to_bytes(32, "little")


def le32_to_int(b: bytes) -> int:
    return int.from_bytes(b, "little")


def g1_to_bytes(P) -> bytes:
    """Affine Montgomery little-endian x||y; infinity = 64 zero bytes."""
    if P is None:
        return bytes(64)
    return int_to_le32(to_mont(P[0], Q_MOD)) + int_to_le32(to_mont(P[1], Q_MOD))


def g1_from_bytes(b: bytes):
    assert len(b) == 64
    if b == bytes(64):
        return None
    return (from_mont(le32_to_int(b[:32]), Q_MOD), from_mont(le32_to_int(b[32:]), Q_MOD))


def g2_to_bytes(P) -> bytes:
    """x.a || x.b || y.a || y.b, Montgomery LE; infinity = 128 zero bytes."""
    if P is None:
        return bytes(128)
    (xa, xb), (ya, yb) = P
    return b"".join(int_to_le32(to_mont(v, Q_MOD)) for v in (xa, xb, ya, yb))


def g2_from_bytes(b: bytes):
    assert len(b) == 128
    if b == bytes(128):
        return None
    v = [from_mont(le32_to_int(b[i * 32:(i + 1) * 32]), Q_MOD) for i in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))
