"""Groth16 prover restatement + snarkjs file formats — TEST INFRASTRUCTURE ONLY.

ORACLE for the hot path `Groth16::Prover::prove()` (reference
`src/groth16.cpp:48-254`), the file formats it consumes
(`src/binfile_utils.cpp:14-62`, `src/zkey_utils.cpp:17-52`,
`src/wtns_utils.cpp:12-25`) and the JSON it emits (`src/groth16.cpp:268-301`,
`src/main_prover.cpp:77-93`).  Pure Python big-int: small cases only.

PARITY UNPINNED (see oracle/bn254.py header): the reference has no tests and
cannot be built here; this oracle is pinned by the pairing-free trapdoor check
`trapdoor_check()` on keys generated from known toxic waste, which is
independent of the prover's NTT/MSM code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
"""
import struct

from . import bn254 as bn
from .bn254 import R_MOD, Q_MOD, G1, G2


# ----------------------------------------------------------------- binfile container (binfile_utils.cpp:14-62)
def write_binfile(magic: bytes, version: int, sections) -> bytes:
    """sections: list of (type:int, payload:bytes) in file order."""
    out = [magic, struct.pack("<II", version, len(sections))]
    for stype, payload in sections:
        out.append(struct.pack("<IQ", stype, len(payload)))
        out.append(payload)
    return b"".join(out)


def read_binfile(data: bytes, magic: bytes, max_version: int):
    """Returns {type: [payload,...]} keeping duplicates in order (binfile_utils.cpp:49-60)."""
    if data[:4] != magic:
        raise ValueError("Invalid file type. It should be %s and it us %s" % (magic.decode(), data[:4].decode("latin1")))
    version, nsec = struct.unpack_from("<II", data, 4)
    if version > max_version:
        raise ValueError("Invalid version. It should be <=%d and it us %d" % (max_version, version))
    pos = 12
    secs = {}
    for _ in range(nsec):
        stype, ssize = struct.unpack_from("<IQ", data, pos)
        pos += 12
        secs.setdefault(stype, []).append(data[pos:pos + ssize])
        pos += ssize
    return secs


# ----------------------------------------------------------------- wtns (wtns_utils.cpp:12-25)
def write_wtns(witness) -> bytes:
    sec1 = struct.pack("<I", 32) + bn.int_to_le32(R_MOD) + struct.pack("<I", len(witness))
    sec2 = b"".join(bn.int_to_le32(w) for w in witness)
    return write_binfile(b"wtns", 2, [(1, sec1), (2, sec2)])


def read_wtns(data: bytes):
    secs = read_binfile(data, b"wtns", 2)
    s1 = secs[1][0]
    n8 = struct.unpack_from("<I", s1, 0)[0]
    prime = int.from_bytes(s1[4:4 + n8], "little")
    nvars = struct.unpack_from("<I", s1, 4 + n8)[0]
    s2 = secs[2][0]
    wit = [int.from_bytes(s2[i * n8:(i + 1) * n8], "little") for i in range(nvars)]
    return {"n8": n8, "prime": prime, "nVars": nvars, "witness": wit}


# ----------------------------------------------------------------- zkey (zkey_utils.cpp:17-52)
class ZKey:
    """Decoded zkey: plain (non-Montgomery) integers / affine points."""

    def __init__(self):
        self.nVars = self.nPublic = self.domainSize = 0
        self.alpha1 = self.beta1 = self.delta1 = None
        self.beta2 = self.gamma2 = self.delta2 = None
        self.IC = []
        self.coefs = []      # list of (m, c, s, value) with value a plain Fr integer
        self.A = []
        self.B1 = []
        self.B2 = []
        self.C = []
        self.H = []


def write_zkey(zk: ZKey) -> bytes:
    sec1 = struct.pack("<I", 1)
    sec2 = (struct.pack("<I", 32) + bn.int_to_le32(Q_MOD) + struct.pack("<I", 32) + bn.int_to_le32(R_MOD)
            + struct.pack("<III", zk.nVars, zk.nPublic, zk.domainSize)
            + bn.g1_to_bytes(zk.alpha1) + bn.g1_to_bytes(zk.beta1) + bn.g2_to_bytes(zk.beta2)
            + bn.g2_to_bytes(zk.gamma2) + bn.g1_to_bytes(zk.delta1) + bn.g2_to_bytes(zk.delta2))
    sec3 = b"".join(bn.g1_to_bytes(p) for p in zk.IC)
    # coef value stored as value*R^2 mod r: Montgomery form of value*R
    r2 = (bn.MONT_R * bn.MONT_R) % R_MOD
    sec4 = struct.pack("<I", len(zk.coefs)) + b"".join(
        struct.pack("<III", m, c, s) + bn.int_to_le32((v * r2) % R_MOD) for (m, c, s, v) in zk.coefs)
    sec5 = b"".join(bn.g1_to_bytes(p) for p in zk.A)
    sec6 = b"".join(bn.g1_to_bytes(p) for p in zk.B1)
    sec7 = b"".join(bn.g2_to_bytes(p) for p in zk.B2)
    sec8 = b"".join(bn.g1_to_bytes(p) for p in zk.C)
    sec9 = b"".join(bn.g1_to_bytes(p) for p in zk.H)
    sec10 = bytes(64) + struct.pack("<I", 0)  # csHash + 0 contributions (not read by the prover)
    return write_binfile(b"zkey", 1, [(1, sec1), (2, sec2), (3, sec3), (4, sec4), (5, sec5),
                                      (6, sec6), (7, sec7), (8, sec8), (9, sec9), (10, sec10)])


def read_zkey(data: bytes) -> ZKey:
    secs = read_binfile(data, b"zkey", 1)
    if struct.unpack_from("<I", secs[1][0], 0)[0] != 1:
        raise ValueError("zkey file is not groth16")
    s2 = secs[2][0]
    zk = ZKey()
    pos = 0
    n8q = struct.unpack_from("<I", s2, pos)[0]; pos += 4
    zk.q = int.from_bytes(s2[pos:pos + n8q], "little"); pos += n8q
    n8r = struct.unpack_from("<I", s2, pos)[0]; pos += 4
    zk.r = int.from_bytes(s2[pos:pos + n8r], "little"); pos += n8r
    zk.nVars, zk.nPublic, zk.domainSize = struct.unpack_from("<III", s2, pos); pos += 12
    zk.alpha1 = bn.g1_from_bytes(s2[pos:pos + 64]); pos += 64
    zk.beta1 = bn.g1_from_bytes(s2[pos:pos + 64]); pos += 64
    zk.beta2 = bn.g2_from_bytes(s2[pos:pos + 128]); pos += 128
    zk.gamma2 = bn.g2_from_bytes(s2[pos:pos + 128]); pos += 128
    zk.delta1 = bn.g1_from_bytes(s2[pos:pos + 64]); pos += 64
    zk.delta2 = bn.g2_from_bytes(s2[pos:pos + 128]); pos += 128
    s4 = secs[4][0]
    ncoefs = len(s4) // (12 + n8r)      # zkey_utils.cpp:49
    r2inv = pow(bn.MONT_R * bn.MONT_R, -1, R_MOD)
    for i in range(ncoefs):
        off = 4 + i * 44                 # groth16.cpp:38 skips the u32 count
        m, c, s = struct.unpack_from("<III", s4, off)
        v = int.from_bytes(s4[off + 12:off + 44], "little")
        zk.coefs.append((m, c, s, (v * r2inv) % R_MOD))
    def g1s(b):
        return [bn.g1_from_bytes(b[i:i + 64]) for i in range(0, len(b), 64)]
    zk.IC = g1s(secs[3][0]) if 3 in secs else []
    zk.A = g1s(secs[5][0])
    zk.B1 = g1s(secs[6][0])
    zk.B2 = [bn.g2_from_bytes(secs[7][0][i:i + 128]) for i in range(0, len(secs[7][0]), 128)]
    zk.C = g1s(secs[8][0])
    zk.H = g1s(secs[9][0])
    return zk


# ----------------------------------------------------------------- R1CS + trapdoor setup
class R1CS:
    """Rows are dicts {signal: value}. Signal 0 is the constant 1; 1..nPublic are public."""

    def __init__(self, nVars, nPublic, A, B, C):
        self.nVars, self.nPublic, self.A, self.B, self.C = nVars, nPublic, A, B, C

    def is_satisfied(self, w):
        dot = lambda row: sum(v * w[s] for s, v in row.items()) % R_MOD
        return all((dot(a) * dot(b) - dot(c)) % R_MOD == 0 for a, b, c in zip(self.A, self.B, self.C))


def _lagrange_at(tau, n):
    """L_j(tau) for the n-th roots of unity, j<n."""
    logn = n.bit_length() - 1
    w = bn.fr_root(logn)
    zt = (pow(tau, n, R_MOD) - 1) % R_MOD
    ninv = pow(n, -1, R_MOD)
    out = []
    wj = 1
    for _ in range(n):
        out.append(zt * wj % R_MOD * ninv % R_MOD * pow((tau - wj) % R_MOD, -1, R_MOD) % R_MOD)
        wj = wj * w % R_MOD
    return out


def setup_scalars(r1cs: R1CS, toxic, domain_size=None):
    """The Fr half of a trapdoor Groth16 setup in the snarkjs zkey layout: every
    table entry's discrete logarithm, no curve arithmetic.  snarkjs appends nPublic+1 rows
    A[m+i][i] = 1.  Returns the dict `trap` used by `trapdoor_check`, plus
    "coefs" (m, c, s, value), "IC" and "C" scalars."""
    tau, alpha, beta, gamma, delta = toxic
    m = len(r1cs.A)
    need = m + r1cs.nPublic + 1
    n = domain_size or (1 << max(1, (need - 1).bit_length()))
    assert n >= need and n & (n - 1) == 0
    A_rows = [dict(row) for row in r1cs.A] + [{i: 1} for i in range(r1cs.nPublic + 1)]
    B_rows = [dict(row) for row in r1cs.B] + [{} for _ in range(r1cs.nPublic + 1)]
    C_rows = [dict(row) for row in r1cs.C] + [{} for _ in range(r1cs.nPublic + 1)]
    L = _lagrange_at(tau, n)
    At = [0] * r1cs.nVars
    Bt = [0] * r1cs.nVars
    Ct = [0] * r1cs.nVars
    coefs = []
    for row, (ra, rb, rc) in enumerate(zip(A_rows, B_rows, C_rows)):
        for s, v in ra.items():
            At[s] = (At[s] + v * L[row]) % R_MOD
            coefs.append((0, row, s, v % R_MOD))
        for s, v in rb.items():
            Bt[s] = (Bt[s] + v * L[row]) % R_MOD
            coefs.append((1, row, s, v % R_MOD))
        for s, v in rc.items():
            Ct[s] = (Ct[s] + v * L[row]) % R_MOD
    ginv = pow(gamma, -1, R_MOD)
    dinv = pow(delta, -1, R_MOD)
    K = [(beta * At[i] + alpha * Bt[i] + Ct[i]) % R_MOD for i in range(r1cs.nVars)]
    # H[i] = L^(2n)_{2i+1}(tau)/delta * G1  
    w2n = bn.fr_root(n.bit_length())          # primitive 2n-th root
    z2 = (pow(tau, 2 * n, R_MOD) - 1) % R_MOD
    inv2n = pow(2 * n, -1, R_MOD)
    Hs = []
    for i in range(n):
        wj = pow(w2n, 2 * i + 1, R_MOD)
        Hs.append(z2 * wj % R_MOD * inv2n % R_MOD * pow((tau - wj) % R_MOD, -1, R_MOD) % R_MOD * dinv % R_MOD)
    return {"At": At, "Bt": Bt, "Ct": Ct, "K": K, "Hs": Hs, "toxic": toxic, "n": n, "coefs": coefs,
            "IC": [K[i] * ginv % R_MOD for i in range(r1cs.nPublic + 1)],
            "C": [K[i] * dinv % R_MOD for i in range(r1cs.nPublic + 1, r1cs.nVars)]}


def setup(r1cs: R1CS, toxic, domain_size=None):
    """Trapdoor Groth16 setup: `setup_scalars` + fixed-base multiplications of the generators.

    toxic = (tau, alpha, beta, gamma, delta).  Returns (ZKey, trap) where trap holds
    the per-signal polynomial evaluations at tau used by `trapdoor_check`."""
    tau, alpha, beta, gamma, delta = toxic
    trap = setup_scalars(r1cs, toxic, domain_size)
    t1 = G1.fixed_base_table(G1.gen)
    t2 = G2.fixed_base_table(G2.gen)
    g1 = lambda k: G1.mul_fixed(t1, k % R_MOD)
    g2 = lambda k: G2.mul_fixed(t2, k % R_MOD)
    zk = ZKey()
    zk.nVars, zk.nPublic, zk.domainSize = r1cs.nVars, r1cs.nPublic, trap["n"]
    zk.alpha1, zk.beta1, zk.delta1 = g1(alpha), g1(beta), g1(delta)
    zk.beta2, zk.gamma2, zk.delta2 = g2(beta), g2(gamma), g2(delta)
    zk.coefs = trap["coefs"]
    zk.A = [g1(x) for x in trap["At"]]
    zk.B1 = [g1(x) for x in trap["Bt"]]
    zk.B2 = [g2(x) for x in trap["Bt"]]
    zk.IC = [g1(x) for x in trap["IC"]]
    zk.C = [g1(x) for x in trap["C"]]
    zk.H = [g1(x) for x in trap["Hs"]]
    return zk, trap


# ----------------------------------------------------------------- the prover (groth16.cpp:48-254)
def compute_h(zk: ZKey, w):
    """Steps 1-5 of prove(): returns standard-form h[i] (groth16.cpp:52-163)."""
    n = zk.domainSize
    a = [0] * n
    b = [0] * n
    for (m, c, s, v) in zk.coefs:                       # groth16.cpp:66-84
        if m == 0:
            a[c] = (a[c] + w[s] * v) % R_MOD
        else:
            b[c] = (b[c] + w[s] * v) % R_MOD
    cc = [(x * y) % R_MOD for x, y in zip(a, b)]         # groth16.cpp:89-96
    w2n = bn.fr_root(n.bit_length())
    def coset(x):                                        # groth16.cpp:102-115
        co = bn.ntt(x, inverse=True)
        sh = [(v * pow(w2n, i, R_MOD)) % R_MOD for i, v in enumerate(co)]
        return bn.ntt(sh)
    ae, be, ce = coset(a), coset(b), coset(cc)
    return [(x * y - z) % R_MOD for x, y, z in zip(ae, be, ce)], (a, b, cc)   # groth16.cpp:158-163


def prove(zk: ZKey, w, r: int, s: int):
    """Returns (A, B, C) affine plain-int points for fixed (r, s) (groth16.cpp:171-253)."""
    h, _ = compute_h(zk, w)
    pih = G1.msm(zk.H, h)                                                  # :173
    pi_a = G1.msm(zk.A, w)                                                 # :183
    pib1 = G1.msm(zk.B1, w)                                                # :190
    pi_b = G2.msm(zk.B2, w)                                                # :197
    pi_c = G1.msm(zk.C, w[zk.nPublic + 1:])                                # :204
    pi_a = G1.add(G1.add(pi_a, zk.alpha1), G1.mul(zk.delta1, r))           # :222-224
    pi_b = G2.add(G2.add(pi_b, zk.beta2), G2.mul(zk.delta2, s))            # :226-228
    pib1 = G1.add(G1.add(pib1, zk.beta1), G1.mul(zk.delta1, s))            # :230-232
    pi_c = G1.add(pi_c, pih)                                               # :234
    pi_c = G1.add(pi_c, G1.mul(pi_a, s))                                   # :236-237
    pi_c = G1.add(pi_c, G1.mul(pib1, r))                                   # :239-240
    rs = (r * s) % R_MOD                                                   # :242-243
    pi_c = G1.sub(pi_c, G1.mul(zk.delta1, rs))                             # :245-246
    return pi_a, pi_b, pi_c


def trapdoor_check(trap, nPublic, w, r, s, proof) -> bool:
    """Pairing-free check of a proof against the toxic waste."""
    tau, alpha, beta, gamma, delta = trap["toxic"]
    n = trap["n"]
    At, Bt, Ct, K = trap["At"], trap["Bt"], trap["Ct"], trap["K"]
    dot = lambda v, lo=0: sum(x * wi for x, wi in zip(v[lo:], w[lo:])) % R_MOD
    a = (alpha + dot(At) + r * delta) % R_MOD
    b = (beta + dot(Bt) + s * delta) % R_MOD
    # H(tau) Z(tau) = A(tau) B(tau) - C(tau) where C := interpolation of (A.w)o(B.w)
    #   for a satisfying witness equals sum w_i C_i(tau)
    hz = (dot(At) * dot(Bt) - dot(Ct)) % R_MOD
    dinv = pow(delta, -1, R_MOD)
    c = ((dot(K, nPublic + 1) + hz) * dinv + s * a + r * b - r * s % R_MOD * delta) % R_MOD
    A, B, C = proof
    return A == G1.mul(G1.gen, a) and B == G2.mul(G2.gen, b) and C == G1.mul(G1.gen, c)


# ----------------------------------------------------------------- JSON (groth16.cpp:268-301)
def proof_to_json(proof) -> str:
    A, B, C = proof
    def g1j(P):
        # an infinity result cannot be printed by the reference either; emit zeros
        x, y = P if P is not None else (0, 0)
        return '["%d","%d","1"]' % (x, y)
    (xa, xb), (ya, yb) = B if B is not None else ((0, 0), (0, 0))
    return ('{"pi_a":%s,"pi_b":[["%d","%d"],["%d","%d"],["1","0"]],"pi_c":%s,"protocol":"groth16"}'
            % (g1j(A), xa, xb, ya, yb, g1j(C)))


def public_to_json(w, nPublic) -> str:
    if nPublic == 0:
        return "null"                                    # main_prover.cpp:85-92 quirk Q7
    return "[" + ",".join('"%d"' % w[i] for i in range(1, nPublic + 1)) + "]"


def proof_to_bytes(proof) -> bytes:
    """Same bytes as Proof<Engine>{A,B,C}: affine Montgomery LE (groth16.hpp:13-24)."""
    A, B, C = proof
    return bn.g1_to_bytes(A) + bn.g2_to_bytes(B) + bn.g1_to_bytes(C)


# ----------------------------------------------------------------- fixture circuits
def multiplier2_r1cs():
    """circom Multiplier2: c = a*b; signals [1, c, a, b] (nVars 4, nPublic 1)."""
    return R1CS(4, 1, [{2: 1}], [{3: 1}], [{1: 1}])


def random_r1cs(rng, n_constraints, n_public, extra_vars=2):
    """Random satisfiable R1CS + witness: each constraint defines a new signal
    x_new = (lin_a)*(lin_b) over earlier signals."""
    n_inputs = n_public + extra_vars
    nVars = 1 + n_inputs + n_constraints
    w = [1] + [rng.randrange(R_MOD) for _ in range(n_inputs)]
    A, B, C = [], [], []
    for k in range(n_constraints):
        avail = len(w)
        ra = {rng.randrange(avail): rng.randrange(1, R_MOD) for _ in range(2)}
        rb = {rng.randrange(avail): rng.randrange(1, R_MOD) for _ in range(2)}
        if k % 5 == 0:
            ra[rng.randrange(avail)] = 1          # small coefficients as in real circuits
        va = sum(v * w[s] for s, v in ra.items()) % R_MOD
        vb = sum(v * w[s] for s, v in rb.items()) % R_MOD
        w.append(va * vb % R_MOD)
        A.append(ra)
        B.append(rb)
        C.append({avail: 1})
    # public signals must be 1..nPublic: they are the first inputs, already in place
    return R1CS(nVars, n_public, A, B, C), w
