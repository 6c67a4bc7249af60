"""Synthetic BN254 workloads of BASELINE.json's configs: a zkey-shaped data
set with n = domainSize = nVars = 2^k, nPublic = 1, m = n-2 constraints, 2 non-zeros per row
in A and in B plus snarkjs's nPublic+1 extra A-rows  =>  nCoefs = 4m + 2.

Point tables are additive chains  T[i] = (k0 + i*kq) * G  (valid, distinct curve points with
KNOWN discrete logs), so every MSM result — and the final proof — can be checked in Fr alone
at full size.  The chain generator is injected (GPU product kernels for bench.py; the C
oracle for CPU-only tests): this module only defines the family and is pure numpy.

A random witness does not satisfy the R1CS; prove() never checks (src/groth16.cpp has no
satisfaction test) and its cost is identical.
"""
import random

import numpy as np

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
R2_MOD_R = 944936681149208446651664254269745548490766851729442924617792859073125903783   # R^2 mod r
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
_G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634),
           (8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531))   # EIP-197


def _mont(x):
    return ((x << 256) % Q_MOD).to_bytes(32, "little")


def g1_gen_bytes():
    """Generator (1, 2) of G1, affine Montgomery bytes."""
    return _mont(1) + _mont(2)


def g2_gen_bytes():
    (xa, xb), (ya, yb) = _G2_GEN
    return _mont(xa) + _mont(xb) + _mont(ya) + _mont(yb)


COEF_DTYPE = np.dtype([("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "u1", (32,))])   # 44 B packed (groth16.hpp:27-35)
assert COEF_DTYPE.itemsize == 44

_R_TOP = R_MOD >> 192      # top 64-bit limb of r


def random_fr_bytes(rng: np.random.Generator, n: int) -> np.ndarray:
    """n values uniform in [0, r) as [n,32] uint8 LE (rejection sampling on 254-bit candidates)."""
    out = np.empty((n, 4), dtype="<u8")
    filled = 0
    while filled < n:
        need = n - filled
        cand = rng.integers(0, 1 << 63, size=(int(need * 1.4) + 16, 4), dtype=np.uint64, endpoint=False)
        cand[:, :3] |= rng.integers(0, 2, size=(cand.shape[0], 3), dtype=np.uint64) << np.uint64(63)
        cand[:, 3] &= np.uint64((1 << 62) - 1)
        ok = cand[cand[:, 3] < np.uint64(_R_TOP)]        # strictly below the top limb: < r, drops a 2^-62 sliver
        take = min(need, ok.shape[0])
        out[filled:filled + take] = ok[:take]
        filled += take
    return out.view(np.uint8).reshape(n, 32)


def make_coefs(k: int, n_public: int, seed: int, n_vars: int = 0) -> np.ndarray:
    """Section-4 image (u32 count + packed records) as a flat uint8 array (signals drawn below n_vars, default the domain size)."""
    n = 1 << k
    m = n - n_public - 1
    rng = np.random.default_rng(0xC0EF0000 + k + seed)
    ncoefs = 4 * m + n_public + 1
    rec = np.zeros(ncoefs, dtype=COEF_DTYPE)
    rows = np.arange(m, dtype=np.uint32)
    cols = rng.integers(0, n_vars or n, size=(m, 4), dtype=np.uint32)
    vals = random_fr_bytes(rng, 4 * m).reshape(m, 4, 32)
    for j in range(4):                       # j = 0,1 -> matrix A ; 2,3 -> matrix B
        sl = rec[j * m:(j + 1) * m]
        sl["m"] = 0 if j < 2 else 1
        sl["c"] = rows
        sl["s"] = cols[:, j]
        sl["v"] = vals[:, j]
    extra = rec[4 * m:]
    extra["m"] = 0
    extra["c"] = m + np.arange(n_public + 1, dtype=np.uint32)
    extra["s"] = np.arange(n_public + 1, dtype=np.uint32)
    extra["v"] = np.frombuffer(R2_MOD_R.to_bytes(32, "little"), dtype=np.uint8)      # value 1, stored *R^2
    # interleave so records are not already row-sorted (the loader must sort)
    perm = rng.permutation(ncoefs)
    rec = rec[perm]
    img = np.empty(4 + ncoefs * 44, dtype=np.uint8)
    img[:4] = np.frombuffer(np.uint32(ncoefs).tobytes(), dtype=np.uint8)
    img[4:] = rec.view(np.uint8).reshape(-1)
    return img


def circuit_n_vars(k: int) -> int:
    """nVars of the circuit-shaped member of the family: three quarters of the domain and not a power of two
    (real circom keys never have nVars = domainSize: the domain is the next power of two above the constraint count)."""
    return 3 * (1 << k) // 4 + 5


def make_witness(k: int, seed: int = 0, kind: str = "uniform", n_vars: int = 0) -> np.ndarray:
    """nVars x 32 B standard form, w[0] = 1 (n_vars = 0: the domain size 2^k).
    kind = "uniform"   : every other entry uniform in [0, r)  — the worst case for the MSMs
    kind = "realistic" : SURVEY §8d secondary line: 80 % of entries in {0, 1}, 15 % < 2^32, 5 % full-size
                         (circom witnesses are dominated by booleans and small values)."""
    n = n_vars or (1 << k)
    rng = np.random.default_rng(0x5EED0000 + k + 1000003 * seed)
    w = random_fr_bytes(rng, n)
    if kind == "realistic":
        cls = rng.random(n)
        small = cls < 0.80
        mid = (cls >= 0.80) & (cls < 0.95)
        w[small] = 0
        w[small, 0] = rng.integers(0, 2, size=int(small.sum()), dtype=np.uint8)
        w[mid, 4:] = 0
    elif kind != "uniform":
        raise ValueError("unknown witness kind %r" % kind)
    w[0] = 0
    w[0, 0] = 1
    return w.reshape(-1)


def workload(k, chain_g1, chain_g2, g1_mul, g2_mul, g1_gen, g2_gen, n_public=1, seed=0, shape="dense"):
    """-> dict usable as a zkey view: numpy uint8 arrays for every section + the dlog table.

    chain_gX(n, P0_bytes, Q_bytes) -> uint8 array; gX_mul(P_bytes, k_int) -> bytes.
    shape = "dense"   : BASELINE's worst case — nVars = domainSize, every table row a distinct point
    shape = "circuit" : what a real circom key looks like to the prover — nVars = 3/4 of the domain + 5 (not a power of
                        two), 3 public signals, and ~30 % of the rows of A and of B1 / B2 (the SAME rows in both: B1_i and B2_i
                        are b_i(tau) in the two groups) all-zero = the point at infinity: wires that never occur in that
                        matrix.  `zero_rows` holds the masks; the discrete logs of the other rows are unchanged, so every
                        MSM result stays checkable in Fr (expected_msm_dlogs).
    """
    if shape == "circuit":
        return _circuit_workload(k, chain_g1, chain_g2, g1_mul, g2_mul, g1_gen, g2_gen, seed)
    if shape != "dense":
        raise ValueError("unknown workload shape %r" % shape)
    n = 1 << k
    prng = random.Random(0xD106 + 31 * k + seed)
    dl = {name: (prng.randrange(1, R_MOD), prng.randrange(1, R_MOD)) for name in ("A", "B", "C", "H")}
    vk = {name: prng.randrange(1, R_MOD) for name in ("alpha", "beta", "delta")}

    def tab1(name, cnt):
        k0, kq = dl[name]
        return chain_g1(cnt, g1_mul(g1_gen, k0), g1_mul(g1_gen, kq))

    k0, kq = dl["B"]
    wl = {
        "k": k, "nVars": n, "nPublic": n_public, "domainSize": n, "nCoefs": 4 * (n - n_public - 1) + n_public + 1,
        "coefs": make_coefs(k, n_public, seed),
        "pointsA": tab1("A", n), "pointsB1": tab1("B", n),
        "pointsB2": chain_g2(n, g2_mul(g2_gen, k0), g2_mul(g2_gen, kq)),
        "pointsC": tab1("C", n - n_public - 1), "pointsH": tab1("H", n),
        "vk_alpha1": np.frombuffer(g1_mul(g1_gen, vk["alpha"]), dtype=np.uint8),
        "vk_beta1": np.frombuffer(g1_mul(g1_gen, vk["beta"]), dtype=np.uint8),
        "vk_beta2": np.frombuffer(g2_mul(g2_gen, vk["beta"]), dtype=np.uint8),
        "vk_delta1": np.frombuffer(g1_mul(g1_gen, vk["delta"]), dtype=np.uint8),
        "vk_delta2": np.frombuffer(g2_mul(g2_gen, vk["delta"]), dtype=np.uint8),
        "dlogs": dl, "vk_dlogs": vk,
    }
    return wl


def _circuit_workload(k, chain_g1, chain_g2, g1_mul, g2_mul, g1_gen, g2_gen, seed):
    n_public = 3
    nv = circuit_n_vars(k)
    wl = workload(k, chain_g1, chain_g2, g1_mul, g2_mul, g1_gen, g2_gen, n_public=n_public, seed=seed)     # dense tables of 2^k rows to cut from
    rng = np.random.default_rng(0xC1AC0000 + 7 * k + seed)
    za, zb = rng.random(nv) < 0.30, rng.random(nv) < 0.30
    za[0] = zb[0] = False
    for name, rows, mask in (("pointsA", 64, za), ("pointsB1", 64, zb), ("pointsB2", 128, zb)):
        t = np.array(wl[name][:nv * rows]).reshape(nv, rows)
        t[mask] = 0                                   # all-zero = infinity
        wl[name] = t.reshape(-1)
    wl["pointsC"] = np.ascontiguousarray(wl["pointsC"][:(nv - n_public - 1) * 64])
    wl["nVars"] = nv
    wl["coefs"] = make_coefs(k, n_public, seed, n_vars=nv)
    wl["zero_rows"] = {"A": za, "B": zb}
    wl["shape"] = "circuit"
    return wl


def weighted_sums(vals: np.ndarray):
    """(sum v_i, sum i*v_i) over 32-byte LE integers, exact (16-bit limbs, chunked uint64 dots)."""
    v = np.ascontiguousarray(vals).view(np.uint8).reshape(-1, 32).view("<u2").astype(np.uint64)    # [n,16]
    n = v.shape[0]
    s0 = 0
    s1 = 0
    chunk = 1 << 18
    for lo in range(0, n, chunk):
        blk = v[lo:lo + chunk]
        idx = np.arange(lo, lo + blk.shape[0], dtype=np.uint64)
        col0 = blk.sum(axis=0)                           # < 2^16 * 2^18
        col1 = (blk * idx[:, None]).sum(axis=0)          # < 2^16 * 2^28 * 2^18 = 2^62
        for j in range(16):
            s0 += int(col0[j]) << (16 * j)
            s1 += int(col1[j]) << (16 * j)
    return s0, s1


def expected_msm_dlogs(wl, witness: np.ndarray, h: np.ndarray):
    """Discrete logs (mod r) of the five MSM results for a workload() data set."""
    npub = wl["nPublic"]
    out = {}
    for name, key in (("pi_a", "A"), ("pib1", "B")):
        wm = np.ascontiguousarray(witness).reshape(-1, 32)
        if "zero_rows" in wl:                         # rows at infinity contribute nothing
            wm = wm.copy()
            wm[wl["zero_rows"][key]] = 0
        sw, swi = weighted_sums(wm)
        k0, kq = wl["dlogs"][key]
        out[name] = (k0 * sw + kq * swi) % R_MOD
    out["pi_b"] = out["pib1"]
    wc = np.ascontiguousarray(witness).reshape(-1, 32)[npub + 1:]
    sc, sci = weighted_sums(wc)
    k0, kq = wl["dlogs"]["C"]
    out["pi_c"] = (k0 * sc + kq * sci) % R_MOD
    sh, shi = weighted_sums(h)
    k0, kq = wl["dlogs"]["H"]
    out["pih"] = (k0 * sh + kq * shi) % R_MOD
    return out


def expected_proof_dlogs(wl, msm_dlogs, r, s):
    """Discrete logs of (A, B, C) per src/groth16.cpp:222-246."""
    vk = wl["vk_dlogs"]
    a = (msm_dlogs["pi_a"] + vk["alpha"] + r * vk["delta"]) % R_MOD
    b = (msm_dlogs["pi_b"] + vk["beta"] + s * vk["delta"]) % R_MOD
    b1 = (msm_dlogs["pib1"] + vk["beta"] + s * vk["delta"]) % R_MOD
    c = (msm_dlogs["pi_c"] + msm_dlogs["pih"] + s * a + r * b1 - (r * s % R_MOD) * vk["delta"]) % R_MOD
    return a, b, c
