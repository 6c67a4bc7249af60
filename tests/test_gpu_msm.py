"""GPU parity: Pippenger MSM over G1/G2 vs golden KATs (edge cases included) and vs
size-independent identities at larger n.  Bit-exact affine Montgomery bytes."""
import random

import numpy as np
import pytest

from conftest import golden_bytes, golden_json
from oracle import bn254 as bn
from oracle.bn254 import G1, G2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["g1_n1", "g1_n2", "g1_n3", "g1_n17", "g1_n1000", "g1_cancel"])
def test_msm_g1_kat(zk, name):
    want = golden_json("kat_msm.json")[name]
    got = zk.msm_g1(golden_bytes("msm_%s_bases.bin" % name), golden_bytes("msm_%s_scalars.bin" % name))
    assert got.hex() == want


@pytest.mark.parametrize("name", ["g2_n1", "g2_n2", "g2_n17", "g2_n300"])
def test_msm_g2_kat(zk, name):
    want = golden_json("kat_msm.json")[name]
    got = zk.msm_g2(golden_bytes("msm_%s_bases.bin" % name), golden_bytes("msm_%s_scalars.bin" % name))
    assert got.hex() == want


def test_msm_empty(zk):
    assert zk.msm_g1(b"", b"") == bytes(64)
    assert zk.msm_g2(b"", b"") == bytes(128)


def _chain_points(n, seed):
    """P_i = (a + i*d) * G computed incrementally with the oracle (cheap: one add per point)."""
    rng = random.Random(seed)
    a, d = rng.randrange(1, bn.R_MOD), rng.randrange(1, bn.R_MOD)
    P, D = G1.mul(G1.gen, a), G1.mul(G1.gen, d)
    pts = []
    for _ in range(n):
        pts.append(P)
        P = G1.add(P, D)
    return pts, a, d


@pytest.mark.parametrize("n", [5000, 40000])
def test_msm_g1_known_dlog(zk, n):
    """sum k_i * (a + i d) G = (sum k_i (a + i d)) G  — independent of any MSM code."""
    pts, a, d = _chain_points(n, n)
    rng = random.Random(n + 1)
    sc = [rng.randrange(bn.R_MOD) for _ in range(n)]
    # skewed tail: many zeros / ones / small values like a real witness
    for i in range(0, n, 3):
        sc[i] = rng.choice([0, 1, 1, 2, rng.randrange(1 << 32)])
    total = sum(k * (a + i * d) for i, k in enumerate(sc)) % bn.R_MOD
    want = bn.g1_to_bytes(G1.mul(G1.gen, total))
    got = zk.msm_g1(b"".join(bn.g1_to_bytes(P) for P in pts), b"".join(bn.int_to_le32(k) for k in sc))
    assert got == want


def test_msm_g1_all_same_point_all_ones(zk):
    """n copies of P with scalar 1: every add in the single bucket hits P+P / 2P+P paths."""
    n = 1000
    P = G1.mul(G1.gen, 987654321)
    got = zk.msm_g1(bn.g1_to_bytes(P) * n, bn.int_to_le32(1) * n)
    assert got == bn.g1_to_bytes(G1.mul(P, n))


def test_msm_g2_known_dlog(zk):
    n = 3000
    rng = random.Random(99)
    a, d = rng.randrange(1, bn.R_MOD), rng.randrange(1, bn.R_MOD)
    P, D = G2.mul(G2.gen, a), G2.mul(G2.gen, d)
    pts = []
    for _ in range(n):
        pts.append(P)
        P = G2.add(P, D)
    sc = [rng.randrange(bn.R_MOD) for _ in range(n)]
    total = sum(k * (a + i * d) for i, k in enumerate(sc)) % bn.R_MOD
    got = zk.msm_g2(b"".join(bn.g2_to_bytes(P) for P in pts), b"".join(bn.int_to_le32(k) for k in sc))
    assert got == bn.g2_to_bytes(G2.mul(G2.gen, total))


def test_msm_g1_linearity_large(zk):
    """2^18 points (beyond the oracle): MSM(k) + MSM(k') == MSM(k + k') on repeated bases."""
    n = 1 << 18
    base_pts, _, _ = _chain_points(256, 4242)
    bases = b"".join(bn.g1_to_bytes(P) for P in base_pts) * (n // 256)
    rng = random.Random(3)
    raw = np.frombuffer(rng.randbytes(32 * n), dtype=np.uint8).reshape(n, 32).copy()
    raw[:, 31] &= 0x0f
    k1 = raw.tobytes()
    raw2 = np.frombuffer(rng.randbytes(32 * n), dtype=np.uint8).reshape(n, 32).copy()
    raw2[:, 31] &= 0x0f
    k2 = raw2.tobytes()
    a1 = np.frombuffer(k1, dtype="<u8").reshape(n, 4)
    # k1 + k2 as 256-bit integers (no overflow: both < 2^252)
    s = [(int.from_bytes(k1[i * 32:(i + 1) * 32], "little") + int.from_bytes(k2[i * 32:(i + 1) * 32], "little")) for i in range(0, n, 1)] if n <= (1 << 14) else None
    if s is None:
        v1 = np.frombuffer(k1, dtype="<u4").reshape(n, 8).astype(np.uint64)
        v2 = np.frombuffer(k2, dtype="<u4").reshape(n, 8).astype(np.uint64)
        acc = v1 + v2
        for j in range(7):
            acc[:, j + 1] += acc[:, j] >> 32
            acc[:, j] &= 0xffffffff
        ksum = acc.astype("<u4").tobytes()
    else:
        ksum = b"".join(bn.int_to_le32(x) for x in s)
    del a1
    r1 = bn.g1_from_bytes(zk.msm_g1(bases, k1))
    r2 = bn.g1_from_bytes(zk.msm_g1(bases, k2))
    r3 = bn.g1_from_bytes(zk.msm_g1(bases, ksum))
    assert G1.add(r1, r2) == r3
    assert G1.is_on_curve(r3) and r3 is not None


def test_eip196_public_vectors_on_the_gpu(zk):
    """zk_msm_g1 / zk_fixed_base_g1 / zk_g1_mul against the public EIP-196 precompile vectors (third-party
    numbers, see tests/test_oracle_py.py): P + Q as a two-term MSM, k*B as a one-term MSM."""
    from test_oracle_py import EIP196_2G, EIP196_ADD, EIP196_MUL
    from oracle import bn254 as bn
    one = bn.int_to_le32(1)
    p, q, s = EIP196_ADD
    assert zk.msm_g1(bn.g1_to_bytes(p) + bn.g1_to_bytes(q), one + one) == bn.g1_to_bytes(s)
    b, k, r = EIP196_MUL
    assert zk.msm_g1(bn.g1_to_bytes(b), bn.int_to_le32(k)) == bn.g1_to_bytes(r)
    assert zk.fixed_base_g1(bn.g1_to_bytes(b), [k, 2]).tobytes() == bn.g1_to_bytes(r) + bn.g1_to_bytes(bn.G1.dbl(b))
    g1 = bn.g1_to_bytes(bn.G1.gen)
    assert zk.msm_g1(g1 + g1, one + one) == bn.g1_to_bytes(EIP196_2G)          # P = Q: the doubling branch of the mixed add
    assert zk.g1_mul(g1, 2) == bn.g1_to_bytes(EIP196_2G)


def test_msm_scalars_at_and_above_the_group_order(zk):
    """Scalars are raw 256-bit integers in the reference (multiMulByScalar takes bytes): r, r + 1, 2r + 5, 2^256 - 1 act as
    their residues mod r, r - 1 as -1 — the digit recoding brings any 256-bit value below r first."""
    r = bn.R_MOD
    ks = [r, r + 1, r - 1, 2 * r + 5, (1 << 256) - 1, 5 * r + 123456789, 0, 1]
    pts, _, _ = _chain_points(len(ks), 777)
    want = None
    for P, k in zip(pts, ks):
        want = G1.add(want, G1.mul(P, k % r))
    got = zk.msm_g1(b"".join(bn.g1_to_bytes(P) for P in pts), b"".join(int(k).to_bytes(32, "little") for k in ks))
    assert got == bn.g1_to_bytes(want)
    # the same scalars over a table long enough for the sorted path (every kernel of the pipeline), against their residues
    n = 5000
    pts, _, _ = _chain_points(n, 4321)
    bases = b"".join(bn.g1_to_bytes(P) for P in pts)
    rng = random.Random(11)
    big = [rng.choice(ks[:6]) if i % 3 == 0 else rng.randrange(r, 1 << 256) for i in range(n)]
    assert zk.msm_g1(bases, b"".join(k.to_bytes(32, "little") for k in big)) == zk.msm_g1(bases, b"".join((k % r).to_bytes(32, "little") for k in big))
