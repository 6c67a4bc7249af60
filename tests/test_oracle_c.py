"""CPU (no GPU): pins the C restatement (oracle/c/zk_oracle.c) against the Python big-int
oracle's committed golden vectors — field KATs, NTT, MSM edge cases, full proofs."""
import random

import pytest

from conftest import CIRCUITS, golden_bytes, golden_json
from oracle import bn254 as bn, c_oracle as co
from oracle.bn254 import G1, G2

le = bn.int_to_le32


def pack(vals):
    return b"".join(le(int(v)) for v in vals)


def unpack(b):
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_mont_mul_kat(field):
    k = golden_json("kat_field.json")[field]
    fn = co.fr_mul_vec if field == "fr" else co.fq_mul_vec
    assert unpack(fn(pack(k["a"]), pack(k["b"]))) == [int(x) for x in k["mont_mul"]]


@pytest.mark.parametrize("n", [2, 8, 64, 2048, 4096])
def test_fft(n):
    if n <= 64:
        k = golden_json("kat_ntt.json")[str(n)]
        x = pack(k["x"])
        assert unpack(co.fr_fft(x)) == [int(v) for v in k["fft"]]
        assert unpack(co.fr_fft(x, inverse=True)) == [int(v) for v in k["ifft"]]
    else:
        x = golden_bytes("ntt_x_%d.bin" % n)
        assert co.fr_fft(x) == golden_bytes("ntt_fft_%d.bin" % n)
        assert co.fr_fft(x, inverse=True) == golden_bytes("ntt_ifft_%d.bin" % n)


@pytest.mark.parametrize("name", ["g1_n1", "g1_n2", "g1_n3", "g1_n17", "g1_n1000", "g1_cancel"])
def test_msm_g1(name):
    got = co.msm_g1(golden_bytes("msm_%s_bases.bin" % name), golden_bytes("msm_%s_scalars.bin" % name))
    assert got.hex() == golden_json("kat_msm.json")[name]


@pytest.mark.parametrize("name", ["g2_n1", "g2_n2", "g2_n17", "g2_n300"])
def test_msm_g2(name):
    got = co.msm_g2(golden_bytes("msm_%s_bases.bin" % name), golden_bytes("msm_%s_scalars.bin" % name))
    assert got.hex() == golden_json("kat_msm.json")[name]


@pytest.mark.parametrize("name", CIRCUITS)
def test_prove(name):
    meta = golden_json(name, "meta.json")
    view = co.ZkeyView(golden_bytes(name, "circuit.zkey"))
    from oracle import groth16_ref as g
    wt = g.read_wtns(golden_bytes(name, "witness.wtns"))
    vals = pack(wt["witness"])
    assert unpack(co.compute_h(view, vals)) == [int(x) for x in meta["h"]]
    sums = co.prove_msm(view, vals)
    assert sums[0:64].hex() == meta["pih"] and sums[64:128].hex() == meta["pi_a"]
    assert sums[128:192].hex() == meta["pib1"] and sums[192:320].hex() == meta["pi_b"] and sums[320:384].hex() == meta["pi_c"]
    assert co.prove(view, vals, int(meta["r"]), int(meta["s"])).hex() == meta["proof_bytes"]


def test_chain_tables_have_known_dlogs():
    rng = random.Random(11)
    k0, kq = rng.randrange(bn.R_MOD), rng.randrange(bn.R_MOD)
    n = 300
    t = co.chain_g1(n, bn.g1_to_bytes(G1.gen), k0, kq).tobytes()
    for i in (0, 1, 2, 150, n - 1):
        assert t[i * 64:(i + 1) * 64] == bn.g1_to_bytes(G1.mul(G1.gen, (k0 + i * kq) % bn.R_MOD))
    t2 = co.chain_g2(40, bn.g2_to_bytes(G2.gen), k0, kq).tobytes()
    for i in (0, 1, 39):
        assert t2[i * 128:(i + 1) * 128] == bn.g2_to_bytes(G2.mul(G2.gen, (k0 + i * kq) % bn.R_MOD))
    assert co.g1_mul(bn.g1_to_bytes(G1.gen), 12345) == bn.g1_to_bytes(G1.mul(G1.gen, 12345))


def test_msm_random_mid_size_vs_dlog():
    """n = 20000 with known discrete logs (beyond the KATs, still seconds)."""
    rng = random.Random(12)
    k0, kq = rng.randrange(bn.R_MOD), rng.randrange(bn.R_MOD)
    n = 20000
    bases = co.chain_g1(n, bn.g1_to_bytes(G1.gen), k0, kq)
    sc = [rng.randrange(bn.R_MOD) for _ in range(n)]
    total = sum(k * (k0 + i * kq) for i, k in enumerate(sc)) % bn.R_MOD
    assert co.msm_g1(bases, pack(sc)) == co.g1_mul(bn.g1_to_bytes(G1.gen), total)


def test_eip196_public_vectors_c():
    """The C restatement against the public EIP-196 precompile vectors (see test_oracle_py.py)."""
    from test_oracle_py import EIP196_2G, EIP196_ADD, EIP196_MUL
    from oracle import bn254 as bn, c_oracle as co
    g1 = bn.g1_to_bytes(bn.G1.gen)
    assert co.g1_mul(g1, 2) == bn.g1_to_bytes(EIP196_2G)
    b, k, r = EIP196_MUL
    assert co.g1_mul(bn.g1_to_bytes(b), k) == bn.g1_to_bytes(r)
    p, q, s = EIP196_ADD
    one = bn.int_to_le32(1)
    assert co.msm_g1(bn.g1_to_bytes(p) + bn.g1_to_bytes(q), one + one) == bn.g1_to_bytes(s)


@pytest.mark.skipif(not co.cpu_has_adx(), reason="the CPU of this box has no BMI2 + ADX")
def test_adx_build_of_the_field_product_equals_the_portable_one():
    """-DZK_ORACLE_ADX (mulx + adcx/adox: the cpu_baseline's fast variant) against the portable C build: the field KATs, the
    edge operands of a Montgomery product, and whole proofs of the golden circuits, byte for byte."""
    import os
    import numpy as np
    here = os.path.dirname(co.__file__)
    co.build(march="native", out="_build/libzkoracle_test_adx.so", adx=True)
    adx = co.load(os.path.join(here, "_build", "libzkoracle_test_adx.so"))
    assert adx.oracle_variant() == b"adx" and co.load().oracle_variant() == b"generic"
    import ctypes as C
    rnd = random.Random(5)
    for field, mod in (("fr", bn.R_MOD), ("fq", bn.Q_MOD)):
        edge = [0, 1, 2, mod - 1, mod - 2, (1 << 253) + 12345, mod // 2, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, (1 << 192) + 7]
        vals = edge + [rnd.randrange(mod) for _ in range(500)]
        a = pack(vals + vals[::-1]) 
        b = pack(vals[::-1] + [vals[(3 * i) % len(vals)] for i in range(len(vals))])
        want = (co.fr_mul_vec if field == "fr" else co.fq_mul_vec)(a, b)
        aa, bb = np.frombuffer(a, dtype=np.uint8), np.frombuffer(b, dtype=np.uint8)
        out = np.zeros(aa.size, dtype=np.uint8)
        getattr(adx, "oracle_%s_mul_vec" % field)(C.c_void_p(out.ctypes.data), C.c_void_p(aa.ctypes.data), C.c_void_p(bb.ctypes.data), aa.size // 32)
        assert out.tobytes() == want
    saved = co._LIB
    try:
        for name in CIRCUITS:
            meta = golden_json(name, "meta.json")
            from oracle import groth16_ref as g
            wt = g.read_wtns(golden_bytes(name, "witness.wtns"))
            vals = pack(wt["witness"])
            co._LIB = saved
            ref = co.prove(co.ZkeyView(golden_bytes(name, "circuit.zkey")), vals, int(meta["r"]), int(meta["s"]))
            co._LIB = adx
            got = co.prove(co.ZkeyView(golden_bytes(name, "circuit.zkey")), vals, int(meta["r"]), int(meta["s"]))
            assert got == ref and got.hex() == meta["proof_bytes"]
    finally:
        co._LIB = saved
