"""GPU parity: Fr/Fq Montgomery product and the Fr NTT family vs the oracle / golden KATs.
Bit-exact (integer arithmetic).  Everything goes through the C-ABI (libzkhip.so)."""
import hashlib
import random

import pytest

from conftest import golden_bytes, golden_json
from oracle import bn254 as bn

pytestmark = pytest.mark.gpu

le = bn.int_to_le32


def pack(vals):
    return b"".join(le(v) for v in vals)


def unpack(b):
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_mont_mul_kat(zk, field):
    k = golden_json("kat_field.json")[field]
    fn = zk.fr_mul_vec if field == "fr" else zk.fq_mul_vec
    out = unpack(fn(pack(int(x) for x in k["a"]), pack(int(x) for x in k["b"])))
    assert out == [int(x) for x in k["mont_mul"]]


@pytest.mark.parametrize("field,p", [("fr", bn.R_MOD), ("fq", bn.Q_MOD)])
def test_mont_mul_random(zk, field, p):
    rng = random.Random(5)
    n = 20000
    a = [rng.randrange(p) for _ in range(n)]
    b = [rng.randrange(p) for _ in range(n)]
    fn = zk.fr_mul_vec if field == "fr" else zk.fq_mul_vec
    out = unpack(fn(pack(a), pack(b)))
    rinv = pow(bn.MONT_R, -1, p)
    assert out == [x * y * rinv % p for x, y in zip(a, b)]


def test_mul_vec_empty(zk):
    assert zk.fr_mul_vec(b"", b"") == b""


@pytest.mark.parametrize("n", [1, 2, 4, 8, 64])
def test_ntt_kat_small(zk, n):
    k = golden_json("kat_ntt.json")[str(n)]
    x = [int(v) for v in k["x"]]
    xm = pack(bn.to_mont(v, bn.R_MOD) for v in x)       # elements stay in Montgomery form
    fwd = [bn.from_mont(v, bn.R_MOD) for v in unpack(zk.fr_ntt(xm, inverse=False))]
    inv = [bn.from_mont(v, bn.R_MOD) for v in unpack(zk.fr_ntt(xm, inverse=True))]
    assert fwd == [int(v) for v in k["fft"]]
    assert inv == [int(v) for v in k["ifft"]]


@pytest.mark.parametrize("n", [2048, 4096])
def test_ntt_kat_large(zk, n):
    # linear map => standard-form input gives standard-form output; golden files are standard form
    x = golden_bytes("ntt_x_%d.bin" % n)
    assert zk.fr_ntt(x, inverse=False) == golden_bytes("ntt_fft_%d.bin" % n)
    assert zk.fr_ntt(x, inverse=True) == golden_bytes("ntt_ifft_%d.bin" % n)
    k = golden_json("kat_ntt.json")[str(n)]
    assert hashlib.sha256(zk.fr_ntt(x)).hexdigest() == k["fft_sha256"]


@pytest.mark.parametrize("logn", [12, 13, 16, 19, 20])
def test_ntt_roundtrip_and_properties(zk, logn):
    """Sizes the Python oracle cannot reach: size-independent properties.
    ifft(fft(x)) == x;  fft(delta_1) is the root table;  fft(const at 0) is constant;  linearity."""
    n = 1 << logn
    rng = random.Random(logn)
    import numpy as np
    raw = np.frombuffer(rng.randbytes(32 * n), dtype=np.uint8).reshape(n, 32).copy()
    raw[:, 31] &= 0x1f                                   # < 2^253 < r
    x = raw.tobytes()
    y = zk.fr_ntt(x, inverse=False)
    assert zk.fr_ntt(y, inverse=True) == x
    # delta at index 1 -> X[i] = w^i
    d = bytearray(32 * n)
    d[32:64] = le(1)
    roots = unpack(zk.fr_ntt(bytes(d)))
    w = bn.fr_root(logn)
    for i in [0, 1, 2, 3, n // 2, n - 1, 12345 % n]:
        assert roots[i] == pow(w, i, bn.R_MOD)
    # linearity on a few outputs: fft(x + d) = fft(x) + fft(d)
    xs = unpack(x)
    s = pack((a + b) % bn.R_MOD for a, b in zip(xs, unpack(bytes(d))))
    ys = unpack(zk.fr_ntt(s))
    yv = unpack(y)
    for i in [0, 1, n // 3, n - 1]:
        assert ys[i] == (yv[i] + roots[i]) % bn.R_MOD


@pytest.mark.parametrize("n", [1, 2, 8, 64, 2048, 4096, 8192])
def test_abc_to_h_matches_oracle(zk, n):
    """The whole groth16.cpp:88-163 pipeline: c=a.b, 3x(ifft, coset shift, fft), h=fromMont(a.b-c)."""
    rng = random.Random(1000 + n)
    a = [rng.randrange(bn.R_MOD) for _ in range(n)]
    b = [rng.randrange(bn.R_MOD) for _ in range(n)]
    c = [x * y % bn.R_MOD for x, y in zip(a, b)]
    logn = n.bit_length() - 1
    w2n = bn.fr_root(logn + 1)

    def coset(v):
        co = bn.ntt(v, inverse=True)
        return bn.ntt([x * pow(w2n, i, bn.R_MOD) % bn.R_MOD for i, x in enumerate(co)])

    ae, be, ce = coset(a), coset(b), coset(c)
    want = [(x * y - z) % bn.R_MOD for x, y, z in zip(ae, be, ce)]
    got = unpack(zk.fr_abc_to_h(pack(bn.to_mont(v, bn.R_MOD) for v in a), pack(bn.to_mont(v, bn.R_MOD) for v in b)))
    assert got == want


@pytest.mark.parametrize("logn", [3, 5, 9, 10, 11, 12, 13, 14, 16, 19, 21, 22, 23])
def test_abc_to_h_against_the_c_restatement_at_every_pass_plan(zk, logn):
    """The coset-evaluation pipeline (csrc/nttpair.hip) at every shape of its pass plan — a single tile (<= 2^11), tile + one
    strided pass of 1..10 bits (256 threads), of 11 bits (512 threads, 2^22), two strided passes (2^23) — against the C
    restatement's bit-reversal FFT (oracle/c/zk_oracle.c, src/groth16.cpp:98-163): c = a o b, three times ifft / coset shift
    / fft, h = fromMontgomery(a.b - c).  Compared on all positions up to 2^14 and on 4096 sampled ones above."""
    import numpy as np
    from oracle import c_oracle as co
    n = 1 << logn
    rng = np.random.default_rng(77 + logn)
    from rapidsnark_old_amd import synth
    a = synth.random_fr_bytes(rng, n).reshape(-1)                  # Montgomery form, like the reference's a[] / b[]
    b = synth.random_fr_bytes(rng, n).reshape(-1)
    c = np.frombuffer(co.fr_mul_vec(a, b), dtype=np.uint8)
    # powers of w_2n in Montgomery form, by doubling (vector products of the restatement)
    w2n = bn.to_mont(bn.fr_root(logn + 1), bn.R_MOD)
    pw = np.frombuffer(le(bn.to_mont(1, bn.R_MOD)), dtype=np.uint8).copy()
    step = w2n
    while pw.size < n * 32:
        pw = np.concatenate([pw, np.frombuffer(co.fr_mul_vec(pw, np.tile(np.frombuffer(le(step), dtype=np.uint8), pw.size // 32)), dtype=np.uint8)])
        step = bn.mont_mul(step, step, bn.R_MOD)

    def coset(v):
        return np.frombuffer(co.fr_fft(co.fr_mul_vec(co.fr_fft(v, inverse=True), pw), inverse=False), dtype=np.uint8)

    ae, be, ce = coset(a), coset(b), coset(c)
    ab = np.frombuffer(co.fr_mul_vec(ae, be), dtype=np.uint8)
    got = np.frombuffer(zk.fr_abc_to_h(a, b), dtype=np.uint8)
    idx = range(n) if logn <= 14 else sorted(set(int(i) for i in rng.integers(0, n, 4096)) | {0, 1, n - 1, n // 2, 2047, 2048})
    rinv = pow(bn.MONT_R, -1, bn.R_MOD)
    for i in idx:
        x = int.from_bytes(ab[32 * i:32 * i + 32].tobytes(), "little")
        z = int.from_bytes(ce[32 * i:32 * i + 32].tobytes(), "little")
        assert int.from_bytes(got[32 * i:32 * i + 32].tobytes(), "little") == (x - z) * rinv % bn.R_MOD, (logn, i)
