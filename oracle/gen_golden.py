"""Generate tests/golden/* from the Python big-int oracle (run once; outputs are committed).

    python -m oracle.gen_golden

TEST INFRASTRUCTURE.  The reference ships no vectors (`package.json:7`), so every fixture
is produced here and pinned by the pairing-free trapdoor check (oracle/groth16_ref.py).
Deterministic: fixed seeds, fixed toxic waste, fixed (r, s).
"""
import json
import os
import random

from . import bn254 as bn
from . import groth16_ref as g
from .bn254 import R_MOD, Q_MOD, G1, G2

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def w(path, data):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    mode = "wb" if isinstance(data, (bytes, bytearray)) else "w"
    with open(path, mode) as f:
        f.write(data)


def circuit_fixture(name, r1cs, witness, seed, domain_size=None):
    rng = random.Random(seed)
    toxic = tuple(rng.randrange(1, R_MOD) for _ in range(5))
    zk, trap = g.setup(r1cs, toxic, domain_size)
    assert r1cs.is_satisfied(witness)
    r, s = rng.randrange(1 << 248), rng.randrange(1 << 248)      # 31 random bytes each (groth16.cpp:216-217)
    proof = g.prove(zk, witness, r, s)
    assert g.trapdoor_check(trap, r1cs.nPublic, witness, r, s, proof), name
    h, (a, b, c) = g.compute_h(zk, witness)
    d = os.path.join(OUT, name)
    w(os.path.join(d, "circuit.zkey"), g.write_zkey(zk))
    w(os.path.join(d, "witness.wtns"), g.write_wtns(witness))
    w(os.path.join(d, "proof.json"), g.proof_to_json(proof))
    w(os.path.join(d, "public.json"), g.public_to_json(witness, r1cs.nPublic))
    hexp = lambda P, enc: enc(P).hex()
    meta = {
        "r": str(r), "s": str(s), "toxic": [str(t) for t in toxic],
        "proof_bytes": g.proof_to_bytes(proof).hex(),
        # the intermediate values the reference's LOG_DEBUG probes name (groth16.cpp:103-207)
        "a_head": [str(x) for x in a[:2]], "b_head": [str(x) for x in b[:2]],
        "h": [str(x) for x in h],
        "pih": hexp(G1.msm(zk.H, h), bn.g1_to_bytes),
        "pi_a": hexp(G1.msm(zk.A, witness), bn.g1_to_bytes),
        "pib1": hexp(G1.msm(zk.B1, witness), bn.g1_to_bytes),
        "pi_b": hexp(G2.msm(zk.B2, witness), bn.g2_to_bytes),
        "pi_c": hexp(G1.msm(zk.C, witness[zk.nPublic + 1:]), bn.g1_to_bytes),
    }
    w(os.path.join(d, "meta.json"), json.dumps(meta, indent=1))
    print(name, "nVars", zk.nVars, "domain", zk.domainSize, "coefs", len(zk.coefs))


def field_kats():
    rng = random.Random(101)
    out = {}
    for name, p in (("fr", R_MOD), ("fq", Q_MOD)):
        edge = [0, 1, p - 1, p - 2, (1 << 256) % p, 2, (p + 1) // 2]
        a = edge + [rng.randrange(p) for _ in range(25)]
        b = list(reversed(edge)) + [rng.randrange(p) for _ in range(25)]
        out[name] = {"a": [str(x) for x in a], "b": [str(x) for x in b],
                     "mont_mul": [str(bn.mont_mul(x, y, p)) for x, y in zip(a, b)]}
    w(os.path.join(OUT, "kat_field.json"), json.dumps(out, indent=1))


def ntt_kats():
    rng = random.Random(202)
    out = {}
    for n in (1, 2, 4, 8, 64, 2048, 4096):
        x = [rng.randrange(R_MOD) for _ in range(n)]
        if n >= 4:
            x[1] = 0
            x[2] = R_MOD - 1
        fwd = bn.ntt(x)
        inv = bn.ntt(x, inverse=True)
        # keep big cases compact: store input seed-regenerable? store everything for n<=64, hashes + heads above
        if n <= 64:
            out[str(n)] = {"x": [str(v) for v in x], "fft": [str(v) for v in fwd], "ifft": [str(v) for v in inv]}
        else:
            import hashlib
            enc = lambda vs: hashlib.sha256(b"".join(bn.int_to_le32(v) for v in vs)).hexdigest()
            out[str(n)] = {"seed": 202, "x_sha256": enc(x), "fft_sha256": enc(fwd), "ifft_sha256": enc(inv),
                           "x_head": [str(v) for v in x[:4]], "fft_head": [str(v) for v in fwd[:4]],
                           "ifft_head": [str(v) for v in inv[:4]]}
            w(os.path.join(OUT, "ntt_x_%d.bin" % n), b"".join(bn.int_to_le32(v) for v in x))
            w(os.path.join(OUT, "ntt_fft_%d.bin" % n), b"".join(bn.int_to_le32(v) for v in fwd))
            w(os.path.join(OUT, "ntt_ifft_%d.bin" % n), b"".join(bn.int_to_le32(v) for v in inv))
    w(os.path.join(OUT, "kat_ntt.json"), json.dumps(out, indent=1))


def msm_kats():
    rng = random.Random(303)
    t1 = G1.fixed_base_table(G1.gen)
    t2 = G2.fixed_base_table(G2.gen)
    cases = {}

    def rand_pts(curve, tbl, n):
        return [curve.mul_fixed(tbl, rng.randrange(1, R_MOD)) for _ in range(n)]

    def build(curve, tbl, enc, n, tag):
        pts = rand_pts(curve, tbl, n)
        sc = [rng.randrange(R_MOD) for _ in range(n)]
        if n >= 17:
            # edge cases the domain has: zero scalar, infinity base, repeated base, P and -P,
            # scalar 1, r-1, small scalars, a scalar >= r (raw 256-bit input, reduced mod r)
            sc[0] = 0
            pts[1] = None
            pts[3] = pts[2]
            sc[3] = sc[2]                       # same bucket everywhere -> doubling path
            pts[5] = curve.neg(pts[4])
            sc[5] = sc[4]                       # P + (-P) in every bucket -> infinity path
            sc[6] = 1
            sc[7] = R_MOD - 1
            sc[8] = 2
            sc[9] = (1 << 16) - 1
            sc[10] = 1 << 15                    # exactly half: digit boundary
            sc[11] = (1 << 15) + 1
            sc[12] = R_MOD + 5                  # >= r
            sc[13] = (1 << 256) - 1
        res = curve.msm(pts, [k % R_MOD for k in sc])
        name = "%s_n%d" % (tag, n)
        w(os.path.join(OUT, "msm_%s_bases.bin" % name), b"".join(enc(P) for P in pts))
        w(os.path.join(OUT, "msm_%s_scalars.bin" % name), b"".join(bn.int_to_le32(k) for k in sc))
        cases[name] = enc(res).hex()
        print("msm", name)

    for n in (1, 2, 3, 17, 1000):
        build(G1, t1, bn.g1_to_bytes, n, "g1")
    for n in (1, 2, 17, 300):
        build(G2, t2, bn.g2_to_bytes, n, "g2")
    # all-cancelling case: result is infinity
    P = G1.mul_fixed(t1, 12345)
    w(os.path.join(OUT, "msm_g1_cancel_bases.bin"), bn.g1_to_bytes(P) + bn.g1_to_bytes(G1.neg(P)))
    w(os.path.join(OUT, "msm_g1_cancel_scalars.bin"), bn.int_to_le32(77) * 2)
    cases["g1_cancel"] = bytes(64).hex()
    w(os.path.join(OUT, "kat_msm.json"), json.dumps(cases, indent=1))


def main():
    field_kats()
    ntt_kats()
    msm_kats()
    circuit_fixture("multiplier2", g.multiplier2_r1cs(), [1, 33, 3, 11], seed=1)
    rng = random.Random(7)
    r1, w1 = g.random_r1cs(rng, 5, 1)
    circuit_fixture("r1cs_n8", r1, w1, seed=2)
    r2, w2 = g.random_r1cs(rng, 50, 3)
    circuit_fixture("r1cs_n64", r2, w2, seed=3)
    # nPublic == 0 (public.json is `null`, quirk Q7) and an explicit larger domain
    r3, w3 = g.random_r1cs(rng, 10, 0)
    circuit_fixture("r1cs_nopub", r3, w3, seed=4, domain_size=32)
    r4, w4 = g.random_r1cs(rng, 200, 2)
    circuit_fixture("r1cs_n256", r4, w4, seed=5)


if __name__ == "__main__":
    main()
