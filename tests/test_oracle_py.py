"""CPU: the Python big-int oracle against its own committed golden vectors and the
mathematical pins it stands on (SURVEY §A.2 constants, r*G = O, trapdoor check)."""
import json
import os
import random

import pytest

from conftest import CIRCUITS, ROOT, golden_bytes, golden_json
from oracle import bn254 as bn, groth16_ref as g
from oracle.bn254 import G1, G2, R_MOD, Q_MOD


def test_constants_of_survey_a2():
    assert bn.ROOT_2_28 == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert pow(bn.ROOT_2_28, 1 << 27, R_MOD) == R_MOD - 1
    assert bn.MONT_R % R_MOD == 6350874878119819312338956282401532410528162663560392320966563075034087161851
    assert bn.MONT_R ** 2 % R_MOD == 944936681149208446651664254269745548490766851729442924617792859073125903783
    assert bn.MONT_R % Q_MOD == 6350874878119819312338956282401532409788428879151445726012394534686998597021
    assert bn.MONT_R ** 2 % Q_MOD == 3096616502983703923843567936837374451735540968419076528771170197431451843209
    assert (-pow(R_MOD, -1, 1 << 64)) % (1 << 64) == 14042775128853446655
    assert (-pow(Q_MOD, -1, 1 << 64)) % (1 << 64) == 9786893198990664585
    assert Q_MOD % 4 == 3 and (R_MOD - 1) % (1 << 28) == 0 and (R_MOD - 1) % (1 << 29) != 0


def test_group_orders_and_curve_membership():
    assert G1.is_on_curve(G1.gen) and G2.is_on_curve(G2.gen)
    assert G1.mul(G1.gen, R_MOD) is None and G2.mul(G2.gen, R_MOD) is None
    P = G1.mul(G1.gen, 5)
    assert G1.add(P, G1.neg(P)) is None and G1.add(P, P) == G1.dbl(P) == G1.mul(G1.gen, 10)
    Q = G2.mul(G2.gen, 7)
    assert G2.add(Q, Q) == G2.mul(G2.gen, 14) and G2.sub(Q, Q) is None
    assert bn.g1_from_bytes(bn.g1_to_bytes(P)) == P and bn.g2_from_bytes(bn.g2_to_bytes(Q)) == Q
    assert bn.g1_to_bytes(None) == bytes(64)


def test_ntt_semantics_of_the_reference_pipeline():
    """idft, *w2n^i, dft  evaluates the polynomial at w2n^(2i+1)."""
    rng = random.Random(1)
    n = 8
    x = [rng.randrange(R_MOD) for _ in range(n)]
    assert bn.ntt(bn.ntt(x), inverse=True) == x
    co = bn.ntt(x, inverse=True)
    w2n = bn.fr_root(4)
    ev = bn.ntt([c * pow(w2n, i, R_MOD) % R_MOD for i, c in enumerate(co)])
    for i in range(n):
        pt = pow(w2n, 2 * i + 1, R_MOD)
        assert ev[i] == sum(c * pow(pt, j, R_MOD) for j, c in enumerate(co)) % R_MOD


@pytest.mark.parametrize("name", CIRCUITS)
def test_golden_proofs_reproduce_and_parse(name):
    meta = golden_json(name, "meta.json")
    zk = g.read_zkey(golden_bytes(name, "circuit.zkey"))
    wt = g.read_wtns(golden_bytes(name, "witness.wtns"))
    assert wt["prime"] == R_MOD and wt["nVars"] == zk.nVars and wt["witness"][0] == 1
    if zk.nVars <= 60:      # keep the CPU suite short: big-int MSMs are slow
        proof = g.prove(zk, wt["witness"], int(meta["r"]), int(meta["s"]))
        assert g.proof_to_bytes(proof).hex() == meta["proof_bytes"]
        assert g.proof_to_json(proof) == golden_bytes(name, "proof.json").decode()
    assert g.public_to_json(wt["witness"], zk.nPublic) == golden_bytes(name, "public.json").decode()
    pj = json.loads(golden_bytes(name, "proof.json"))
    assert pj["protocol"] == "groth16" and pj["pi_a"][2] == "1" and pj["pi_b"][2] == ["1", "0"]


def test_trapdoor_check_accepts_valid_and_rejects_tampered():
    rng = random.Random(42)
    r1, w = g.random_r1cs(rng, 6, 2)
    toxic = tuple(rng.randrange(1, R_MOD) for _ in range(5))
    zk, trap = g.setup(r1, toxic)
    r, s = rng.randrange(1 << 248), rng.randrange(1 << 248)
    proof = g.prove(zk, w, r, s)
    assert g.trapdoor_check(trap, 2, w, r, s, proof)
    bad = (proof[0], proof[1], G1.add(proof[2], G1.gen))
    assert not g.trapdoor_check(trap, 2, w, r, s, bad)
    w2 = list(w)
    w2[-1] = (w2[-1] + 1) % R_MOD                       # unsatisfying witness
    assert not g.trapdoor_check(trap, 2, w2, r, s, g.prove(zk, w2, r, s))


def test_binfile_errors_follow_the_reference_texts():
    with pytest.raises(ValueError, match="Invalid file type"):
        g.read_binfile(b"abcd" + bytes(8), b"zkey", 1)
    with pytest.raises(ValueError, match="Invalid version"):
        g.read_binfile(b"zkey" + (9).to_bytes(4, "little") + bytes(4), b"zkey", 1)


# Public BN254 (alt_bn128) vectors that were NOT produced by this repository: the "chfast1" cases of the
# Ethereum precompile test suites for EIP-196 (ecAdd 0x06 / ecMul 0x07) and the value of 2*G1 that
# appears throughout the EIP-196 test corpus.  The reference holds no vectors of its own,
# so these are the only third-party numbers the oracle's curve arithmetic can be pinned to.
EIP196_2G = (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
             0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)
EIP196_ADD = ((0x18b18acfb4c2c30276db5411368e7185b311dd124691610c5d3b74034e093dc9,
               0x063c909c4720840cb5134cb9f59fa749755796819658d32efc0d288198f37266),
              (0x07c2b7f58a84bd6145f00c9c2bc0bb1a187f20ff2c92963a88019e7c6a014eed,
               0x06614e20c147e940f2d70da3f74c9a17df361706a4485c742bd6788478fa17d7),
              (0x2243525c5efd4b9c3d3c45ac0ca3fe4dd85e830a4ce6b65fa1eeaee202839703,
               0x301d1d33be6da8e509df21cc35964723180eed7532537db9ae5e7d48f195c915))
EIP196_MUL = ((0x2bd3e6d0f3b142924f5ca7b49ce5b9d54c4703d7ae5648e61d02268b1a0a9fb7,
               0x21611ce0a6af85915e2f1d70300909ce2e49dfad4a4619c8390cae66cefdb204),
              0x11138ce750fa15c2,
              (0x070a8d6a982153cae4be29d434e8faef8a47b274a053f5a4ee2a6c9c13c31e5c,
               0x031b8ce914eba3a9ffb989f9cdd5b0f01943074bf4f0f315690ec3cec6981afc))


def test_eip196_public_vectors():
    assert G1.dbl(G1.gen) == EIP196_2G == G1.add(G1.gen, G1.gen) == G1.mul(G1.gen, 2)
    p, q, s = EIP196_ADD
    assert G1.is_on_curve(p) and G1.is_on_curve(q) and G1.add(p, q) == s and G1.add(q, p) == s
    b, k, r = EIP196_MUL
    assert G1.is_on_curve(b) and G1.mul(b, k) == r
    # EIP-197 generator of G2 (the constant the pairing precompile fixes): on the twist, order r
    assert G2.gen == ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
                       11559732032986387107991004021392285783925812861821192530917403151452391805634),
                      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
                       4082367875863433681332203403145435568316851327593401208105741076214120093531))
    assert G2.is_on_curve(G2.gen) and G2.mul(G2.gen, R_MOD) is None


# ---------------------------------------------------------------- pairing + Groth16 verification (oracle/pairing.py)
def test_pairing_is_bilinear_nondegenerate_and_of_order_r():
    from oracle import pairing as pr
    e = pr.pairing(bn.G1_GEN, bn.G2_GEN)
    assert e != pr.F12_ONE
    assert pr.f12_pow(e, bn.R_MOD) == pr.F12_ONE
    a, b = 0x1234567890abcdef1234, 0xfedcba98765432100123456789
    P, Q = bn.G1.mul(bn.G1_GEN, a), bn.G2.mul(bn.G2_GEN, b)
    assert pr.pairing(P, Q) == pr.f12_pow(e, a * b % bn.R_MOD)
    assert pr.pairing(P, bn.G2_GEN) == pr.pairing(bn.G1_GEN, bn.G2.mul(bn.G2_GEN, a))
    negP = (P[0], (-P[1]) % bn.Q_MOD)
    assert pr.pairing_product_is_one([(P, Q), (negP, Q)])
    assert not pr.pairing_product_is_one([(P, Q), (P, Q)])
    assert pr.pairing(None, Q) == pr.F12_ONE


def test_every_golden_proof_satisfies_the_groth16_verification_equation():
    """Independent of the trapdoor and of every prover in this repository: e(A,B) = e(alpha,beta) e(vk_x,gamma) e(C,delta) with
    the key's own section 2 / section 3 — and a proof with one coordinate changed, or another public input, does not."""
    import importlib.util
    import json as _json
    spec = importlib.util.spec_from_file_location("refcheck_verify", os.path.join(ROOT, "tools", "refcheck", "verify.py"))
    v = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(v)
    from oracle import pairing as pr
    for name in ("multiplier2", "r1cs_n8", "r1cs_nopub", "r1cs_n64", "r1cs_n256"):
        d = os.path.join(ROOT, "tests", "golden", name)
        assert v.verify_files(os.path.join(d, "proof.json"), os.path.join(d, "public.json"), os.path.join(d, "circuit.zkey")), name
    d = os.path.join(ROOT, "tests", "golden", "multiplier2")
    vk = v.load_vk(os.path.join(d, "circuit.zkey"))
    pj = _json.load(open(os.path.join(d, "proof.json")))
    pub = [int(x) for x in _json.load(open(os.path.join(d, "public.json")))]
    A, B, C = v.g1(pj["pi_a"]), v.g2(pj["pi_b"]), v.g1(pj["pi_c"])
    assert pr.groth16_verify(vk, pub, (A, B, C))
    assert not pr.groth16_verify(vk, [pub[0] + 1] + pub[1:], (A, B, C))                   # another statement
    assert not pr.groth16_verify(vk, pub, (bn.G1.dbl(A), B, C))                           # a valid point, the wrong one
    assert not pr.groth16_verify(vk, pub, (A, B, (C[0], (C[1] + 1) % bn.Q_MOD)))          # not on the curve
