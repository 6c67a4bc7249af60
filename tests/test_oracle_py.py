"""CPU: the Python big-int oracle against its own committed golden vectors and the
mathematical pins it stands on (SURVEY §A.2 constants, r*G = O, trapdoor check)."""
import json
import random

import pytest

from conftest import CIRCUITS, golden_bytes, golden_json
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
