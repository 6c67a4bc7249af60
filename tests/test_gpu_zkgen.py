"""rapidsnark_old_amd.zkgen — the product-side generator of trapdoor-VALID keys: every proof
of such a key must be (a G1, b G2, c G1) with a, b, c computed from the toxic waste in Fr alone
(pairing-free check of SURVEY §8c item 2).  Checked through the C-ABI, the C restatement and the CLI."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import bn254 as bn, c_oracle as co

pytestmark = pytest.mark.gpu

G1B = bn.g1_to_bytes(bn.G1.gen)
G2B = bn.g2_to_bytes(bn.G2.gen)


def test_coef_accumulate_operator_matches_oracle(zk):
    """zk_fr_coef_accumulate = the loop of src/groth16.cpp:62-85: against big-int arithmetic on a synthetic record set."""
    from rapidsnark_old_amd import synth
    k = 8
    n = 1 << k
    img = synth.make_coefs(k, 1, 0)
    w = synth.make_witness(k)
    a, b = zk.fr_coef_accumulate(img, (img.size - 4) // 44, n, w)
    rec = np.frombuffer(img[4:].tobytes(), dtype=synth.COEF_DTYPE)
    wi = [int.from_bytes(w[32 * i:32 * i + 32].tobytes(), "little") for i in range(n)]
    want = [[0] * n, [0] * n]
    for r_ in rec:
        v = int.from_bytes(r_["v"].tobytes(), "little")
        want[int(r_["m"])][int(r_["c"])] = (want[int(r_["m"])][int(r_["c"])] + bn.mont_mul(wi[int(r_["s"])], v, bn.R_MOD)) % bn.R_MOD
    got_a = [int.from_bytes(a[32 * i:32 * i + 32].tobytes(), "little") for i in range(n)]
    got_b = [int.from_bytes(b[32 * i:32 * i + 32].tobytes(), "little") for i in range(n)]
    assert got_a == want[0] and got_b == want[1]


@pytest.mark.parametrize("k,npub", [(6, 1), (12, 2), (16, 3)])
def test_generated_key_is_valid(zk, k, npub):
    from rapidsnark_old_amd import zkgen
    from rapidsnark_old_amd import views
    key = zkgen.generate(k, npub, seed=7)
    assert key["nVars"] == 1 << k and key["pointsC"].size == (key["nVars"] - npub - 1) * 64
    r, s = 0x13579BDF, (1 << 247) - 99
    a, b, c = zkgen.expected_proof_dlogs(key, r, s)
    want = zk.g1_mul(G1B, a) + zk.g2_mul(G2B, b) + zk.g1_mul(G1B, c)
    for precomp in (False, True):
        p = views.ProverFromView(zk, key, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=precomp)
        assert p.prove_host(key["witness"], r, s) == want
        p.lib.zk_prover_destroy(p.h)
    # the independent CPU restatement proves the same key to the same bytes
    assert co.prove(co.ZkeyView(key), key["witness"], r, s) == want
    # a witness that does not satisfy the circuit does NOT pass the trapdoor identity
    bad = key["witness"].copy()
    bad[32 * (key["nVars"] - 1)] ^= 1
    p = views.ProverFromView(zk, key, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False)
    assert p.prove_host(bad, r, s) != want
    p.lib.zk_prover_destroy(p.h)
    vk = zkgen.verification_key(key)
    assert vk["nPublic"] == npub and len(vk["IC"]) == npub + 1 and vk["vk_alpha_1"][2] == "1"
    tau, alpha, beta, gamma, delta = key["trap"]["toxic"]
    assert [int(x) for x in vk["vk_alpha_1"][:2]] == list(bn.G1.mul(bn.G1.gen, alpha))
    g2 = bn.G2.mul(bn.G2.gen, gamma)
    assert [[int(x) for x in pair] for pair in vk["vk_gamma_2"][:2]] == [list(g2[0]), list(g2[1])]


def test_circuit_like_key_is_valid_sparse_and_verifies(zk):
    """zkgen.generate(circuit_like=True): nVars = 3/4 of the domain + 5, a witness dominated by booleans, all-zero rows in
    A / B1 / B2 for the wires those matrices never read — and still a VALID key: the GPU proof equals the toxic-waste
    prediction (scalar multiplications by the oracle), the C restatement's proof, and passes the pairing check."""
    from oracle import groth16_ref as og, pairing
    from rapidsnark_old_amd import zkgen, views
    k, npub = 13, 2
    key = zkgen.generate(k, npub, seed=5, circuit_like=True)
    nv = key["nVars"]
    assert nv == 3 * (1 << k) // 4 + 5 and key["domainSize"] == 1 << k
    w = np.asarray(key["witness"]).reshape(nv, 32)
    boolean = (w[:, 1:].max(axis=1) == 0) & (w[:, 0] <= 1)
    full = w[:, 16:].max(axis=1) > 0
    assert 0.7 < boolean.mean() < 0.9 and 0.02 < full.mean() < 0.2
    za = np.asarray(key["pointsA"]).reshape(nv, 64).max(axis=1) == 0
    zb1 = np.asarray(key["pointsB1"]).reshape(nv, 64).max(axis=1) == 0
    zb2 = np.asarray(key["pointsB2"]).reshape(nv, 128).max(axis=1) == 0
    assert za.mean() > 0.25 and zb1.mean() > 0.25 and (zb1 == zb2).all() and not za.all()
    rec = np.frombuffer(np.asarray(key["coefs"])[4:].tobytes(), dtype=synth_dtype())
    assert (rec["v"].max(axis=1) > 0).all() and rec["s"].max() < nv
    r, s = 0xBADC0FFEE, (1 << 240) + 31337
    a, b, c = zkgen.expected_proof_dlogs(key, r, s)
    pts = (bn.G1.mul(bn.G1.gen, a), bn.G2.mul(bn.G2.gen, b), bn.G1.mul(bn.G1.gen, c))
    want = bn.g1_to_bytes(pts[0]) + bn.g2_to_bytes(pts[1]) + bn.g1_to_bytes(pts[2])
    for precomp in (False, True):
        p = views.ProverFromView(zk, key, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=precomp)
        assert p.prove_host(key["witness"], r, s) == want
        p.lib.zk_prover_destroy(p.h)
    assert co.prove(co.ZkeyView(key), key["witness"], r, s) == want
    vkj = zkgen.verification_key(key)
    vk = {"alpha1": tuple(int(x) for x in vkj["vk_alpha_1"][:2]), "IC": [tuple(int(x) for x in pt[:2]) for pt in vkj["IC"]]}
    for name in ("beta2", "gamma2", "delta2"):
        j = vkj["vk_%s_2" % name[:-1]]
        vk[name] = ((int(j[0][0]), int(j[0][1])), (int(j[1][0]), int(j[1][1])))
    pub = [int.from_bytes(w[i].tobytes(), "little") for i in range(1, npub + 1)]
    assert pairing.groth16_verify(vk, pub, pts)
    assert og.proof_to_json(pts) == zk.proof_to_json(want)


def test_semaphore_like_key_is_valid_and_has_the_shape_it_claims(zk):
    """zkgen.generate(semaphore_like=True) — the proxy for BASELINE configs[4]: 4 public signals, chains of x^5 S-box rounds
    between Merkle-style muxes (constraints thousands deep), nearly every signal a full-size field element, one or two
    coefficients per row and matrix — and a VALID key: GPU proof = toxic-waste prediction = C restatement, pairing check passes."""
    from oracle import pairing
    from rapidsnark_old_amd import zkgen, views
    k, npub = 12, 4
    key = zkgen.generate(k, npub, seed=2, semaphore_like=True)
    nv = key["nVars"]
    assert nv == 3 * (1 << k) // 4 + 5 and key["nPublic"] == 4
    w = np.asarray(key["witness"]).reshape(nv, 32)
    boolean = (w[:, 1:].max(axis=1) == 0) & (w[:, 0] <= 1)
    full = w[:, 24:].max(axis=1) > 0
    assert boolean.mean() < 0.1 and full.mean() > 0.85
    rec = np.frombuffer(np.asarray(key["coefs"])[4:].tobytes(), dtype=synth_dtype())
    assert (rec["v"].max(axis=1) > 0).all() and rec.size < 4 * nv          # sparse rows: < 2 coefficients per row and matrix on average
    # the constraints really form a chain: the witness satisfies A.w o B.w = C.w row by row (checked through the operator)
    am, bm = zk.fr_coef_accumulate(key["coefs"], rec.size, 1 << k, key["witness"])
    m = nv - 1 - 64
    one = np.tile(np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint8), m)
    ab = np.frombuffer(zk.fr_mul_vec(zk.fr_mul_vec(am[:m * 32], bm[:m * 32]), one), dtype=np.uint8).reshape(m, 32)      # (aR)(bR)/R/R = a b
    assert (ab == w[65:]).all()
    r, s = 0xFEEDFACE, (1 << 250) + 99
    a, b, c = zkgen.expected_proof_dlogs(key, r, s)
    pts = (bn.G1.mul(bn.G1.gen, a), bn.G2.mul(bn.G2.gen, b), bn.G1.mul(bn.G1.gen, c))
    want = bn.g1_to_bytes(pts[0]) + bn.g2_to_bytes(pts[1]) + bn.g1_to_bytes(pts[2])
    p = views.ProverFromView(zk, key, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True)
    assert p.prove_host(key["witness"], r, s) == want
    p.lib.zk_prover_destroy(p.h)
    assert co.prove(co.ZkeyView(key), key["witness"], r, s) == want
    vkj = zkgen.verification_key(key)
    vk = {"alpha1": tuple(int(x) for x in vkj["vk_alpha_1"][:2]), "IC": [tuple(int(x) for x in pt[:2]) for pt in vkj["IC"]]}
    for name in ("beta2", "gamma2", "delta2"):
        j = vkj["vk_%s_2" % name[:-1]]
        vk[name] = ((int(j[0][0]), int(j[0][1])), (int(j[1][0]), int(j[1][1])))
    pub = [int.from_bytes(w[i].tobytes(), "little") for i in range(1, npub + 1)]
    assert pairing.groth16_verify(vk, pub, pts)


def synth_dtype():
    from rapidsnark_old_amd import synth
    return synth.COEF_DTYPE


def test_zkgen_tool_at_2p20_with_cli(zk, tmp_path):
    """tools/zkgen.py 20 --prove: a valid key at BASELINE configs[1]'s size written to disk (~1.1 GB .zkey), proved by
    the one-shot CLI, proof.json checked against the toxic waste; the multi-GPU path proves the same file."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "zkgen.py"), "20", str(tmp_path), "--prove"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "trapdoor check of proof.json: PASS" in res.stdout, res.stdout + res.stderr
    vk = json.load(open(tmp_path / "verification_key.json"))
    assert vk["protocol"] == "groth16" and vk["curve"] == "bn128" and len(vk["IC"]) == 3
    pub = json.load(open(tmp_path / "public.json"))
    assert len(pub) == 2
    # the same files through the Groth16 VERIFICATION equation (oracle/pairing.py): needs neither the toxic waste nor any
    # arithmetic of the product — once with the verification_key.json zkgen wrote, once with the .zkey's own sections 2 and 3
    for vkfile in ("verification_key.json", "circuit.zkey"):
        v = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "refcheck", "verify.py"), str(tmp_path / "proof.json"), str(tmp_path / "public.json"),
                            str(tmp_path / vkfile)], capture_output=True, text=True, timeout=300)
        assert v.returncode == 0 and "OK: the proof verifies" in v.stdout, v.stdout + v.stderr
    r, s = 0x0123456789ABCDEF, (1 << 200) + 12345
    mp = zk.MultiProver(str(tmp_path / "circuit.zkey"), [0, 0, 0, 0], precomp=True)
    proof = mp.prove(str(tmp_path / "witness.wtns"), r, s)
    mp.close()
    assert zk.proof_to_json(proof) == open(tmp_path / "proof.json").read()
