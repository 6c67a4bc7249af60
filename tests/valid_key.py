"""Trapdoor-valid Groth16 keys at sizes the pure-Python oracle cannot reach: the Fr half of the
setup (every table entry's discrete logarithm) comes from the oracle (`setup_scalars`), the curve half from
the product's batch fixed-base kernel (`zk_fixed_base_g1/g2`).  Test infrastructure only."""
import random
import struct

import numpy as np

from oracle import bn254 as bn
from oracle import groth16_ref as g

R_MOD = bn.R_MOD


def build(zk, k, seed, n_public=2):
    """-> (wl, witness_u8, trap, w_ints): `wl` is a zkey view (dict of numpy arrays, same keys as
    rapidsnark_old_amd.synth.workload) of a VALID key over a random satisfiable R1CS with domain 2^k;
    the witness satisfies it."""
    rng = random.Random(seed)
    n = 1 << k
    m = n - n_public - 1                                   # snarkjs appends nPublic + 1 rows
    r1cs, w = g.random_r1cs(rng, m, n_public, extra_vars=2)
    assert r1cs.is_satisfied(w)
    toxic = tuple(rng.randrange(1, R_MOD) for _ in range(5))
    tau, alpha, beta, gamma, delta = toxic
    trap = g.setup_scalars(r1cs, toxic, domain_size=n)
    G1B, G2B = bn.g1_to_bytes(bn.G1.gen), bn.g2_to_bytes(bn.G2.gen)
    r2 = (bn.MONT_R * bn.MONT_R) % R_MOD
    coefs = struct.pack("<I", len(trap["coefs"])) + b"".join(
        struct.pack("<III", mm, c, s) + bn.int_to_le32((v * r2) % R_MOD) for (mm, c, s, v) in trap["coefs"])
    u8 = lambda b: np.frombuffer(bytes(b), dtype=np.uint8)
    wl = {
        "k": k, "nVars": r1cs.nVars, "nPublic": n_public, "domainSize": n, "nCoefs": len(trap["coefs"]),
        "coefs": u8(coefs),
        "pointsA": zk.fixed_base_g1(G1B, trap["At"]), "pointsB1": zk.fixed_base_g1(G1B, trap["Bt"]),
        "pointsB2": zk.fixed_base_g2(G2B, trap["Bt"]),
        "pointsC": zk.fixed_base_g1(G1B, trap["C"]), "pointsH": zk.fixed_base_g1(G1B, trap["Hs"]),
        "pointsIC": zk.fixed_base_g1(G1B, trap["IC"]),
        "vk_alpha1": u8(zk.g1_mul(G1B, alpha)), "vk_beta1": u8(zk.g1_mul(G1B, beta)), "vk_beta2": u8(zk.g2_mul(G2B, beta)),
        "vk_gamma2": u8(zk.g2_mul(G2B, gamma)), "vk_delta1": u8(zk.g1_mul(G1B, delta)), "vk_delta2": u8(zk.g2_mul(G2B, delta)),
    }
    wit = np.frombuffer(b"".join(bn.int_to_le32(x) for x in w), dtype=np.uint8)
    return wl, wit, trap, w


def zkey_bytes(wl):
    b = lambda name: np.asarray(wl[name]).tobytes()
    sec2 = (struct.pack("<I", 32) + bn.int_to_le32(bn.Q_MOD) + struct.pack("<I", 32) + bn.int_to_le32(R_MOD)
            + struct.pack("<III", wl["nVars"], wl["nPublic"], wl["domainSize"])
            + b("vk_alpha1") + b("vk_beta1") + b("vk_beta2") + b("vk_gamma2") + b("vk_delta1") + b("vk_delta2"))
    return g.write_binfile(b"zkey", 1, [(1, struct.pack("<I", 1)), (2, sec2), (3, b("pointsIC")), (4, b("coefs")),
                                        (5, b("pointsA")), (6, b("pointsB1")), (7, b("pointsB2")), (8, b("pointsC")),
                                        (9, b("pointsH")), (10, bytes(68))])


def wtns_bytes(wl, wit):
    sec1 = struct.pack("<I", 32) + bn.int_to_le32(R_MOD) + struct.pack("<I", wl["nVars"])
    return g.write_binfile(b"wtns", 2, [(1, sec1), (2, np.asarray(wit).tobytes())])


def expected_proof_dlogs(trap, n_public, w, r, s):
    """(a, b, c): discrete logs of pi_a (G1), pi_b (G2), pi_c (G1) — the arithmetic of
    oracle.groth16_ref.trapdoor_check without its three scalar multiplications."""
    tau, alpha, beta, gamma, delta = trap["toxic"]
    At, Bt, Ct, K = trap["At"], trap["Bt"], trap["Ct"], trap["K"]
    dot = lambda v, lo=0: sum(x * wi for x, wi in zip(v[lo:], w[lo:])) % R_MOD
    a = (alpha + dot(At) + r * delta) % R_MOD
    b = (beta + dot(Bt) + s * delta) % R_MOD
    hz = (dot(At) * dot(Bt) - dot(Ct)) % R_MOD
    dinv = pow(delta, -1, R_MOD)
    c = ((dot(K, n_public + 1) + hz) * dinv + s * a + r * b - r * s % R_MOD * delta) % R_MOD
    return a, b, c
