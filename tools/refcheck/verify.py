#!/usr/bin/env python3
"""Groth16 verification of a proof.json / public.json pair (the check snarkjs `groth16 verify` makes; rapidsnark itself only
proves, /root/reference/src/main_prover.cpp:74-93):

    python3 tools/refcheck/verify.py proof.json public.json verification_key.json
    python3 tools/refcheck/verify.py proof.json public.json circuit.zkey          (the key's own alpha, beta, gamma, delta, IC)

Independent of the toxic waste and of every line of the prover: usable on a real Semaphore / iden3-auth key whose trapdoor
nobody knows, on a proof made by this repository's `prover`, or on one made by a real rapidsnark.  Exit status 0 = the
pairing equation holds.  Uses oracle/pairing.py (pure Python, ~1 s); test infrastructure, never imported by the product.
It does NOT lift the "parity unpinned" status of DESIGN.md section 2 — only a real rapidsnark run through refcheck.py can:
a valid proof and a bit-identical proof are different claims."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bn254 as bn, groth16_ref as g, pairing     # noqa: E402


def g1(j):
    x, y = int(j[0]), int(j[1])
    if len(j) > 2 and int(j[2]) == 0:
        return None
    return (x, y)


def g2(j):
    if len(j) > 2 and int(j[2][0]) == 0 and int(j[2][1]) == 0:
        return None
    return ((int(j[0][0]), int(j[0][1])), (int(j[1][0]), int(j[1][1])))


def vk_from_zkey(path):
    """alpha1, beta2, gamma2, delta2 (section 2) and IC (section 3) of a .zkey — only these two sections are read, so a
    multi-gigabyte key costs nothing (layout: /root/reference/src/zkey_utils.cpp:17-52)."""
    import struct
    with open(path, "rb") as f:
        if f.read(4) != b"zkey":
            raise SystemExit("not a zkey file")
        _version, nsec = struct.unpack("<II", f.read(8))
        want = {}
        for _ in range(nsec):
            sid, size = struct.unpack("<IQ", f.read(12))
            if sid in (1, 2, 3) and sid not in want:
                want[sid] = f.read(size)
            else:
                f.seek(size, 1)
    if struct.unpack_from("<I", want[1], 0)[0] != 1:
        raise SystemExit("zkey file is not groth16")
    s2 = want[2]
    n8q = struct.unpack_from("<I", s2, 0)[0]
    pos = 4 + n8q
    n8r = struct.unpack_from("<I", s2, pos)[0]
    pos += 4 + n8r + 12
    alpha1 = bn.g1_from_bytes(s2[pos:pos + 64]); pos += 64 + 64          # (beta1 skipped)
    beta2 = bn.g2_from_bytes(s2[pos:pos + 128]); pos += 128
    gamma2 = bn.g2_from_bytes(s2[pos:pos + 128]); pos += 128 + 64        # (delta1 skipped)
    delta2 = bn.g2_from_bytes(s2[pos:pos + 128])
    ic = [bn.g1_from_bytes(want[3][i:i + 64]) for i in range(0, len(want[3]), 64)]
    return {"alpha1": alpha1, "beta2": beta2, "gamma2": gamma2, "delta2": delta2, "IC": ic}


def load_vk(path):
    with open(path, "rb") as f:
        magic = f.read(4)
    if magic == b"zkey":
        return vk_from_zkey(path)
    j = json.load(open(path))
    if j.get("protocol", "groth16") != "groth16" or j.get("curve", "bn128") not in ("bn128", "bn254"):
        raise SystemExit("verification key is not groth16 / bn128")
    return {"alpha1": g1(j["vk_alpha_1"]), "beta2": g2(j["vk_beta_2"]), "gamma2": g2(j["vk_gamma_2"]), "delta2": g2(j["vk_delta_2"]),
            "IC": [g1(p) for p in j["IC"]]}


def verify_files(proof_path, public_path, vk_path):
    pj = json.load(open(proof_path))
    pub = json.load(open(public_path))
    pub = [] if pub is None else [int(x) for x in pub]       # `null` when nPublic == 0 (the reference's quirk)
    proof = (g1(pj["pi_a"]), g2(pj["pi_b"]), g1(pj["pi_c"]))
    return pairing.groth16_verify(load_vk(vk_path), pub, proof)


def main():
    if len(sys.argv) != 4:
        raise SystemExit(__doc__)
    ok = verify_files(*sys.argv[1:4])
    print("OK: the proof verifies" if ok else "INVALID: the pairing equation does not hold")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
