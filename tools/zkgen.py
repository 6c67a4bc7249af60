#!/usr/bin/env python3
"""zkgen — write a trapdoor-VALID Groth16 key at a benchmark size (needs a GPU).

    python tools/zkgen.py <log2n> <outdir> [--npublic N] [--seed S] [--circuit-like | --semaphore-like] [--prove]

Writes <outdir>/circuit.zkey, witness.wtns, verification_key.json, toxic.json (see
rapidsnark-old_amd/zkgen.py).  --prove also runs the one-shot CLI `prover` on the written files with a
fixed (r, s), writes proof.json / public.json, and checks the proof against the discrete logs
computed from the toxic waste (pairing-free trapdoor check).  Off-box:
    snarkjs groth16 verify verification_key.json public.json proof.json
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("log2n", type=int)
    ap.add_argument("outdir")
    ap.add_argument("--npublic", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--prove", action="store_true")
    ap.add_argument("--circuit-like", action="store_true", help="nVars = 3/4 of the domain + 5, 80 %% boolean signals, all-zero table rows (zkgen.generate)")
    ap.add_argument("--semaphore-like", action="store_true", help="the shape class of Semaphore / iden3 auth: chains of x^5 S-box rounds between Merkle-style muxes, "
                                                                  "nearly every signal full-size (zkgen.generate; use --npublic 4)")
    args = ap.parse_args()
    import rapidsnark_old_amd as zk
    from rapidsnark_old_amd import zkgen, synth
    t = time.time()
    key = zkgen.generate(args.log2n, args.npublic, args.seed, circuit_like=args.circuit_like, semaphore_like=args.semaphore_like)
    t_gen = time.time() - t
    t = time.time()
    zkgen.write_all(key, args.outdir)
    print("generated 2^%d key in %.1f s (nVars %d, nCoefs %d), wrote files in %.1f s" % (args.log2n, t_gen, key["nVars"], key["nCoefs"], time.time() - t))
    if args.prove:
        r, s = 0x0123456789ABCDEF, (1 << 200) + 12345
        le = lambda x: int(x).to_bytes(32, "little").hex()
        env = dict(os.environ, ZKHIP_FIXED_R=le(r), ZKHIP_FIXED_S=le(s))
        f = lambda name: os.path.join(args.outdir, name)
        t = time.time()
        subprocess.check_call([os.path.join(ROOT, "rapidsnark-old_amd", "prover"), f("circuit.zkey"), f("witness.wtns"), f("proof.json"), f("public.json")], env=env)
        print("prover CLI: %.2f s wall" % (time.time() - t))
        a, b, c = zkgen.expected_proof_dlogs(key, r, s)
        want = zk.g1_mul(synth.g1_gen_bytes(), a) + zk.g2_mul(synth.g2_gen_bytes(), b) + zk.g1_mul(synth.g1_gen_bytes(), c)
        ok = open(f("proof.json")).read() == zk.proof_to_json(want)
        print("trapdoor check of proof.json:", "PASS" if ok else "FAIL")
        return 0 if ok else 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
