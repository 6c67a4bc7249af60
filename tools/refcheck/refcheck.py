#!/usr/bin/env python3
"""Parity kit: pin tests/golden/* against a REAL rapidsnark binary (last row).

The build container cannot compile the reference's hot path (its `depends/ffiasm` submodule is empty
and nasm is absent), so this repository's parity status is "unpinned at the reference boundary":
every golden proof.json was produced by the repository's own oracle.  Whoever has a real
`rapidsnark` prover (iden3/rapidsnark-old `build/prover`, dynamically linked against libsodium) can
close that gap with one command:

    python3 tools/refcheck/refcheck.py /path/to/rapidsnark/build/prover

For every fixture under tests/golden/ that holds circuit.zkey + witness.wtns + meta.json it
  1. builds tools/refcheck/randombytes_shim.c (fixed r, s instead of randombytes_buf),
  2. runs  LD_PRELOAD=librandshim.so  prover circuit.zkey witness.wtns proof.json public.json
     with the fixture's (r, s),
  3. compares proof.json and public.json byte for byte with the committed goldens.
Exit status 0 = every fixture identical.  Nothing here is imported or executed by the product.

`--ours` runs this repository's own `rapidsnark-old_amd/prover` the same way (through its
ZKHIP_FIXED_R/S hook instead of the shim) — a self-test of the kit on a GPU box.

`--files circuit.zkey witness.wtns` pins files that are NOT fixtures — the ones snarkjs makes (reference
README.md:44-58; the full command sequence for circom's Multiplier2 is in INTEGRATION.md section 7):
the rapidsnark binary proves them under the shim with a fixed (r, s) (`--r`, `--s`: decimal, < 2^248),
and its proof.json / public.json are compared with `--with oracle` (default: the C restatement,
oracle/c/zk_oracle.c, on the CPU — no GPU needed) or `--with ours` (this repository's `prover` on a GPU box).
With `--ours --files ...` and no binary the two sides are this repository's prover and its oracle.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def le_hex(x):
    return int(x).to_bytes(32, "little").hex()


def run_prover(binary, ours, zkey, wtns, r, s, out_dir, tag):
    """One CLI run with (r, s) fixed: the shim for a rapidsnark binary, ZKHIP_FIXED_R/S for this repository's prover."""
    env = dict(os.environ)
    if ours:
        env["ZKHIP_FIXED_R"], env["ZKHIP_FIXED_S"] = le_hex(r), le_hex(s)
    else:
        shim = os.path.join(out_dir, "librandshim.so")
        if not os.path.exists(shim):
            subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-o", shim, os.path.join(HERE, "randombytes_shim.c")])
        env["LD_PRELOAD"] = shim + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
        env["ZKREF_R"], env["ZKREF_S"] = le_hex(r), le_hex(s)
    pj, qj = os.path.join(out_dir, tag + ".proof.json"), os.path.join(out_dir, tag + ".public.json")
    res = subprocess.run([binary, zkey, wtns, pj, qj], env=env, capture_output=True, text=True)
    if res.returncode != 0:
        raise SystemExit("%s failed (rc %d): %s" % (binary, res.returncode, res.stderr.strip()[-300:]))
    return open(pj, "rb").read(), open(qj, "rb").read()


def oracle_jsons(zkey, wtns, r, s):
    """proof.json / public.json of the C restatement (oracle/c/zk_oracle.c) on the CPU; the JSON text comes from the
    library's host-only zk_proof_to_json / zk_public_to_json (no GPU involved)."""
    sys.path.insert(0, ROOT)
    from oracle import c_oracle as co, groth16_ref as g
    import rapidsnark_old_amd as zk
    wt = g.read_wtns(open(wtns, "rb").read())
    vals = b"".join(int(v).to_bytes(32, "little") for v in wt["witness"])
    view = co.ZkeyView(open(zkey, "rb").read())
    proof = co.prove(view, vals, r, s)
    n_public = view.v.nPublic
    return zk.proof_to_json(proof).encode(), zk.public_to_json(vals, n_public).encode()


def pin_files(args):
    zkey, wtns = args.files
    r, s = int(args.r), int(args.s)
    if not (r < (1 << 248) and s < (1 << 248)):
        raise SystemExit("(r, s) must fit the 31 bytes the reference draws (src/groth16.cpp:216-217)")
    tmp = tempfile.mkdtemp(prefix="refcheck_")
    ours_bin = os.path.join(ROOT, "rapidsnark-old_amd", "prover")
    if args.prover:
        left_name, left = args.prover, run_prover(args.prover, False, zkey, wtns, r, s, tmp, "ref")
    else:
        left_name, left = ours_bin, run_prover(ours_bin, True, zkey, wtns, r, s, tmp, "ours")
    if args.other == "ours" and args.prover:
        right_name, right = ours_bin, run_prover(ours_bin, True, zkey, wtns, r, s, tmp, "ours")
    else:
        right_name, right = "oracle/c/zk_oracle.c", oracle_jsons(zkey, wtns, r, s)
    ok_p, ok_q = left[0] == right[0], left[1] == right[1]
    print("%s vs %s on %s: proof.json %s   public.json %s" % (left_name, right_name, os.path.basename(zkey),
                                                             "IDENTICAL" if ok_p else "DIFFERS", "IDENTICAL" if ok_q else "DIFFERS"))
    if not ok_p:
        print("  left : %s\n  right: %s" % (left[0][:400], right[0][:400]))
    print("parity on these files: %s" % ("PINNED" if ok_p and ok_q else "MISMATCH"))
    return 0 if ok_p and ok_q else 1


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("prover", nargs="?", help="path to a real rapidsnark `prover` binary")
    ap.add_argument("--ours", action="store_true", help="check this repository's own prover binary instead (needs a GPU)")
    ap.add_argument("--files", nargs=2, metavar=("ZKEY", "WTNS"), help="pin these files (e.g. snarkjs-made) instead of tests/golden/*")
    ap.add_argument("--with", dest="other", choices=("oracle", "ours"), default="oracle", help="--files: what the binary's output is compared with")
    ap.add_argument("--r", default="123456789123456789123456789", help="--files: r (decimal, < 2^248)")
    ap.add_argument("--s", default="987654321987654321987654321", help="--files: s (decimal, < 2^248)")
    args = ap.parse_args()
    if not args.prover and not args.ours:
        ap.error("give the path of a rapidsnark prover binary, or --ours")
    if args.files:
        return pin_files(args)
    tmp = tempfile.mkdtemp(prefix="refcheck_")
    env_base = dict(os.environ)
    if args.ours:
        binary = os.path.join(ROOT, "rapidsnark-old_amd", "prover")
    else:
        binary = args.prover
        shim = os.path.join(tmp, "librandshim.so")
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-o", shim, os.path.join(HERE, "randombytes_shim.c")])
        env_base["LD_PRELOAD"] = shim + (":" + env_base["LD_PRELOAD"] if env_base.get("LD_PRELOAD") else "")
    bad = 0
    for name in sorted(os.listdir(GOLDEN)):
        d = os.path.join(GOLDEN, name)
        need = [os.path.join(d, f) for f in ("circuit.zkey", "witness.wtns", "meta.json", "proof.json", "public.json")]
        if not all(os.path.isfile(f) for f in need):
            continue
        meta = json.load(open(need[2]))
        r, s = int(meta["r"]), int(meta["s"])
        assert r < (1 << 248) and s < (1 << 248), "golden (r, s) must fit the 31 bytes the reference draws"
        env = dict(env_base)
        if args.ours:
            env["ZKHIP_FIXED_R"], env["ZKHIP_FIXED_S"] = le_hex(r), le_hex(s)
        else:
            env["ZKREF_R"], env["ZKREF_S"] = le_hex(r), le_hex(s)
        pj, qj = os.path.join(tmp, name + ".proof.json"), os.path.join(tmp, name + ".public.json")
        res = subprocess.run([binary, need[0], need[1], pj, qj], env=env, capture_output=True, text=True)
        if res.returncode != 0:
            print("%-14s prover failed (rc %d): %s" % (name, res.returncode, res.stderr.strip()[-200:]))
            bad += 1
            continue
        ok_p = open(pj, "rb").read() == open(need[3], "rb").read()
        ok_q = open(qj, "rb").read() == open(need[4], "rb").read()
        print("%-14s proof.json %s   public.json %s" % (name, "IDENTICAL" if ok_p else "DIFFERS", "IDENTICAL" if ok_q else "DIFFERS"))
        bad += (not ok_p) + (not ok_q)
    print("parity vs %s: %s" % (binary, "PINNED (all fixtures byte-identical)" if not bad else "%d mismatches" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
