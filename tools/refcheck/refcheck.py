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


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("prover", nargs="?", help="path to a real rapidsnark `prover` binary")
    ap.add_argument("--ours", action="store_true", help="check this repository's own prover binary instead (needs a GPU)")
    args = ap.parse_args()
    if not args.prover and not args.ours:
        ap.error("give the path of a rapidsnark prover binary, or --ours")
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
