"""Wall time of the one-shot CLI (`prover <zkey> <wtns> <proof.json> <public.json>`) on a synthetic
2^k zkey written to disk, with the phase times of ZKHIP_VERBOSE=1.
    python tools/cli_timing.py [log2n=20] [dir=/tmp/zk_cli] [runs=2]"""
import os, struct, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def binfile(path, magic, version, sections):
    with open(path, "wb") as f:
        f.write(magic + struct.pack("<II", version, len(sections)))
        for typ, data in sections:
            f.write(struct.pack("<IQ", typ, len(data)))
            f.write(data)


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    d = sys.argv[2] if len(sys.argv) > 2 else "/tmp/zk_cli"
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    os.makedirs(d, exist_ok=True)
    wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
    w = synth.make_witness(k, seed=1)
    b = lambda name: np.asarray(wl[name]).tobytes()
    sec2 = (struct.pack("<I", 32) + Q_MOD.to_bytes(32, "little") + struct.pack("<I", 32) + R_MOD.to_bytes(32, "little")
            + struct.pack("<III", wl["nVars"], wl["nPublic"], wl["domainSize"])
            + b("vk_alpha1") + b("vk_beta1") + b("vk_beta2") + b("vk_beta2") + b("vk_delta1") + b("vk_delta2"))
    zpath, wpath = os.path.join(d, "c.zkey"), os.path.join(d, "w.wtns")
    binfile(zpath, b"zkey", 1, [(1, struct.pack("<I", 1)), (2, sec2), (3, bytes(64 * (wl["nPublic"] + 1))), (4, b("coefs")),
                                (5, b("pointsA")), (6, b("pointsB1")), (7, b("pointsB2")), (8, b("pointsC")), (9, b("pointsH")), (10, bytes(68))])
    binfile(wpath, b"wtns", 2, [(1, struct.pack("<I", 32) + R_MOD.to_bytes(32, "little") + struct.pack("<I", wl["nVars"])), (2, np.asarray(w).tobytes())])
    print("zkey %.1f MB, wtns %.1f MB" % (os.path.getsize(zpath) / 1e6, os.path.getsize(wpath) / 1e6), flush=True)
    exe = os.path.join(ROOT, "rapidsnark-old_amd", "prover")
    def evict(path):          # drop the file's clean pages from the page cache: the next run reads it from disk ("cold")
        fd = os.open(path, os.O_RDONLY)
        os.fsync(fd)
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        os.close(fd)

    for mode in ("0", "1"):
        for i in range(runs):
            if i == 0:
                evict(zpath)
                evict(wpath)
            print("=== %s page cache" % ("COLD" if i == 0 else "warm"))
            env = dict(os.environ, ZKHIP_VERBOSE="1", ZKHIP_PRECOMP=mode, ZKHIP_T0=repr(time.time()))
            t0 = time.perf_counter()
            out = subprocess.run([exe, zpath, wpath, os.path.join(d, "p.json"), os.path.join(d, "q.json")], capture_output=True, text=True, env=env)
            dt = time.perf_counter() - t0
            print("--- ZKHIP_PRECOMP=%s run %d: wall %.2f s rc %d" % (mode, i, dt, out.returncode))
            print(out.stderr.strip())
    os.remove(zpath)
    os.remove(wpath)


if __name__ == "__main__":
    main()
