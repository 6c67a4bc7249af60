"""Same-box A/B of the one-shot CLI's way out: default (_exit once the files are written) against ZKHIP_CLEAN_EXIT=1
(prover destroyed, HIP exit handlers run).  python tools/cli_exit_ab.py [log2n=22] [runs=5]"""
import os, struct, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth
from tools.cli_timing import binfile, R_MOD, Q_MOD

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = "/tmp/zk_cli_ab"
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
exe = os.path.join(ROOT, "rapidsnark-old_amd", "prover")
res = {"default": [], "ZKHIP_CLEAN_EXIT=1": []}
for i in range(runs + 1):
    for name in res:
        env = dict(os.environ)
        if name != "default":
            env["ZKHIP_CLEAN_EXIT"] = "1"
        t0 = time.perf_counter()
        rc = subprocess.run([exe, zpath, wpath, os.path.join(d, "p.json"), os.path.join(d, "q.json")], capture_output=True, env=env).returncode
        dt = time.perf_counter() - t0
        assert rc == 0
        if i:
            res[name].append(dt)          # (run 0 warms the page cache)
for name, v in res.items():
    print("2^%d one-shot CLI wall, %-20s median %.3f s   all: %s" % (k, name, sorted(v)[len(v) // 2], " ".join("%.3f" % x for x in v)))
os.remove(zpath)
os.remove(wpath)
