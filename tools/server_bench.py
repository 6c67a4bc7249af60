#!/usr/bin/env python3
"""proverServer in throughput mode (BASELINE configs[4] shape): N concurrent /input requests against a
trapdoor-valid 2^k key, proofs/s through the REST API.

    python tools/server_bench.py [log2n=16] [requests=64] [workers=0]

No Semaphore / iden3-auth zkey or circom witness generator exists in this image: the key comes from
rapidsnark_old_amd.zkgen (a random R1CS of the same size class) and the "witness generator" is a stub that
copies the satisfying witness (its cost on a real deployment is host time the server overlaps with the GPU).
ZKHIP_WORKERS lists the GPUs (one replica + dispatcher each; a GPU may be listed twice), ZKHIP_QUEUE the depth."""
import concurrent.futures
import json
import os
import socket
import stat
import subprocess
import sys
import tempfile
import time
import urllib.request

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def http(port, method, path, body=None):
    req = urllib.request.Request("http://127.0.0.1:%d%s" % (port, path), data=body, method=method)
    with urllib.request.urlopen(req, timeout=60) as resp:
        return resp.read()


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    nreq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    workers = sys.argv[3] if len(sys.argv) > 3 else "0"
    from rapidsnark_old_amd import zkgen
    d = tempfile.mkdtemp(prefix="zksrv_")
    key = zkgen.generate(k, 2, seed=1)
    zkgen.write_all(key, d)
    os.rename(os.path.join(d, "circuit.zkey"), os.path.join(d, "auth.zkey"))
    os.makedirs(os.path.join(d, "build"))
    gen = os.path.join(d, "build", "auth")
    open(gen, "w").write("#!/bin/sh\ncp %s \"$2\"\n" % os.path.join(d, "witness.wtns"))
    os.chmod(gen, os.stat(gen).st_mode | stat.S_IEXEC)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ZKHIP_QUEUE=str(nreq + 8), ZKHIP_WORKERS=workers, ZKHIP_WITNESS_THREADS="8")
    srv = subprocess.Popen([os.path.join(ROOT, "rapidsnark-old_amd", "proverServer"), str(port), os.path.join(d, "auth.zkey")], cwd=d, env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        for _ in range(600):
            try:
                http(port, "GET", "/status"); break
            except Exception:
                time.sleep(0.1)
        body = b'{"in": "1"}'
        for warm in range(4):       # replicas' lazy per-slot allocations
            job = json.loads(http(port, "POST", "/input/auth", body))["job"]
            while json.loads(http(port, "GET", "/status/%d" % job))["status"] == "busy":
                time.sleep(0.002)
        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(16) as ex:
            jobs = list(ex.map(lambda _: json.loads(http(port, "POST", "/input/auth", body))["job"], range(nreq)))
        ok = 0
        for job in jobs:
            while True:
                doc = json.loads(http(port, "GET", "/status/%d" % job))
                if doc["status"] != "busy":
                    break
                time.sleep(0.001)
            ok += doc["status"] == "success"
        dt = time.perf_counter() - t0
        print(json.dumps({"log2n": k, "requests": nreq, "workers": workers, "succeeded": ok, "seconds": round(dt, 3),
                          "proofs_per_s": round(nreq / dt, 1), "ms_per_proof": round(dt / nreq * 1e3, 2)}))
    finally:
        srv.terminate(); srv.wait(10)


if __name__ == "__main__":
    main()
