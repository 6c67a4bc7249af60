#!/usr/bin/env python3
"""proverServer in throughput mode (BASELINE configs[4] shape): N concurrent /input requests against a
trapdoor-valid 2^k key, proofs/s through the REST API.

    python tools/server_bench.py [log2n=16] [requests=64] [workers=0] [route=input|witness] [key=random|semaphore]

route = input: POST /input/:circuit, the reference's route (a witness-generator process per request: here a stub that
copies the satisfying witness); route = witness: POST /witness/:circuit with the .wtns image as the body (no process, no
files).  The client keeps one HTTP/1.1 connection per thread (keep-alive).  Every proof is compared with the one the toxic
waste predicts.

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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


import http.client as httpc
import threading

_tls = threading.local()


def http(port, method, path, body=None):
    """one keep-alive connection per client thread"""
    c = getattr(_tls, "conn", None)
    for attempt in range(2):
        if c is None:
            c = _tls.conn = httpc.HTTPConnection("127.0.0.1", port, timeout=60)
        try:
            c.request(method, path, body=body)
            r = c.getresponse()
            data = r.read()
            if r.status == 503:
                raise BlockingIOError("queue full")
            return data
        except (httpc.HTTPException, ConnectionError, OSError) as exc:
            if isinstance(exc, BlockingIOError):
                raise
            c.close()
            c = _tls.conn = None
            if attempt:
                raise


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    nreq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    workers = sys.argv[3] if len(sys.argv) > 3 else "0"
    route = sys.argv[4] if len(sys.argv) > 4 else "input"
    shape = sys.argv[5] if len(sys.argv) > 5 else "random"
    from rapidsnark_old_amd import zkgen
    d = tempfile.mkdtemp(prefix="zksrv_")
    # key = semaphore: the Semaphore / iden3-auth SHAPE class (4 public signals, S-box chains between Merkle-style muxes, nearly
    # every signal full-size) on a trapdoor-valid zkgen key — a proxy: no real Semaphore zkey exists in this image
    key = zkgen.generate(k, 4, seed=1, semaphore_like=True) if shape == "semaphore" else zkgen.generate(k, 2, seed=1)
    zkgen.write_all(key, d)
    os.rename(os.path.join(d, "circuit.zkey"), os.path.join(d, "auth.zkey"))
    os.makedirs(os.path.join(d, "build"))
    gen = os.path.join(d, "build", "auth")
    open(gen, "w").write("#!/bin/sh\ncp %s \"$2\"\n" % os.path.join(d, "witness.wtns"))
    os.chmod(gen, os.stat(gen).st_mode | stat.S_IEXEC)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    # fixed (r, s): every returned proof is compared with the one the toxic waste predicts (pairing-free trapdoor check)
    import rapidsnark_old_amd as zk
    from rapidsnark_old_amd import synth
    r, s_ = 0x0F1E2D3C4B5A6978, (1 << 231) + 4242
    a, b, c = zkgen.expected_proof_dlogs(key, r, s_)
    want = zk.proof_to_json(zk.g1_mul(synth.g1_gen_bytes(), a) + zk.g2_mul(synth.g2_gen_bytes(), b) + zk.g1_mul(synth.g1_gen_bytes(), c))
    le = lambda x: int(x).to_bytes(32, "little").hex()
    env = dict(os.environ, ZKHIP_QUEUE=str(nreq + 8), ZKHIP_WORKERS=workers, ZKHIP_WITNESS_THREADS="8", ZKHIP_FIXED_R=le(r), ZKHIP_FIXED_S=le(s_))
    srv = subprocess.Popen([os.path.join(ROOT, "rapidsnark-old_amd", "proverServer"), str(port), os.path.join(d, "auth.zkey")], cwd=d, env=env,
                           stdout=subprocess.DEVNULL, stderr=open(os.environ["SERVER_LOG"], "w") if os.environ.get("SERVER_LOG") else subprocess.DEVNULL)
    try:
        for _ in range(600):
            try:
                http(port, "GET", "/status"); break
            except Exception:
                time.sleep(0.1)
        body = b'{"in": "1"}' if route == "input" else open(os.path.join(d, "witness.wtns"), "rb").read()
        post = "/input/auth" if route == "input" else "/witness/auth"
        for warm in range(4):       # replicas' lazy per-slot allocations
            job = json.loads(http(port, "POST", post, body))["job"]
            while json.loads(http(port, "GET", "/status/%d" % job))["status"] == "busy":
                time.sleep(0.002)
        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(16) as ex:
            jobs = list(ex.map(lambda _: json.loads(http(port, "POST", post, body))["job"], range(nreq)))
        ok = verified = 0
        for job in jobs:
            while True:
                doc = json.loads(http(port, "GET", "/status/%d" % job))
                if doc["status"] != "busy":
                    break
                time.sleep(0.001)
            ok += doc["status"] == "success"
            verified += doc.get("proof") == want
        dt = time.perf_counter() - t0
        print(json.dumps({"key": ("zkgen --semaphore-like proxy (nPublic 4, nVars %d; no real Semaphore / iden3-auth zkey in the image)" % key["nVars"]) if shape == "semaphore" else "zkgen random R1CS",
                          "log2n": k, "requests": nreq, "workers": workers, "route": post, "succeeded": ok, "proofs_equal_to_the_trapdoor_prediction": verified, "seconds": round(dt, 3),
                          "proofs_per_s": round(nreq / dt, 1), "ms_per_proof": round(dt / nreq * 1e3, 2)}))
    finally:
        if srv.poll() is not None:
            print("SERVER DIED with code", srv.returncode)
        srv.terminate(); srv.wait(10)


if __name__ == "__main__":
    main()
