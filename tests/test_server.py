"""`proverServer <port> <zkey>...` — the REST shell of the reference
(src/main_proofserver.cpp:11-45, src/proverapi.cpp, src/fullprover.cpp): routes, payload shapes,
job state machine, witness-generator hand-off (./build/<circuit> ./build/input_<circuit>.json
./build/<circuit>.wtns)."""
import json
import os
import shutil
import socket
import stat
import subprocess
import sys
import time
import urllib.error
import urllib.request

import pytest

from conftest import ROOT, golden_bytes, golden_json, golden_path

SERVER = os.path.join(ROOT, "rapidsnark-old_amd", "proverServer")


def test_usage_and_exit_code():
    r = subprocess.run([SERVER], capture_output=True, text=True)
    assert r.returncode == 255
    assert r.stderr == "Invalid number of parameters:\nUsage: proverServer <port> <circuit1.zkey> <circuit2.zkey> ... <circuitN.zkey> \n"
    assert subprocess.run([SERVER, "9080"], capture_output=True).returncode == 255


def test_bad_zkey_is_an_error_not_a_crash(tmp_path):
    r = subprocess.run([SERVER, "0", golden_path("multiplier2", "witness.wtns")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 255 and "Invalid file type. It should be zkey" in r.stderr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _http(port, method, path, body=None):
    req = urllib.request.Request("http://127.0.0.1:%d%s" % (port, path), data=body, method=method,
                                 headers={"Content-Type": "application/json", "Accept": "application/json"})
    try:
        with urllib.request.urlopen(req, timeout=30) as resp:
            return resp.status, resp.read(), resp.headers.get("Content-Type")
    except urllib.error.HTTPError as e:
        return e.code, e.read(), e.headers.get("Content-Type")


def _le_hex(x):
    return int(x).to_bytes(32, "little").hex()


@pytest.mark.gpu
def test_rest_flow_matches_golden(tmp_path):
    names = ["r1cs_n8", "r1cs_n64"]
    build = tmp_path / "build"
    build.mkdir()
    zkeys = []
    for n in names:
        z = tmp_path / (n + ".zkey")                       # circuit name = file stem
        shutil.copy(golden_path(n, "circuit.zkey"), z)
        zkeys.append(str(z))
        gen = build / n                                     # stand-in for the circom witness generator
        gen.write_text("#!/bin/sh\necho generating $1\ncp %s \"$2\"\n" % golden_path(n, "witness.wtns"))
        gen.chmod(gen.stat().st_mode | stat.S_IEXEC)
    bad = build / "broken"
    meta = golden_json("r1cs_n64", "meta.json")
    port = _free_port()
    env = dict(os.environ, ZKHIP_FIXED_R=_le_hex(meta["r"]), ZKHIP_FIXED_S=_le_hex(meta["s"]))
    srv = subprocess.Popen([SERVER, str(port)] + zkeys, cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        for _ in range(600):
            try:
                st, body, ctype = _http(port, "GET", "/status")
                break
            except (ConnectionError, urllib.error.URLError):
                assert srv.poll() is None, srv.stderr.read().decode()
                time.sleep(0.1)
        assert st == 200 and json.loads(body) == {"status": "ready"} and ctype == "application/json"
        assert _http(port, "POST", "/start")[0] == 200 and _http(port, "POST", "/stop")[0] == 200
        assert _http(port, "POST", "/cancel")[0] == 200          # nothing running: no-op
        assert _http(port, "GET", "/nope")[0] == 404

        def run(circuit, body):
            assert _http(port, "POST", "/input/" + circuit, body)[0] == 200
            for _ in range(3000):
                st = json.loads(_http(port, "GET", "/status")[1])
                if st["status"] != "busy":
                    return st
                time.sleep(0.01)
            raise AssertionError("stuck busy")

        st = run("r1cs_n64", b'{"a": "1", "b": ["2", 3]}')
        assert st["status"] == "success"
        assert st["proof"] == golden_bytes("r1cs_n64", "proof.json").decode()       # a STRING holding JSON (fullprover.cpp:232)
        assert st["pubData"] == golden_bytes("r1cs_n64", "public.json").decode()
        assert (tmp_path / "build" / "input_r1cs_n64.json").exists()
        raw = _http(port, "GET", "/status")[1].decode()
        assert raw.startswith('{"proof":"{\\"pi_a\\":') and raw.endswith(',"status":"success"}')   # nlohmann key order
        # second circuit on the same server
        st = run("r1cs_n8", b"{}")
        assert st["status"] == "success" and json.loads(st["proof"])["protocol"] == "groth16"
        assert st["pubData"] == golden_bytes("r1cs_n8", "public.json").decode()
        # malformed body: the reference dies on the uncaught parse error (Q3); here the job fails
        st = run("r1cs_n8", b'{"a": ')
        assert st["status"] == "failed" and "JSON" in st["error"]
        st = run("nosuchcircuit", b"{}")
        assert st["status"] == "failed" and "unknown circuit" in st["error"]
        assert srv.poll() is None
        st = run("r1cs_n64", b"[]")                              # still serving after the failures
        assert st["status"] == "success"
    finally:
        srv.terminate()
        srv.wait(10)


def _start_server(tmp_path, names, env_extra):
    build = tmp_path / "build"
    build.mkdir(exist_ok=True)
    zkeys = []
    for n in names:
        z = tmp_path / (n + ".zkey")
        shutil.copy(golden_path(n, "circuit.zkey"), z)
        zkeys.append(str(z))
        gen = build / n                                     # stand-in for the circom witness generator: argv as the reference passes it
        gen.write_text("#!/bin/sh\necho generating $1\nsleep 0.05\ncp %s \"$2\"\n" % golden_path(n, "witness.wtns"))
        gen.chmod(gen.stat().st_mode | stat.S_IEXEC)
    port = _free_port()
    env = dict(os.environ, **env_extra)
    srv = subprocess.Popen([SERVER, str(port)] + zkeys, cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    for _ in range(600):
        try:
            _http(port, "GET", "/status")
            return srv, port
        except (ConnectionError, urllib.error.URLError):
            assert srv.poll() is None, srv.stderr.read().decode()
            time.sleep(0.1)
    raise AssertionError("server did not come up")


@pytest.mark.gpu
@pytest.mark.parametrize("batch", ["1", "4", "8"])
def test_throughput_mode_concurrent_requests_all_golden(tmp_path, batch):
    """BASELINE configs[4] shape: ZKHIP_QUEUE + ZKHIP_WORKERS — many /input requests fired at once, every
    one of them answered with the golden proof of ITS circuit (fixed r, s), none dropped or replaced.
    Two worker replicas share the box's one GPU (ZKHIP_WORKERS=0,0): one dispatcher thread per replica,
    up to eight submissions in flight on each, witness generators running beside them; ZKHIP_BATCH: every
    job that is ready for the same circuit when a dispatcher takes one rides in the same submission
    (zk_prove_batch_submit), 1 = one job per submission."""
    import concurrent.futures
    names = ["r1cs_n8", "r1cs_n64", "r1cs_n256"]
    meta = golden_json("r1cs_n64", "meta.json")            # the fixtures share (r, s)? no: use each fixture's own below
    metas = {n: golden_json(n, "meta.json") for n in names}
    assert len({(m["r"], m["s"]) for m in metas.values()}) >= 1
    # fixed (r, s) come from the environment, one pair per server: run one server per (r, s) group
    groups = {}
    for n in names:
        groups.setdefault((metas[n]["r"], metas[n]["s"]), []).append(n)
    for (r, s), group in groups.items():
        sub = tmp_path / ("g%d" % (hash((r, s)) & 0xffff))
        sub.mkdir()
        srv, port = _start_server(sub, group, {"ZKHIP_FIXED_R": _le_hex(r), "ZKHIP_FIXED_S": _le_hex(s), "ZKHIP_QUEUE": "64",
                                               "ZKHIP_WORKERS": "0,0", "ZKHIP_WITNESS_THREADS": "3", "ZKHIP_BATCH": batch})
        try:
            assert json.loads(_http(port, "GET", "/status")[1]) == {"status": "ready"}
            reqs = [group[i % len(group)] for i in range(24)]

            def fire(circuit):
                st, body, ctype = _http(port, "POST", "/input/" + circuit, b'{"in": 1}')
                assert st == 200 and ctype == "application/json"
                return circuit, json.loads(body)["job"]

            with concurrent.futures.ThreadPoolExecutor(8) as ex:
                issued = list(ex.map(fire, reqs))
            assert len({j for _, j in issued}) == len(reqs)                    # every request got its own job
            for circuit, job in issued:
                for _ in range(3000):
                    doc = json.loads(_http(port, "GET", "/status/%d" % job)[1])
                    if doc["status"] != "busy":
                        break
                    time.sleep(0.01)
                assert doc["status"] == "success", doc
                assert doc["proof"] == golden_bytes(circuit, "proof.json").decode()
                assert doc["pubData"] == golden_bytes(circuit, "public.json").decode()
            # failures stay per job; the queue keeps serving
            bad = json.loads(_http(port, "POST", "/input/" + group[0], b'{"x": ')[1])["job"]
            unknown = json.loads(_http(port, "POST", "/input/nosuch", b"{}")[1])["job"]
            good = json.loads(_http(port, "POST", "/input/" + group[0], b"{}")[1])["job"]
            docs = {}
            for job in (bad, unknown, good):
                for _ in range(3000):
                    docs[job] = json.loads(_http(port, "GET", "/status/%d" % job)[1])
                    if docs[job]["status"] != "busy":
                        break
                    time.sleep(0.01)
            assert docs[bad]["status"] == "failed" and "JSON" in docs[bad]["error"]
            assert docs[unknown]["status"] == "failed" and "unknown circuit" in docs[unknown]["error"]
            assert docs[good]["status"] == "success"
            assert json.loads(_http(port, "GET", "/status/999999")[1])["status"] == "failed"
            assert json.loads(_http(port, "GET", "/status")[1])["status"] == "success"     # the most recent job
            assert srv.poll() is None
        finally:
            srv.terminate()
            srv.wait(10)


@pytest.mark.gpu
def test_queue_full_is_503(tmp_path):
    srv, port = _start_server(tmp_path, ["r1cs_n8"], {"ZKHIP_QUEUE": "1", "ZKHIP_WITNESS_THREADS": "1"})
    try:
        codes = [_http(port, "POST", "/input/r1cs_n8", b"{}")[0] for _ in range(12)]
        assert codes[0] == 200 and 503 in codes           # one witness thread, 50 ms generator: the one-deep queue overflows
        assert set(codes) <= {200, 503}
    finally:
        srv.terminate()
        srv.wait(10)


@pytest.mark.gpu
def test_results_are_kept_for_their_owner_and_unfinished_jobs_are_never_dropped(tmp_path):
    """Throughput mode remembers jobs for GET /status/<id>: the most recent ZKHIP_KEEP_RESULTS FINISHED ones, and every job that
    is still waiting or running.  (The first version dropped the oldest job once 4096 were known, whatever its state: a client that
    submitted 8192 requests before it polled found half of them "unknown".)  Here: one witness thread, a 50 ms generator, a result
    memory of TWO — twelve jobs submitted at once are all "busy" or "success" when polled straight away, never unknown; once all
    are done every result is there, and the next job's arrival trims the memory to the newest finished one beside itself."""
    srv, port = _start_server(tmp_path, ["r1cs_n8"], {"ZKHIP_QUEUE": "16", "ZKHIP_WITNESS_THREADS": "1", "ZKHIP_KEEP_RESULTS": "2", "ZKHIP_WORKERS": "0"})
    try:
        jobs = []
        for _ in range(12):
            code, body, _ = _http(port, "POST", "/input/r1cs_n8", b"{}")
            assert code == 200
            jobs.append(json.loads(body)["job"])
        first = [json.loads(_http(port, "GET", "/status/%d" % j)[1]) for j in jobs]
        assert all(d["status"] in ("busy", "success") for d in first[2:]), first      # (jobs 1-2 may already be finished AND replaced)
        assert sum(d["status"] == "busy" for d in first) >= 6
        for _ in range(2000):                                   # until the newest one is done
            if json.loads(_http(port, "GET", "/status/%d" % jobs[-1])[1])["status"] != "busy":
                break
            time.sleep(0.01)
        last = [json.loads(_http(port, "GET", "/status/%d" % j)[1]) for j in jobs]
        assert all(d["status"] == "success" for d in last), last          # nothing was dropped while it was unfinished
        # the memory is trimmed when the next job arrives: finished results beyond the two newest make room, the new job stays
        code, body, _ = _http(port, "POST", "/input/r1cs_n8", b"{}")
        extra = json.loads(body)["job"]
        for _ in range(2000):
            doc = json.loads(_http(port, "GET", "/status/%d" % extra)[1])
            if doc["status"] != "busy":
                break
            time.sleep(0.01)
        assert doc["status"] == "success"
        after = [json.loads(_http(port, "GET", "/status/%d" % j)[1]) for j in jobs]
        assert after[0] == {"error": "unknown job", "status": "failed"} and after[-1]["status"] == "success"
        assert sum(d["status"] == "success" for d in after) == 1
    finally:
        srv.terminate()
        srv.wait(10)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,shape", [("1", "circuit"), ("4", "semaphore")])
def test_semaphore_class_throughput_every_proof_checked(zk, tmp_path, batch, shape):
    """BASELINE configs[4] on a key of its size AND shape class: no Semaphore / iden3-auth zkey exists in the image, so the key
    is a trapdoor-VALID R1CS with a 2^15 domain (rapidsnark_old_amd.zkgen; Semaphore is ~2^14..2^16 constraints) — once the
    circuit-like preset (80 % boolean signals), once the semaphore-like one (4 public signals, S-box chains between
    Merkle-style muxes, nearly every signal full-size).
    proverServer in throughput mode — two replicas on the box's GPU, queue of 64, ZKHIP_BATCH 1 and 4 — takes 64
    concurrent /input requests; with fixed (r, s) EVERY returned proof must be (i) the bytes the one-shot CLI writes for
    the same files and (ii) the proof whose discrete logs follow from the toxic waste (pairing-free trapdoor check).
    Reference shape: src/fullprover.cpp:69-101,154-159, src/main_proofserver.cpp:32-40."""
    import concurrent.futures
    from rapidsnark_old_amd import synth, zkgen
    k, nreq = 15, 64
    key = zkgen.generate(k, 4, seed=3, semaphore_like=True) if shape == "semaphore" else zkgen.generate(k, 2, seed=3, circuit_like=True)
    assert key["nVars"] == 3 * (1 << k) // 4 + 5            # both presets: nVars is never the domain size
    zkgen.write_all(key, str(tmp_path))
    os.rename(tmp_path / "circuit.zkey", tmp_path / "auth.zkey")
    r, s = 0x0F1E2D3C4B5A6978, (1 << 231) + 4242
    env_rs = {"ZKHIP_FIXED_R": _le_hex(r), "ZKHIP_FIXED_S": _le_hex(s)}
    # (ii) expected proof from the toxic waste alone: discrete logs in Fr, then three scalar multiplications and the JSON
    # text by the ORACLE (oracle/bn254.py, oracle/groth16_ref.py) — no curve arithmetic of the product in the expectation
    from oracle import bn254 as obn, groth16_ref as og
    a, b, c = zkgen.expected_proof_dlogs(key, r, s)
    want = og.proof_to_json((obn.G1.mul(obn.G1.gen, a), obn.G2.mul(obn.G2.gen, b), obn.G1.mul(obn.G1.gen, c)))
    # (i) the one-shot CLI on the same files
    cli = subprocess.run([os.path.join(ROOT, "rapidsnark-old_amd", "prover"), str(tmp_path / "auth.zkey"), str(tmp_path / "witness.wtns"),
                          str(tmp_path / "proof.json"), str(tmp_path / "public.json")], env=dict(os.environ, **env_rs), capture_output=True, text=True, timeout=300)
    assert cli.returncode == 0, cli.stderr
    cli_proof, cli_public = open(tmp_path / "proof.json").read(), open(tmp_path / "public.json").read()
    assert cli_proof == want
    build = tmp_path / "build"
    build.mkdir()
    gen = build / "auth"                                    # stand-in for the circom witness generator (same argv as the reference passes)
    gen.write_text("#!/bin/sh\ncp %s \"$2\"\n" % (tmp_path / "witness.wtns"))
    gen.chmod(gen.stat().st_mode | stat.S_IEXEC)
    port = _free_port()
    env = dict(os.environ, ZKHIP_QUEUE="64", ZKHIP_WORKERS="0,0", ZKHIP_WITNESS_THREADS="4", ZKHIP_BATCH=batch, **env_rs)
    srv = subprocess.Popen([SERVER, str(port), str(tmp_path / "auth.zkey")], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        for _ in range(1200):
            try:
                _http(port, "GET", "/status")
                break
            except (ConnectionError, urllib.error.URLError):
                assert srv.poll() is None, srv.stderr.read().decode()
                time.sleep(0.1)

        def fire(_):
            for _try in range(200):                          # 503 = queue full: the client retries, nothing is dropped
                st, body, _ct = _http(port, "POST", "/input/auth", b'{"in": "1"}')
                if st == 200:
                    return json.loads(body)["job"]
                assert st == 503
                time.sleep(0.005)
            raise AssertionError("queue never drained")

        with concurrent.futures.ThreadPoolExecutor(16) as ex:
            jobs = list(ex.map(fire, range(nreq)))
        assert len(set(jobs)) == nreq
        for job in jobs:
            for _ in range(6000):
                doc = json.loads(_http(port, "GET", "/status/%d" % job)[1])
                if doc["status"] != "busy":
                    break
                time.sleep(0.005)
            assert doc["status"] == "success", doc
            assert doc["proof"] == cli_proof and doc["proof"] == want
            assert doc["pubData"] == cli_public
        assert srv.poll() is None
    except Exception:
        srv.terminate()
        try:
            err = srv.communicate(timeout=10)[1].decode(errors="replace")
        except Exception:                                   # noqa: BLE001
            err = "(no stderr)"
        print("proverServer exit code %s, stderr tail:\n%s" % (srv.returncode, err[-3000:]))
        raise
    finally:
        srv.terminate()
        srv.wait(10)


@pytest.mark.gpu
def test_keep_alive_connections_and_in_process_witness(tmp_path):
    """The front end serves several requests per connection (HTTP/1.1 keep-alive, several connections at once on its
    worker threads), and POST /witness/:circuit takes the .wtns image itself — no generator process, no files in
    ./build — and yields the same golden proof as the generator hand-off."""
    import http.client
    name = "r1cs_n64"
    meta = golden_json(name, "meta.json")
    srv, port = _start_server(tmp_path, [name], {"ZKHIP_FIXED_R": _le_hex(meta["r"]), "ZKHIP_FIXED_S": _le_hex(meta["s"]), "ZKHIP_QUEUE": "32",
                                                 "ZKHIP_WORKERS": "0", "ZKHIP_HTTP_THREADS": "4"})
    try:
        conns = [http.client.HTTPConnection("127.0.0.1", port, timeout=30) for _ in range(3)]     # three live connections, four workers
        wt = golden_bytes(name, "witness.wtns")
        jobs = []
        for i in range(12):
            c = conns[i % 3]
            if i % 2:
                c.request("POST", "/witness/" + name, body=wt, headers={"Content-Type": "application/octet-stream"})
            else:
                c.request("POST", "/input/" + name, body=b'{"in": 1}', headers={"Content-Type": "application/json"})
            r = c.getresponse()
            body = r.read()
            assert r.status == 200 and r.getheader("Connection") == "keep-alive", (r.status, body)
            jobs.append(json.loads(body)["job"])
        assert len(set(jobs)) == 12
        for i, job in enumerate(jobs):
            c = conns[i % 3]
            for _ in range(3000):
                c.request("GET", "/status/%d" % job)
                doc = json.loads(c.getresponse().read())
                if doc["status"] != "busy":
                    break
                time.sleep(0.005)
            assert doc["status"] == "success", doc
            assert doc["proof"] == golden_bytes(name, "proof.json").decode() and doc["pubData"] == golden_bytes(name, "public.json").decode()
        # the per-job files are gone on every path; a bad image fails its own job only
        left = [f for f in os.listdir(tmp_path / "build") if f.endswith(".wtns") or f.startswith("input_")]
        assert left == [], left
        c = conns[0]
        c.request("POST", "/witness/" + name, body=b"not a wtns file")
        bad = json.loads(c.getresponse().read())["job"]
        c.request("GET", "/status/%d" % bad)
        doc = json.loads(c.getresponse().read())
        assert doc["status"] == "failed" and "Invalid file type" in doc["error"]
        c.request("GET", "/status", headers={"Connection": "close"})
        r = c.getresponse()
        assert r.status == 200 and r.getheader("Connection") == "close"
        r.read()
        # a chunked body is refused (the front end only frames by Content-Length) and the connection is closed, not misparsed
        c2 = http.client.HTTPConnection("127.0.0.1", port, timeout=30)
        c2.request("POST", "/input/" + name, body=iter([b'{"in":', b' 1}']), headers={"Transfer-Encoding": "chunked"})
        r = c2.getresponse()
        assert r.status == 501 and r.getheader("Connection") == "close"
        r.read()
        assert srv.poll() is None
    finally:
        srv.terminate()
        srv.wait(10)


@pytest.mark.gpu
def test_cancel_in_throughput_mode_drops_what_has_not_reached_a_gpu(tmp_path):
    """POST /cancel with a queue (src/fullprover.cpp:206-214 aborts the one running proof; with a queue it drops every job
    that has not reached a GPU: waiting for a witness generator, inside one, or ready for a dispatcher).  The server goes on
    serving afterwards."""
    name = "r1cs_n64"
    meta = golden_json(name, "meta.json")
    build = tmp_path / "build"
    build.mkdir()
    z = tmp_path / (name + ".zkey")
    shutil.copy(golden_path(name, "circuit.zkey"), z)
    gen = build / name
    gen.write_text("#!/bin/sh\nsleep 0.4\ncp %s \"$2\"\n" % golden_path(name, "witness.wtns"))      # a slow generator
    gen.chmod(gen.stat().st_mode | stat.S_IEXEC)
    port = _free_port()
    env = dict(os.environ, ZKHIP_FIXED_R=_le_hex(meta["r"]), ZKHIP_FIXED_S=_le_hex(meta["s"]), ZKHIP_QUEUE="16", ZKHIP_WORKERS="0",
               ZKHIP_WITNESS_THREADS="1")
    srv = subprocess.Popen([SERVER, str(port), str(z)], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        for _ in range(600):
            try:
                _http(port, "GET", "/status")
                break
            except (ConnectionError, urllib.error.URLError):
                assert srv.poll() is None, srv.stderr.read().decode()
                time.sleep(0.1)
        jobs = [json.loads(_http(port, "POST", "/input/" + name, b"{}")[1])["job"] for _ in range(5)]
        time.sleep(0.1)                                   # job 1 is inside the generator, 2..5 wait for it
        assert _http(port, "POST", "/cancel")[0] == 200
        docs = {}
        for job in jobs:
            for _ in range(400):
                docs[job] = json.loads(_http(port, "GET", "/status/%d" % job)[1])
                if docs[job]["status"] != "busy":
                    break
                time.sleep(0.01)
        assert [docs[j]["status"] for j in jobs] == ["aborted"] * 5, docs
        after = json.loads(_http(port, "POST", "/input/" + name, b"{}")[1])["job"]
        for _ in range(600):
            doc = json.loads(_http(port, "GET", "/status/%d" % after)[1])
            if doc["status"] != "busy":
                break
            time.sleep(0.01)
        assert doc["status"] == "success" and doc["proof"] == golden_bytes(name, "proof.json").decode()
        left = [f for f in os.listdir(build) if f.endswith(".wtns") or f.startswith("input_")]
        assert left == [], left
        assert srv.poll() is None
    finally:
        srv.terminate()
        srv.wait(10)


@pytest.mark.gpu
def test_server_falls_back_to_plain_tables_when_the_precomputed_ones_do_not_fit(zk, tmp_path):
    """proverServer creates its provers with window-precomputed tables by default (13 x the table memory).  Where those do
    not fit the GPU's free memory — here: a 2^20 key while another process holds all but 7 GiB of the HBM; in the field:
    2^26 constraints — the prover is created with the tables as they are in the zkey instead, says so, and proves the same bytes."""
    import struct
    import numpy as np
    from rapidsnark_old_amd import synth
    from tools.cli_timing import binfile, R_MOD, Q_MOD
    k = 20
    wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
    w = synth.make_witness(k, seed=3)
    b = lambda name: np.asarray(wl[name]).tobytes()
    sec2 = (struct.pack("<I", 32) + Q_MOD.to_bytes(32, "little") + struct.pack("<I", 32) + R_MOD.to_bytes(32, "little")
            + struct.pack("<III", wl["nVars"], wl["nPublic"], wl["domainSize"])
            + b("vk_alpha1") + b("vk_beta1") + b("vk_beta2") + b("vk_beta2") + b("vk_delta1") + b("vk_delta2"))
    zpath, wpath = tmp_path / "big.zkey", tmp_path / "big.wtns"
    binfile(str(zpath), b"zkey", 1, [(1, struct.pack("<I", 1)), (2, sec2), (3, bytes(64 * (wl["nPublic"] + 1))), (4, b("coefs")),
                                     (5, b("pointsA")), (6, b("pointsB1")), (7, b("pointsB2")), (8, b("pointsC")), (9, b("pointsH")), (10, bytes(68))])
    binfile(str(wpath), b"wtns", 2, [(1, struct.pack("<I", 32) + R_MOD.to_bytes(32, "little") + struct.pack("<I", wl["nVars"])), (2, np.asarray(w).tobytes())])
    del wl
    fixed = {"ZKHIP_FIXED_R": _le_hex(12345), "ZKHIP_FIXED_S": _le_hex(67890)}
    # the reference bytes: the one-shot CLI (tables as in the zkey) with the same (r, s)
    cli = subprocess.run([os.path.join(ROOT, "rapidsnark-old_amd", "prover"), str(zpath), str(wpath), str(tmp_path / "p.json"), str(tmp_path / "q.json")],
                         env=dict(os.environ, **fixed), capture_output=True, text=True, timeout=300)
    assert cli.returncode == 0, cli.stderr
    hog = subprocess.Popen([sys.executable, "-c",
                            "import time, torch\nfree, _ = torch.cuda.mem_get_info()\nx = torch.empty(free - (7 << 30), dtype=torch.uint8, device='cuda')\n"
                            "print('held', flush=True)\ntime.sleep(300)\n"], stdout=subprocess.PIPE, text=True)
    srv = None
    try:
        assert hog.stdout.readline().strip() == "held"
        (tmp_path / "build").mkdir(exist_ok=True)
        port = _free_port()
        env = dict(os.environ, ZKHIP_QUEUE="4", ZKHIP_WORKERS="0", **fixed)
        env.pop("ZKHIP_PRECOMP", None)
        srv = subprocess.Popen([SERVER, str(port), str(zpath)], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        for _ in range(900):
            try:
                _http(port, "GET", "/status")
                break
            except (ConnectionError, urllib.error.URLError):
                assert srv.poll() is None, srv.stderr.read().decode()
                time.sleep(0.1)
        code, body, _ = _http(port, "POST", "/witness/big", wpath.read_bytes())
        assert code == 200
        job = json.loads(body)["job"]
        for _ in range(3000):
            doc = json.loads(_http(port, "GET", "/status/%d" % job)[1])
            if doc["status"] != "busy":
                break
            time.sleep(0.01)
        assert doc["status"] == "success", doc
        assert doc["proof"] == (tmp_path / "p.json").read_text() and doc["pubData"] == (tmp_path / "q.json").read_text()
        srv.terminate()
        err = srv.communicate(timeout=20)[1].decode()
        srv = None
        assert "do not fit the GPU's free memory" in err, err
    finally:
        hog.kill()
        if srv is not None:
            srv.kill()
