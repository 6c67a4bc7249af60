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
