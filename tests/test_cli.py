"""The `prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>` executable
(reference src/main_prover.cpp:23-103): argv, messages, exit codes, output bytes."""
import os
import subprocess

import pytest

from conftest import CIRCUITS, ROOT, golden_bytes, golden_json, golden_path

PROVER = os.path.join(ROOT, "rapidsnark-old_amd", "prover")


def run(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([PROVER, *args], capture_output=True, text=True, errors="replace", env=e, timeout=300)


def test_usage_and_exit_code():
    r = run()
    assert r.returncode == 255                                   # `return -1` (main_prover.cpp:28)
    assert r.stderr == "Invalid number of parameters:\nUsage: prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>\n"
    assert run("a", "b", "c").returncode == 255


def test_bad_inputs_are_reported_not_aborted(tmp_path):
    # quirk Q1: the reference throws a pointer here and aborts; we print the same text and exit -1
    r = run(golden_path("multiplier2", "witness.wtns"), golden_path("multiplier2", "witness.wtns"), str(tmp_path / "p"), str(tmp_path / "q"))
    assert r.returncode == 255 and r.stderr == "Invalid file type. It should be zkey and it us wtns\n"
    r = run("/nonexistent.zkey", "x", "y", "z")
    assert r.returncode == 255 and r.stderr.startswith("open")
    bad = bytearray(golden_bytes("multiplier2", "circuit.zkey"))
    bad[4] = 7
    (tmp_path / "v.zkey").write_bytes(bytes(bad))
    r = run(str(tmp_path / "v.zkey"), golden_path("multiplier2", "witness.wtns"), str(tmp_path / "p"), str(tmp_path / "q"))
    assert r.returncode == 255 and r.stderr == "Invalid version. It should be <=1 and it us 7\n"
    trunc = golden_bytes("multiplier2", "circuit.zkey")[:300]
    (tmp_path / "t.zkey").write_bytes(trunc)
    r = run(str(tmp_path / "t.zkey"), golden_path("multiplier2", "witness.wtns"), str(tmp_path / "p"), str(tmp_path / "q"))
    assert r.returncode == 255 and "end of file" in r.stderr
    # witness of another circuit: quirk Q8 (OOB read in the reference) is an error here
    r = run(golden_path("r1cs_n8", "circuit.zkey"), golden_path("multiplier2", "witness.wtns"), str(tmp_path / "p"), str(tmp_path / "q"))
    assert r.returncode == 255 and "nVars" in r.stderr


def _le_hex(x):
    return int(x).to_bytes(32, "little").hex()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CIRCUITS)
def test_cli_output_bytes(tmp_path, name):
    meta = golden_json(name, "meta.json")
    pj, qj = tmp_path / "proof.json", tmp_path / "public.json"
    r = run(golden_path(name, "circuit.zkey"), golden_path(name, "witness.wtns"), str(pj), str(qj),
            env={"ZKHIP_FIXED_R": _le_hex(meta["r"]), "ZKHIP_FIXED_S": _le_hex(meta["s"])})
    assert r.returncode == 0, r.stderr
    assert pj.read_bytes() == golden_bytes(name, "proof.json")          # compact, no trailing newline
    assert qj.read_bytes() == golden_bytes(name, "public.json")


@pytest.mark.gpu
def test_cli_random_rs_differs(tmp_path):
    outs = []
    for i in range(2):
        pj = tmp_path / ("p%d.json" % i)
        r = run(golden_path("r1cs_n8", "circuit.zkey"), golden_path("r1cs_n8", "witness.wtns"), str(pj), str(tmp_path / "q.json"))
        assert r.returncode == 0, r.stderr
        outs.append(pj.read_bytes())
    assert outs[0] != outs[1]


@pytest.mark.gpu
def test_parity_kit_self_check_on_our_prover():
    """tools/refcheck/refcheck.py --ours: the kit that pins tests/golden/* against a real rapidsnark binary,
    run against this repository's own `prover` (every fixture's proof.json / public.json byte-identical)."""
    import subprocess
    import sys
    from conftest import ROOT
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "refcheck", "refcheck.py"), "--ours"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("IDENTICAL") == 10 and "PINNED" in res.stdout


@pytest.mark.gpu
def test_parity_kit_files_mode_on_our_prover():
    """refcheck.py --ours --files ZKEY WTNS: the one-step pin for files that are not fixtures (what snarkjs writes), self-checked with
    this repository's prover on one side and the C restatement on the other, at the kit's default (r, s)."""
    import subprocess
    import sys
    from conftest import ROOT
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "refcheck", "refcheck.py"), "--ours", "--files",
                          golden_path("multiplier2", "circuit.zkey"), golden_path("multiplier2", "witness.wtns")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("IDENTICAL") == 2 and "PINNED" in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0,0,0,0,0", "0,0,0"])
def test_cli_over_several_devices(tmp_path, devices):
    """ZKHIP_DEVICES=...: the reference's own argv, one proof split over several GPUs (zk_multi_prover; here
    every shard on the box's one GPU).  2/8 shards partition the chain, 3 replicate it; same output bytes."""
    name = "r1cs_n256"
    meta = golden_json(name, "meta.json")
    le = lambda x: int(x).to_bytes(32, "little").hex()
    r = run(golden_path(name, "circuit.zkey"), golden_path(name, "witness.wtns"), str(tmp_path / "p.json"), str(tmp_path / "q.json"),
            env={"ZKHIP_DEVICES": devices, "ZKHIP_FIXED_R": le(meta["r"]), "ZKHIP_FIXED_S": le(meta["s"])})
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "p.json").read_bytes() == golden_bytes(name, "proof.json")
    assert (tmp_path / "q.json").read_bytes() == golden_bytes(name, "public.json")


# Every environment variable INTEGRATION.md section 5 documents for the library / CLI, with a non-default value: none of them
# may change the proof.  The retired measurement probes (compiled out of the shipped library: -DZK_PROBES) are set too —
# ZKHIP_GATHER_MASK used to give WRONG proofs with exit code 0; in the default build it must have no effect at all.
_DOCUMENTED = [{"ZKHIP_VERBOSE": "1"}, {"ZKHIP_SERIAL": "1"}, {"ZKHIP_PRECOMP": "1"}, {"ZKHIP_PRECOMP": "0"}, {"ZKHIP_PRECOMP": "2"},
               {"ZKHIP_PRECOMP": "2", "ZKHIP_DEVICES": "0,0"}, {"ZKHIP_DEVICE": "0"},
               {"ZKHIP_LANES": "1"}, {"ZKHIP_LANES": "3", "ZKHIP_LANE_STREAMS": "1"}, {"ZKHIP_TAIL": "0"}, {"ZKHIP_TAIL": "2"},
               {"ZKHIP_GRAPH": "1"}, {"ZKHIP_BATCH_ABC": "0"}, {"ZKHIP_BATCH_ABC": "1", "ZKHIP_PRECOMP": "1"},
               {"ZKHIP_DEVICES": "0,0,0,0", "ZKHIP_REPLICATED_CHAIN": "1"}, {"GPU_MAX_HW_QUEUES": "8"}, {"ZKHIP_CLEAN_EXIT": "1"},
               {"ZKHIP_SPARSE_WITNESS": "1", "ZKHIP_PRECOMP": "1"}]
_RETIRED_PROBES = {"ZKHIP_GATHER_MASK": "0xff", "ZKHIP_ACC_ROUND_WGS": "1", "ZKHIP_ACC_CHUNK_MIN": "4", "ZKHIP_ACC_CHUNK_MAX": "8",
                   "ZKHIP_STAGE_SYNC": "1", "ZKHIP_NTT_THREADS": "512", "ZKHIP_NTT_TILE": "8", "ZKHIP_REDUCE_BITS": "0",
                   "ZKHIP_REDUCE_CHUNK": "2", "ZKHIP_S1_PRIO": "1", "ZKHIP_LONE_ORDER": "1"}


@pytest.mark.gpu
@pytest.mark.parametrize("extra", _DOCUMENTED + [_RETIRED_PROBES], ids=lambda e: ",".join("%s=%s" % kv for kv in sorted(e.items()))[:60])
def test_environment_switches_never_change_the_proof(tmp_path, extra):
    name = "r1cs_n256"
    meta = golden_json(name, "meta.json")
    env = {"ZKHIP_FIXED_R": _le_hex(meta["r"]), "ZKHIP_FIXED_S": _le_hex(meta["s"])}
    env.update(extra)
    r = run(golden_path(name, "circuit.zkey"), golden_path(name, "witness.wtns"), str(tmp_path / "p.json"), str(tmp_path / "q.json"), env=env)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "p.json").read_bytes() == golden_bytes(name, "proof.json")
    assert (tmp_path / "q.json").read_bytes() == golden_bytes(name, "public.json")


def test_shipped_library_reads_no_probe_variable():
    """The default build of libzkhip.so does not even contain the names of the measurement probes."""
    blob = open(os.path.join(ROOT, "rapidsnark-old_amd", "libzkhip.so"), "rb").read()
    for name in _RETIRED_PROBES:
        assert name.encode() not in blob, name


def test_damaged_files_end_in_an_error_exit_never_in_a_signal(tmp_path):
    """Truncated files, flipped header bytes, section sizes that point past the end of the file: the reference indexes
    sections blindly (quirks Q1/Q8); here every such input must end with exit code 255 and a message (or, when the damage
    only hit curve points or scalars, with a proof) — never with a signal.  Runs without a GPU too (the files are parsed
    before the device is asked for)."""
    import random
    rng = random.Random(20260928)
    zkey, wtns = golden_bytes("r1cs_n8", "circuit.zkey"), golden_bytes("r1cs_n8", "witness.wtns")
    cases = []
    for _ in range(14):
        cases.append(("z", zkey[:rng.randrange(0, len(zkey))]))
        cases.append(("w", wtns[:rng.randrange(0, len(wtns))]))
    for _ in range(16):                                      # one byte of the header / section table / first sections changed
        for kind, blob in (("z", zkey), ("w", wtns)):
            b = bytearray(blob)
            i = rng.randrange(0, min(len(b), 400))
            b[i] = rng.choice([0, 1, 0x7f, 0x80, 0xff, b[i] ^ 0x40])
            cases.append((kind, bytes(b)))
    for kind, blob in (("z", zkey), ("w", wtns)):            # every section size field in turn: huge
        pos, nsec = 12, int.from_bytes(blob[8:12], "little")
        for _ in range(nsec):
            size = int.from_bytes(blob[pos + 4:pos + 12], "little")
            b = bytearray(blob)
            b[pos + 4:pos + 12] = (1 << 62).to_bytes(8, "little")
            cases.append((kind, bytes(b)))
            pos += 12 + size
    for n, (kind, blob) in enumerate(cases):
        zp, wp = tmp_path / "c.zkey", tmp_path / "w.wtns"
        zp.write_bytes(blob if kind == "z" else zkey)
        wp.write_bytes(blob if kind == "w" else wtns)
        r = run(str(zp), str(wp), str(tmp_path / "p.json"), str(tmp_path / "q.json"))
        assert r.returncode in (0, 255), (n, kind, len(blob), r.returncode, r.stderr[-300:])
        assert r.returncode == 0 or r.stderr.strip(), (n, kind, "silent failure")


@pytest.mark.gpu
def test_cli_reports_an_output_file_it_could_not_write(tmp_path):
    """The program leaves through _exit once both files are written: a proof that did NOT reach the disk must not exit 0."""
    r = run(golden_path("r1cs_n8", "circuit.zkey"), golden_path("r1cs_n8", "witness.wtns"), str(tmp_path / "no_such_dir" / "proof.json"), str(tmp_path / "q.json"))
    assert r.returncode == 255 and "could not write" in r.stderr
    r = run(golden_path("r1cs_n8", "circuit.zkey"), golden_path("r1cs_n8", "witness.wtns"), str(tmp_path / "p.json"), "/dev/full")
    assert r.returncode == 255 and "could not write /dev/full" in r.stderr
