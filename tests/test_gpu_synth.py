"""GPU parity on the synthetic benchmark family (BASELINE configs 2-3):
GPU chain tables vs the C oracle's; GPU prove() vs the C restatement bit-for-bit at mid
sizes; and the discrete-log identities that stay checkable at the full 2^20 / 2^22 sizes."""
import numpy as np
import pytest

from oracle import bn254 as bn, c_oracle as co

pytestmark = pytest.mark.gpu

G1B = bn.g1_to_bytes(bn.G1.gen)
G2B = bn.g2_to_bytes(bn.G2.gen)


def test_chain_tables_match_oracle(zk):
    p0, q = co.g1_mul(G1B, 1234567), co.g1_mul(G1B, 7654321)
    for n in (1, 63, 64, 65, 5000):
        assert zk.synth_chain_g1(n, p0, q).tobytes() == co.chainp_g1(n, p0, q).tobytes()
    p0, q = co.g2_mul(G2B, 1234567), co.g2_mul(G2B, 7654321)
    for n in (1, 65, 1000):
        assert zk.synth_chain_g2(n, p0, q).tobytes() == co.chainp_g2(n, p0, q).tobytes()
    assert zk.g1_mul(G1B, bn.R_MOD - 5) == co.g1_mul(G1B, bn.R_MOD - 5)
    assert zk.g2_mul(G2B, 99) == co.g2_mul(G2B, 99)


def _gpu_workload(zk, k):
    from rapidsnark_old_amd import synth
    return synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())


def _prover(zk, wl, **kw):
    import bench
    return bench.ProverFromView(zk, wl, device=0, shard_index=kw.get("shard_index", 0), shard_count=kw.get("shard_count", 1),
                                window_bits=kw.get("window_bits", 0), timings=False, precomp=kw.get("precomp", False))


@pytest.mark.parametrize("k", [10, 14, 16])
def test_synthetic_prove_bit_exact_vs_c_oracle(zk, k):
    import torch
    from rapidsnark_old_amd import synth
    wl = _gpu_workload(zk, k)
    wl_cpu = co.synth_workload(k)
    for name in ("pointsA", "pointsB1", "pointsB2", "pointsC", "pointsH", "coefs"):
        assert np.array_equal(np.asarray(wl[name]), np.asarray(wl_cpu[name])), name
    w = synth.make_witness(k, seed=3)
    view = co.ZkeyView(wl_cpu)
    p = _prover(zk, wl)
    wd = torch.from_numpy(w).to("cuda:0")
    r, s = 0xabcdef0123456789, (1 << 247) + 12345
    assert p.prove_msm_dev(wd.data_ptr()) == co.prove_msm(view, w)
    assert p.prove_dev(wd.data_ptr(), r, s) == co.prove(view, w, r, s)


@pytest.mark.parametrize("wbits", [8, 11, 13])
def test_window_bits_do_not_change_results(zk, wbits):
    import torch
    from rapidsnark_old_amd import synth
    k = 12
    wl = _gpu_workload(zk, k)
    w = synth.make_witness(k)
    wd = torch.from_numpy(w).to("cuda:0")
    base = _prover(zk, wl).prove_msm_dev(wd.data_ptr())
    assert _prover(zk, wl, window_bits=wbits).prove_msm_dev(wd.data_ptr()) == base


def test_full_size_2p20_dlog_identities(zk):
    """BASELINE configs[1] size.  h is cross-checked against the C oracle (FFT pipeline on the
    host cores); the five MSMs and the assembled proof against their known discrete logs —
    independent of any MSM implementation."""
    import torch
    from rapidsnark_old_amd import synth
    k = 20
    wl = _gpu_workload(zk, k)
    w = synth.make_witness(k)
    h = co.compute_h(co.ZkeyView(wl), w)
    dl = synth.expected_msm_dlogs(wl, w, np.frombuffer(h, dtype=np.uint8))
    p = _prover(zk, wl)
    wd = torch.from_numpy(w).to("cuda:0")
    sums = p.prove_msm_dev(wd.data_ptr())
    assert sums[0:64] == co.g1_mul(G1B, dl["pih"])            # also proves the GPU's h == oracle h
    assert sums[64:128] == co.g1_mul(G1B, dl["pi_a"])
    assert sums[128:192] == co.g1_mul(G1B, dl["pib1"])
    assert sums[192:320] == co.g2_mul(G2B, dl["pi_b"])
    assert sums[320:384] == co.g1_mul(G1B, dl["pi_c"])
    r, s = 3141592653589793, 2718281828459045
    a, b, c = synth.expected_proof_dlogs(wl, dl, r, s)
    assert p.prove_dev(wd.data_ptr(), r, s) == co.g1_mul(G1B, a) + co.g2_mul(G2B, b) + co.g1_mul(G1B, c)
    # sharded (multi-GPU split emulated on one device) gives the same partial-sum total
    parts = [_prover(zk, wl, shard_index=i, shard_count=4).prove_msm_dev(wd.data_ptr()) for i in range(4)]
    assert p.prove_finish(parts) is not None
    import ctypes as C
    from rapidsnark_old_amd import lib as L
    arr = (L.zk_msm_sums * 4)(*[L.zk_msm_sums.from_buffer_copy(x) for x in parts])
    out = L.zk_proof()
    rb = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8)
    sb = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8)
    L.check(p.lib.zk_prove_finish(p.h, arr, 4, rb.ctypes.data, sb.ctypes.data, C.byref(out)))
    assert bytes(out) == co.g1_mul(G1B, a) + co.g2_mul(G2B, b) + co.g1_mul(G1B, c)


def _write_synth_files(tmp_path, wl, w):
    """A real .zkey / .wtns pair on disk for a synthetic workload (oracle's binfile writer)."""
    import struct
    from oracle import groth16_ref as g
    b = lambda k: np.asarray(wl[k]).tobytes()
    sec2 = (struct.pack("<I", 32) + bn.int_to_le32(bn.Q_MOD) + struct.pack("<I", 32) + bn.int_to_le32(bn.R_MOD)
            + struct.pack("<III", wl["nVars"], wl["nPublic"], wl["domainSize"])
            + b("vk_alpha1") + b("vk_beta1") + b("vk_beta2") + b("vk_beta2") + b("vk_delta1") + b("vk_delta2"))
    zkey = g.write_binfile(b"zkey", 1, [(1, struct.pack("<I", 1)), (2, sec2), (3, bytes(64 * (wl["nPublic"] + 1))), (4, b("coefs")),
                                       (5, b("pointsA")), (6, b("pointsB1")), (7, b("pointsB2")), (8, b("pointsC")), (9, b("pointsH")),
                                       (10, bytes(68))])
    sec1 = struct.pack("<I", 32) + bn.int_to_le32(bn.R_MOD) + struct.pack("<I", wl["nVars"])
    wtns = g.write_binfile(b"wtns", 2, [(1, sec1), (2, np.asarray(w).tobytes())])
    (tmp_path / "c.zkey").write_bytes(zkey)
    (tmp_path / "w.wtns").write_bytes(wtns)
    return str(tmp_path / "c.zkey"), str(tmp_path / "w.wtns"), len(zkey)


@pytest.mark.parametrize("k,kind", [(16, "uniform"), (18, "realistic")])
def test_cli_end_to_end_at_scale(zk, tmp_path, k, kind):
    """`prover` on a real multi-hundred-MB .zkey file (2^18: ~390 MB): mmap reader, CSR build,
    upload, prove, JSON — byte-for-byte against the C restatement for a fixed (r, s)."""
    import os
    import subprocess
    from conftest import ROOT
    from rapidsnark_old_amd import synth
    wl = co.synth_workload(k)
    w = synth.make_witness(k, seed=5, kind=kind)
    zpath, wpath, zbytes = _write_synth_files(tmp_path, wl, w)
    r, s = 0x1122334455667788, (1 << 200) + 99
    env = dict(os.environ, ZKHIP_FIXED_R=int(r).to_bytes(32, "little").hex(), ZKHIP_FIXED_S=int(s).to_bytes(32, "little").hex())
    out = subprocess.run([os.path.join(ROOT, "rapidsnark-old_amd", "prover"), zpath, wpath, str(tmp_path / "p.json"), str(tmp_path / "q.json")],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr
    want = co.prove(co.ZkeyView(wl), w, r, s)
    assert (tmp_path / "p.json").read_text() == zk.proof_to_json(want)
    assert (tmp_path / "q.json").read_text() == zk.public_to_json(w, 1)
    assert zbytes > (1 << k) * 300


def test_realistic_witness_msm_parity(zk):
    """Skewed scalars (80 % in {0,1}): the sort and the load-balanced accumulation must give the same
    MSM sums as the C restatement (bucket "1" of window 0 holds ~40 % of all entries)."""
    import torch
    from rapidsnark_old_amd import synth
    k = 16
    wl = _gpu_workload(zk, k)
    w = synth.make_witness(k, seed=9, kind="realistic")
    vals = np.frombuffer(w, dtype=np.uint8).reshape(-1, 32)
    assert (vals[:, 1:].max(axis=1) == 0).mean() > 0.7          # really skewed
    p = _prover(zk, wl)
    wd = torch.from_numpy(w).to("cuda:0")
    assert p.prove_msm_dev(wd.data_ptr()) == co.prove_msm(co.ZkeyView(co.synth_workload(k)), w)


@pytest.mark.parametrize("k,wbits", [(12, 0), (16, 0), (16, 17), (18, 20)])
def test_precomp_mode_matches_plain_mode(zk, k, wbits):
    """Window-precomputed tables (two-level sort, one bucket set, up to 2^19 buckets) vs the plain path."""
    import torch
    from rapidsnark_old_amd import synth
    wl = _gpu_workload(zk, k)
    for kind in ("uniform", "realistic"):
        w = synth.make_witness(k, seed=2, kind=kind)
        wd = torch.from_numpy(w).to("cuda:0")
        base = _prover(zk, wl).prove_msm_dev(wd.data_ptr())
        assert _prover(zk, wl, precomp=True, window_bits=wbits).prove_msm_dev(wd.data_ptr()) == base
