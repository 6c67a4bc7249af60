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
    from rapidsnark_old_amd import views
    return views.ProverFromView(zk, wl, device=0, shard_index=kw.get("shard_index", 0), shard_count=kw.get("shard_count", 1),
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
        # rows for every second window (ZK_FLAG_PRECOMP_HALF): odd windows add the neighbouring row into a second bucket set
        half = _prover(zk, wl, precomp=2, window_bits=wbits)
        assert half.prove_msm_dev(wd.data_ptr()) == base
        plan = half.info()
        assert plan["precomputed_tables"] == 2 and plan["bucket_sets_h"] == 2 and plan["table_rows_h"] == (plan["windows_h"] + 1) // 2


def test_fixed_base_batch_matches_host_scalar_mul(zk):
    """zk_fixed_base_g1/g2 against the host scalar multiplication (independent 4x64-bit code) and the
    oracle, including 0, 1, r-1 and a 256-bit value >= r."""
    import random
    rng = random.Random(99)
    ks = [0, 1, 2, bn.R_MOD - 1, bn.R_MOD, (1 << 256) - 1] + [rng.randrange(bn.R_MOD) for _ in range(70)]
    g1 = zk.fixed_base_g1(G1B, ks).tobytes()
    g2 = zk.fixed_base_g2(G2B, ks).tobytes()
    for i, k in enumerate(ks):
        assert g1[64 * i:64 * i + 64] == bn.g1_to_bytes(bn.G1.mul(bn.G1.gen, k % bn.R_MOD)), i
        assert g2[128 * i:128 * i + 128] == bn.g2_to_bytes(bn.G2.mul(bn.G2.gen, k % bn.R_MOD)), i


@pytest.mark.parametrize("k,precomp", [(10, False), (14, True), (16, True)])
def test_valid_key_at_scale_passes_trapdoor_check(zk, tmp_path, k, precomp):
    """a trapdoor-VALID key over a random satisfiable R1CS with a 2^k domain (tables from
    the GPU fixed-base kernel), a satisfying witness, and the proof identity of §8c item 2 checked in Fr:
    A = a*G1, B = b*G2, C = c*G1 with (a, b, c) computed from the toxic waste.  The same proof must come
    out of the C restatement (bit-exact) and of the one-shot CLI on the written .zkey / .wtns."""
    import os
    import subprocess
    import valid_key as vk
    from conftest import ROOT
    wl, wit, trap, w = vk.build(zk, k, seed=1000 + k)
    zpath, wpath = tmp_path / "valid.zkey", tmp_path / "valid.wtns"
    zpath.write_bytes(vk.zkey_bytes(wl))
    wpath.write_bytes(vk.wtns_bytes(wl, wit))
    r, s = 0x0123456789ABCDEF0123, (1 << 247) - 12345
    p = zk.Prover(str(zpath), precomp=precomp)
    proof = p.prove(wpath.read_bytes(), r=r, s=s)
    p.close()
    a, b, c = vk.expected_proof_dlogs(trap, wl["nPublic"], w, r, s)
    assert proof[0:64] == zk.g1_mul(G1B, a)
    assert proof[64:192] == zk.g2_mul(G2B, b)
    assert proof[192:256] == zk.g1_mul(G1B, c)
    assert proof == co.prove(co.ZkeyView(wl), wit, r, s)
    env = dict(os.environ, ZKHIP_FIXED_R=int(r).to_bytes(32, "little").hex(), ZKHIP_FIXED_S=int(s).to_bytes(32, "little").hex())
    out = subprocess.run([os.path.join(ROOT, "rapidsnark-old_amd", "prover"), str(zpath), str(wpath), str(tmp_path / "p.json"), str(tmp_path / "q.json")],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr
    assert (tmp_path / "p.json").read_text() == zk.proof_to_json(proof)
    assert (tmp_path / "q.json").read_text() == zk.public_to_json(wit, wl["nPublic"])


@pytest.mark.parametrize("k,n_vars,n_public", [(13, 5000, 7), (13, 8191, 1), (13, 12345, 40), (12, 4097, 0), (14, 16384 + 4096, 2)])
@pytest.mark.parametrize("precomp", [False, True, 2])
def test_irregular_shapes_bit_exact_vs_c_oracle(zk, k, n_vars, n_public, precomp):
    """Real circuits have nVars != domainSize and several public signals; the synthetic family has nVars = domainSize and
    one.  Mid-size keys with fewer / more signals than domain rows, nPublic from 0 to 40, odd table lengths: the GPU
    proof against the C restatement bit for bit (all five MSMs and the assembled proof), both table modes."""
    import torch
    from rapidsnark_old_amd import synth
    n = 1 << k
    big = _gpu_workload(zk, k + 1)                          # a pool of 2n valid points per table to cut the tables from
    wl = dict(_gpu_workload(zk, k))
    wl["nVars"], wl["nPublic"] = n_vars, n_public
    for name, width in (("pointsA", 64), ("pointsB1", 64), ("pointsB2", 128)):
        wl[name] = np.ascontiguousarray(np.asarray(big[name]).reshape(-1)[: n_vars * width])
    wl["pointsC"] = np.ascontiguousarray(np.asarray(big["pointsC"]).reshape(-1)[: (n_vars - n_public - 1) * 64])
    img = np.asarray(wl["coefs"]).copy()                    # u32 count + 44-byte records: fold the signal indices into [0, nVars)
    rec = img[4:].view(synth.COEF_DTYPE)
    rng = np.random.default_rng(k * 1000 + n_vars)
    rec["s"] = np.where(rec["s"] < n_vars, rec["s"], rec["s"] % n_vars) if n_vars <= n else rng.integers(0, n_vars, size=rec.shape[0], dtype=np.uint32)
    wl["coefs"] = img
    w = synth.random_fr_bytes(rng, n_vars).reshape(-1).copy()
    w[:32] = np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint8)
    view = co.ZkeyView(wl)
    p = _prover(zk, wl, precomp=precomp)
    wd = torch.from_numpy(w).to("cuda:0")
    r, s = 0x1357924680ACE, (1 << 200) + 99
    assert p.prove_msm_dev(wd.data_ptr()) == co.prove_msm(view, w)
    assert p.prove_dev(wd.data_ptr(), r, s) == co.prove(view, w, r, s)
    p.lib.zk_prover_destroy(p.h)


@pytest.mark.parametrize("precomp", [False, True, 2])
def test_circuit_shaped_key_bit_exact_vs_c_oracle(zk, precomp):
    """The circuit-shaped member of the family (synth.workload(shape="circuit"): nVars = 3/4 of the domain + 5, three public
    signals, ~30 % of the rows of A and of B1 / B2 at infinity) with the 80/15/5 witness — what bench.py's `also_realistic`
    leg times at 2^22 — at 2^14 against the C restatement: the five MSM sums, the assembled proof (host-witness entry point,
    as the reference's prove(wtns)), and the known-discrete-log prediction of the MSM sums."""
    import torch
    from rapidsnark_old_amd import synth
    k = 14
    wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes(), shape="circuit")
    w = synth.make_witness(k, seed=11, kind="realistic", n_vars=wl["nVars"])
    view = co.ZkeyView(wl)
    p = _prover(zk, wl, precomp=precomp)
    wd = torch.from_numpy(w).to("cuda:0")
    r, s = 0xC1AC017, (1 << 190) + 7
    sums = p.prove_msm_dev(wd.data_ptr())
    assert sums == co.prove_msm(view, w)
    want = synth.expected_msm_dlogs(wl, w, np.zeros(32, dtype=np.uint8))
    assert sums[64:128] == zk.g1_mul(G1B, want["pi_a"]) and sums[192:320] == zk.g2_mul(G2B, want["pi_b"]) and sums[320:384] == zk.g1_mul(G1B, want["pi_c"])
    ref = co.prove(view, w, r, s)
    assert p.prove_dev(wd.data_ptr(), r, s) == ref
    assert p.prove_host(w, r, s) == ref
    p.lib.zk_prover_destroy(p.h)


def test_sparse_witness_flag_changes_the_window_not_the_sums(zk):
    """ZK_FLAG_SPARSE_WITNESS: the four witness MSMs on a 16-bit window (sixteen table rows per point instead of fourteen at
    2^19, a sixteenth of the buckets) — more table memory, the same five MSM sums for a circuit-like witness and for a
    uniformly random one; MSM H keeps its window."""
    import torch
    from rapidsnark_old_amd import synth, views
    k = 19
    wl = _gpu_workload(zk, k)
    used = []
    sums = {}
    for sparse in (False, True):
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        p = views.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True, sparse_witness=sparse)
        torch.cuda.synchronize()
        used.append(free0 - torch.cuda.mem_get_info()[0])
        for kind in ("realistic", "uniform"):
            w = synth.make_witness(k, seed=4, kind=kind)
            wd = torch.from_numpy(w).to("cuda:0")
            sums[(sparse, kind)] = p.prove_msm_dev(wd.data_ptr())
        p.lib.zk_prover_destroy(p.h)
    assert sums[(True, "realistic")] == sums[(False, "realistic")] and sums[(True, "uniform")] == sums[(False, "uniform")]
    assert sums[(True, "realistic")] != sums[(True, "uniform")]
    assert used[1] > used[0] * 1.02, used          # 16 rows per point for A, B1, B2, C instead of 14 (the bucket arrays shrink: +3.6 % in all)


@pytest.mark.parametrize("precomp", [False, True, 2])
def test_proofs_submitted_beside_others_bit_exact_vs_c_oracle(zk, precomp):
    """From 2^17 a proof submitted while another one is in flight runs fewer, longer level-1 lanes (at least 128 entries each, one
    round of lanes up to 1280: csrc/prover_pipeline.hip, "entries per level-1 lane") — another cut of the bucket runs, the same sums.
    Three proofs in flight on an unsharded prover and on shard 1 of 2 (the rule goes by the shard's size), against the C
    restatement of src/groth16.cpp:171-204 and, assembled with shard 0's share, of the whole prove()."""
    import torch
    from rapidsnark_old_amd import synth
    k = 18                                                              # (a shard of two is 2^17: the smallest the rule takes)
    wl = _gpu_workload(zk, k)
    view = co.ZkeyView(wl)
    ws = [synth.make_witness(k, seed=s) for s in (1, 2, 3)]
    wd = [torch.from_numpy(w).to("cuda:0") for w in ws]
    want = [co.prove_msm(view, w) for w in ws]
    p = _prover(zk, wl, precomp=precomp)
    assert p.info()["window_bits_h"] == (16 if precomp else 12)           # (the plan of a 2^18 vector: 2^15 / 2^11 buckets)
    assert p.prove_msm_dev(wd[0].data_ptr()) == want[0]                 # a lone proof: the latency plan
    for d in wd:
        p.submit_dev(d.data_ptr())
    assert [p.collect_msm() for _ in wd] == want                        # the second and third: the busy plan
    import ctypes as C
    p.lib.zk_prover_destroy(p.h)
    p.h = C.c_void_p()
    shard = [_prover(zk, wl, precomp=precomp, shard_index=i, shard_count=2) for i in range(2)]
    lone = [[q.prove_msm_dev(d.data_ptr()) for d in wd] for q in shard]
    for d in wd:
        shard[1].submit_dev(d.data_ptr())
    busy = [shard[1].collect_msm() for _ in wd]
    assert busy == lone[1]
    r, s = 0x1234567, (1 << 200) + 99
    vk = {name: np.asarray(wl[name]).tobytes() for name in ("vk_alpha1", "vk_beta1", "vk_beta2", "vk_delta1", "vk_delta2")}
    for i, w in enumerate(ws):
        assert zk.assemble(vk, [lone[0][i], busy[i]], r, s) == co.prove(view, w, r, s)


def test_reserve_allocates_every_slot_of_the_ring_at_start_up(zk):
    """zk_prover_reserve (include/zkhip.h): the workspace of every slot and lane a pipeline of the reserved depth walks exists
    when the call returns — out of memory is a start-up error (Groth16::makeProver's fallback chain depends on it) and the first
    proofs do not pay for allocations.  Device memory in use must grow at reserve and NOT move while the pipeline fills; a
    shallower reservation afterwards gives the slots beyond its ring back."""
    from rapidsnark_old_amd import synth
    k = 16
    wl = _gpu_workload(zk, k)
    w = synth.make_witness(k, seed=5)
    p = _prover(zk, wl, precomp=True)
    before = p.info()["device_bytes_in_use"]
    depth = 4
    p.reserve(depth)
    reserved = p.info()["device_bytes_in_use"]
    assert reserved - before > (depth - 1) * 8 * (1 << 20), "reserve allocated nothing"
    r, s = 12345, 67890
    for _ in range(depth):
        p.submit_host(w, r, s)
    full = p.info()["device_bytes_in_use"]
    assert full == reserved, "slots were allocated at submit: %d bytes" % (full - reserved)
    proofs = [p.collect() for _ in range(depth)]
    assert len(set(proofs)) == 1 and proofs[0] == p.prove_host(w, r, s)
    p.reserve(2)
    assert p.info()["device_bytes_in_use"] < reserved, "slots beyond the shallower ring were kept"
    for _ in range(2):
        p.submit_host(w, r, s)
    assert [p.collect() for _ in range(2)] == proofs[:2]
