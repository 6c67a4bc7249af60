"""GPU parity at BASELINE.json's full sizes: 2^22 on one GPU (configs[2]) and 2^24 split into eight
point-range shards (configs[3], the eight shards run one after the other on the one GPU of the test
box).  Nothing here can be compared against a CPU proof in seconds, so the checks are the
size-independent ones the synthetic family was built for: every point table has KNOWN
discrete logs, so each MSM result and the assembled proof are checked in Fr alone (three scalar
multiplications on the host), h is cross-checked against the C restatement's FFT pipeline, and the
NTT pass plans used at these sizes (10+6+6 and 10+7+7 bits) get round-trip / delta / linearity checks."""
import ctypes as C

import numpy as np
import pytest

from oracle import bn254 as bn, c_oracle as co

pytestmark = pytest.mark.gpu

G1B = bn.g1_to_bytes(bn.G1.gen)
G2B = bn.g2_to_bytes(bn.G2.gen)
_WL = {}


def _workload(zk, k):
    from rapidsnark_old_amd import synth
    if k not in _WL:
        _WL.clear()                      # one big workload at a time in host RAM
        _WL[k] = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
    return _WL[k]


def _prover(zk, wl, **kw):
    from rapidsnark_old_amd import views
    return views.ProverFromView(zk, wl, device=0, shard_index=kw.get("shard_index", 0), shard_count=kw.get("shard_count", 1),
                                window_bits=0, timings=False, precomp=kw.get("precomp", False))


def _destroy(p):
    p.lib.zk_prover_destroy(p.h)
    p.h = C.c_void_p()


def _check_sums(sums, dl):
    assert sums[0:64] == co.g1_mul(G1B, dl["pih"])            # also proves the GPU's h == the oracle's h
    assert sums[64:128] == co.g1_mul(G1B, dl["pi_a"])
    assert sums[128:192] == co.g1_mul(G1B, dl["pib1"])
    assert sums[192:320] == co.g2_mul(G2B, dl["pi_b"])
    assert sums[320:384] == co.g1_mul(G1B, dl["pi_c"])


def test_full_size_2p22_dlog_identities(zk):
    """BASELINE configs[2]: all five MSMs and the assembled proof at 2^22, tables as in the zkey and
    window-precomputed (c = 20, one shared set of 2^19 buckets), synchronous and with two host-witness
    proofs in flight; h against the C restatement of src/groth16.cpp:98-163."""
    import torch
    from rapidsnark_old_amd import synth
    k = 22
    wl = _workload(zk, k)
    w = synth.make_witness(k)
    h = co.compute_h(co.ZkeyView(wl), w)
    dl = synth.expected_msm_dlogs(wl, w, np.frombuffer(h, dtype=np.uint8))
    r, s = 3141592653589793, 2718281828459045
    a, b, c = synth.expected_proof_dlogs(wl, dl, r, s)
    want = co.g1_mul(G1B, a) + co.g2_mul(G2B, b) + co.g1_mul(G1B, c)
    wd = torch.from_numpy(w).to("cuda:0")
    for precomp in (False, True, 2):         # 2: rows for every second window (7 x the tables, two sets of 2^19 buckets)
        p = _prover(zk, wl, precomp=precomp)
        _check_sums(p.prove_msm_dev(wd.data_ptr()), dl)
        assert p.prove_dev(wd.data_ptr(), r, s) == want
        # the reference's contract: witness in host memory, two proofs in flight
        p.submit_host(w, r, s)
        p.submit_host(w, r, s)
        assert p.collect() == want and p.collect() == want
        _destroy(p)


def test_2p24_sharded8_on_one_gpu(zk):
    """BASELINE configs[3]: a 2^24 proof split into eight point-range shards.  The eight
    shard provers run one after the other on this box's one GPU; their partial sums go through
    zk_prove_finish exactly as rank 0 of an 8-GPU job would do it, and the result must be the proof
    with the known discrete logs."""
    import torch
    from rapidsnark_old_amd import synth
    k, shards = 24, 8
    wl = _workload(zk, k)
    w = synth.make_witness(k)
    h = co.compute_h(co.ZkeyView(wl), w)
    dl = synth.expected_msm_dlogs(wl, w, np.frombuffer(h, dtype=np.uint8))
    r, s = 0x123456789ABCDEF, (1 << 240) + 7
    a, b, c = synth.expected_proof_dlogs(wl, dl, r, s)
    want = co.g1_mul(G1B, a) + co.g2_mul(G2B, b) + co.g1_mul(G1B, c)
    wd = torch.from_numpy(w).to("cuda:0")
    parts = []
    for i in range(shards):
        p = _prover(zk, wl, shard_index=i, shard_count=shards, precomp=(i % 2 == 1))    # both table modes
        parts.append(p.prove_msm_dev(wd.data_ptr()))
        _destroy(p)
    assert len(set(parts)) == shards
    vk = {name: np.asarray(wl[name]).tobytes() for name in ("vk_alpha1", "vk_beta1", "vk_beta2", "vk_delta1", "vk_delta2")}
    assert zk.assemble(vk, parts, r, s) == want
    # configs[3] AS DESIGNED (north_star: "the five MSMs and the NTT partitioned across the 8 GPUs"): the same proof through
    # zk_multi_prover with eight shards and the chain PARTITIONED — blocks of 2^21 (local pass plan of a 2^21 block + three
    # cross stages, k_ntt_cross<.,3>), A.w/B.w rows split by block, peer writes and cross-device events (all on device 0
    # here), partial sums added on the host — window-precomputed tables, then tables as in the zkey
    from rapidsnark_old_amd import views
    for precomp in (True, False):
        mp = views.MultiProverFromView(zk, wl, [0] * shards, precomp=precomp)
        assert mp.n_shards == shards and mp.chain_partitioned
        assert mp.prove(w, r, s) == want
        mp.close()
    _WL.clear()


@pytest.mark.parametrize("logn", [22, 24])
def test_ntt_properties_at_full_size(zk, logn):
    """zk_fr_ntt at the domain sizes of configs[2]/[3] (pass plans 10+6+6 and 10+7+7): ifft(fft(x)) = x,
    the transform of a delta at position j is the geometric sequence w^(j k), and linearity on a
    sample of outputs."""
    from rapidsnark_old_amd import synth
    n = 1 << logn
    rng = np.random.default_rng(logn)
    x = synth.random_fr_bytes(rng, n).reshape(-1).copy()
    X = zk.fr_ntt(x, inverse=False)
    assert zk.fr_ntt(X, inverse=True) == x.tobytes()
    # delta at j (Montgomery one) -> w_n^(j k) in Montgomery form
    j = 12345 % n
    d = np.zeros(n * 32, dtype=np.uint8)
    one_m = bn.int_to_le32(bn.to_mont(1, bn.R_MOD))
    d[32 * j:32 * j + 32] = np.frombuffer(one_m, dtype=np.uint8)
    D = zk.fr_ntt(d, inverse=False)
    wn = pow(bn.ROOT_2_28, 1 << (28 - logn), bn.R_MOD)
    for kk in (0, 1, 2, 1023, 2048, n // 2, n // 2 + 77, n - 1, int(rng.integers(0, n))):
        got = int.from_bytes(D[32 * kk:32 * kk + 32], "little")
        assert got == bn.to_mont(pow(wn, j * kk, bn.R_MOD), bn.R_MOD), kk
    # linearity: x + delta_j transforms to X + D (sampled)
    xj = (int.from_bytes(x[32 * j:32 * j + 32].tobytes(), "little") + bn.to_mont(1, bn.R_MOD)) % bn.R_MOD
    x[32 * j:32 * j + 32] = np.frombuffer(bn.int_to_le32(xj), dtype=np.uint8)
    X2 = zk.fr_ntt(x, inverse=False)
    for kk in (0, 5, 4097, n // 2 + 1, n - 2, int(rng.integers(0, n))):
        a, b, c = (int.from_bytes(t[32 * kk:32 * kk + 32], "little") for t in (X, D, X2))
        assert (a + b) % bn.R_MOD == c, kk
