"""One proof on several GPUs with the chain PARTITIONED across them (north_star: "the five
MSMs and the NTT partitioned across the 8 GPUs"), exercised on the ONE GPU of the test box: every shard
prover lives on device 0, so the peer writes / cross-device events of zk_multi_prover become same-device
copies and waits, and the all_to_all of the one-process-per-GPU path (zk_shard_*) is played by tensor
copies between the shards' exchange buffers.  Results must be the golden / single-GPU proofs bit for bit."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden_bytes, golden_json, golden_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["r1cs_n64", "r1cs_n256"])
@pytest.mark.parametrize("shards", [2, 3, 4, 8])
@pytest.mark.parametrize("precomp", [False, True, 2])
def test_multi_prover_equals_golden(zk, name, shards, precomp):
    """zk_multi_prove: 2/4/8 shards partition the chain (cross stages of 1/2/3 index bits), 3 shards fall
    back to the replicated chain; same proof bytes either way."""
    meta = golden_json(name, "meta.json")
    mp = zk.MultiProver(golden_path(name, "circuit.zkey"), [0] * shards, precomp=precomp)
    assert mp.n_shards == shards and mp.chain_partitioned == (shards in (2, 4, 8))
    wt = golden_bytes(name, "witness.wtns")
    for _ in range(2):
        assert mp.prove(wt, r=int(meta["r"]), s=int(meta["s"])).hex() == meta["proof_bytes"]
    mp.close()


def test_multi_prover_pipeline_and_nopub(zk):
    """submit/collect with three proofs in flight on four shards; nPublic = 0 fixture."""
    name = "r1cs_n256"
    single = zk.Prover(golden_path(name, "circuit.zkey"))
    base = np.array(single._wtns_values(golden_bytes(name, "witness.wtns")), dtype=np.uint8)
    mp = zk.MultiProver(golden_path(name, "circuit.zkey"), [0, 0, 0, 0], precomp=True)
    hosts, want, rs = [], [], [(21 + i, 900 + 5 * i) for i in range(7)]
    for i, (r, s) in enumerate(rs):
        w = base.copy()
        w[32 * (3 + i)] ^= 0x11
        hosts.append(w)
        want.append(single.prove(w.tobytes(), r, s))
    got = []
    for i, (r, s) in enumerate(rs):
        if i >= 3:
            got.append(mp.collect())
        mp.submit(hosts[i], r, s)
    got += [mp.collect(), mp.collect(), mp.collect()]
    assert got == want
    with pytest.raises(zk.ZkHipError):
        mp.collect()
    mp.close()
    single.close()
    meta = golden_json("r1cs_nopub", "meta.json")
    mp = zk.MultiProver(golden_path("r1cs_nopub", "circuit.zkey"), [0, 0])
    assert mp.prove(golden_bytes("r1cs_nopub", "witness.wtns"), r=int(meta["r"]), s=int(meta["s"])).hex() == meta["proof_bytes"]
    mp.close()


@pytest.mark.parametrize("k,shards", [(12, 8), (16, 4), (20, 8)])
def test_partitioned_chain_at_scale(zk, tmp_path, k, shards):
    """Synthetic family: the partitioned chain (local pass plans of 2^(k - log2 G) blocks + cross stages)
    against the unsharded prover, on a .zkey file written to disk; precomputed tables on the shards."""
    import torch
    from oracle import c_oracle as co
    from rapidsnark_old_amd import synth
    from test_gpu_synth import _write_synth_files
    wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
    w = synth.make_witness(k, seed=4)
    zpath, wpath, _ = _write_synth_files(tmp_path, wl, w)
    r, s = 0x5555AAAA, (1 << 245) + 3
    single = zk.Prover(zpath)
    want = single.prove(w, r, s)
    single.close()
    mp = zk.MultiProver(zpath, [0] * shards, precomp=True)
    assert mp.chain_partitioned
    assert mp.prove(w, r, s) == want
    mp.submit(w, r, s)
    mp.submit(w, r, s)
    assert mp.collect() == want and mp.collect() == want
    mp.close()


@pytest.mark.parametrize("shards", [2, 8])
def test_shard_step_api_with_emulated_all_to_all(zk, shards):
    """zk_shard_begin / zk_shard_step as a one-process-per-GPU job drives them (rapidsnark_old_amd.dist.
    ShardedChain), with dist.all_to_all_single played by copies between the ranks' registered buffers:
    part r of every rank's send buffer lands in rank r's receive buffer."""
    import torch
    from rapidsnark_old_amd import lib as L
    from rapidsnark_old_amd.dist import ShardedChain
    name = "r1cs_n256"
    meta = golden_json(name, "meta.json")
    provers = [zk.Prover(golden_path(name, "circuit.zkey"), shard_index=i, shard_count=shards, partitioned_chain=True) for i in range(shards)]
    wt = np.array(provers[0]._wtns_values(golden_bytes(name, "witness.wtns")), dtype=np.uint8)
    chains = []

    def make_exchange(rank):
        def exchange(dst, src):            # called per rank in lock-step below: only stage this rank's request
            pending.append((rank, dst, src))
        return exchange

    pending = []
    for i, p in enumerate(provers):
        chains.append(ShardedChain(p._lib, p._h, None, torch.device("cuda:0"), exchange=make_exchange(i)))

    def run_all_to_all():          # stream-ordered copies on torch's current (default) stream: no host synchronisation
        reqs = sorted(pending, key=lambda t: t[0])
        assert [t[0] for t in reqs] == list(range(shards))
        srcs = [t[2].clone() for t in reqs]
        for rnk, dst, _ in reqs:
            part = dst.numel() // shards
            for sidx in range(shards):      # all_to_all_single: part `rnk` of rank sidx's buffer -> part sidx of mine
                dst[sidx * part:(sidx + 1) * part] = srcs[sidx][rnk * part:(rnk + 1) * part]
        pending.clear()

    # drive the ranks in lock-step: each phase of every rank, then the exchange
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for ch, p in zip(chains, provers):
        L.check(p._lib.zk_shard_begin(p._h, C.c_void_p(wt.ctypes.data), None, None, None, stream))
        ch.exchange(ch.recv, ch.send)
    run_all_to_all()
    for step, (dst, src) in ((L.ZK_STEP_CROSS_INVERSE, ("send", "recv")), (L.ZK_STEP_LOCAL, ("recv", "send")), (L.ZK_STEP_CROSS_FORWARD, ("send", "recv"))):
        for ch, p in zip(chains, provers):
            L.check(p._lib.zk_shard_step(p._h, step, stream))
            ch.exchange(getattr(ch, dst), getattr(ch, src))
        run_all_to_all()
    parts = []
    for ch, p in zip(chains, provers):
        L.check(p._lib.zk_shard_step(p._h, L.ZK_STEP_FINISH, stream))
        parts.append(p.collect_msm())
    proof = provers[0].prove_finish(parts, r=int(meta["r"]), s=int(meta["s"]))
    assert proof.hex() == meta["proof_bytes"]
    # a partitioned prover refuses the one-call entry points
    with pytest.raises(zk.ZkHipError):
        provers[0].prove_msm(wt.tobytes())
    for p in provers:
        p.close()


@pytest.mark.parametrize("ranks,k", [(2, 16), (8, 14)])
def test_bench_multi_rank_path_on_one_gpu(ranks, k):
    """bench.py --gpus N exactly as the driver launches it (torch.distributed.run, one process per rank), with every rank on
    cuda:0 and the collectives over gloo (ZK_BENCH_SHARE_GPU=1: the only difference to a real node is the backend of
    all_to_all_single / all_gather): ShardedChain.submit_host_sliced, the four all_to_all rounds per proof, the 384-byte
    all_gather, rank 0's assembly.  The run checks one sharded proof against an unsharded prover itself."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    import gc
    import socket
    gc.collect()                       # provers of earlier tests still hold streams (hardware queues) of the one GPU the ranks share
    # the ranks share ONE GPU with this pytest process: few hardware queues each, or the driver time-slices them (the command
    # alone takes 4-8 s; 15 of 15 stand-alone runs passed, one run inside a long pytest session once sat in its time-out)
    env = dict(os.environ, ZK_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", GPU_MAX_HW_QUEUES="2")
    res = None
    for attempt in range(2):
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "4", "--warmup", "1",
               "--log2n", str(k), "--no-cpu"]
        try:
            res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
            break
        except subprocess.TimeoutExpired:
            if attempt:
                raise
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-2000:] + res.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["rccl_ranks"] == ranks and d["steps"] == 4
    # the first-contact kit ran before anything was timed: rendezvous, one all_to_all of the proof's exchange size checked
    # byte for byte against a host reference, the 384-byte all_gather
    fc = d["first_contact"]
    assert fc["world"] == ranks and fc["all_to_all_bytes_ok"] is True and fc["all_gather_ok"] is True and fc["chain_partitioned"] is True
    assert fc["exchange_bytes_per_rank"] == 3 * ((1 << k) // ranks) * 32 and fc["all_to_all_ms"] > 0 and fc["backend"] == "gloo"
    assert d["multi_gpu_proof_equals_single_gpu_proof"] is True
    assert "chain partitioned" in d["config"]["parallelism"] and d["config"]["witness_upload"].startswith("each rank uploads 1/N")
    assert set(d["exchange_ms"]) == {"witness_all_gather", "all_to_all_1_to_cross_inverse", "all_to_all_2_to_local",
                                     "all_to_all_3_to_cross_forward", "all_to_all_4_to_finish"}
    assert d["value"] > 0 and d["scaling"] == "strong"
    # the other use of N GPUs, timed in the same run: independent replicas, weak scaling
    assert d["replicas"]["scaling"] == "weak" and d["replicas"]["value"] > 0 and d["replicas"]["steps_per_gpu"] == 4
