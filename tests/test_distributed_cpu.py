"""CPU, world_size 2, gloo: the N>1 path of bench.py / SURVEY §8e — every rank proves its
point-range shard, one all_gather of the 384-byte partial-sum records, rank 0 assembles.
The per-shard MSM stage is stood in by the C oracle on sliced tables (the GPU stage is covered
by tests/test_gpu_prove.py::test_sharded_equals_whole); the exchange + host assembly are the
product's own code (rapidsnark_old_amd.dist.gather_partials, zk_assemble)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shard_view(zk_bytes, wt_vals, idx, cnt):
    """Slice every table like csrc/prover_create.hip::prover_create does (contiguous index ranges)."""
    from oracle import c_oracle as co
    full = co.ZkeyView(zk_bytes)
    v = full.v
    import ctypes as C

    def arr(ptr, nbytes):
        return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)).copy() if nbytes else np.zeros(0, np.uint8)

    nV, nPub, n = v.nVars, v.nPublic, v.domainSize
    per_v, per_h = -(-nV // cnt), -(-n // cnt)
    vlo, vhi = min(per_v * idx, nV), min(per_v * (idx + 1), nV)
    hlo, hhi = min(per_h * idx, n), min(per_h * (idx + 1), n)
    A = arr(v.pointsA, nV * 64)[vlo * 64:vhi * 64]
    B1 = arr(v.pointsB1, nV * 64)[vlo * 64:vhi * 64]
    B2 = arr(v.pointsB2, nV * 128)[vlo * 128:vhi * 128]
    H = arr(v.pointsH, n * 64)[hlo * 64:hhi * 64]
    Call = arr(v.pointsC, (nV - nPub - 1) * 64)
    w = np.frombuffer(wt_vals, dtype=np.uint8).reshape(nV, 32)
    first = nPub + 1
    clo, chi = max(vlo, first), max(vhi, max(vlo, first))
    Cs = Call[(clo - first) * 64:(chi - first) * 64]
    return (A, B1, B2, Cs, H, w[vlo:vhi].reshape(-1), w[clo:chi].reshape(-1), (hlo, hhi), full)


def _worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        import rapidsnark_old_amd as zk
        from oracle import c_oracle as co, groth16_ref as g, bn254 as bn
        gold = os.path.join(ROOT, "tests", "golden", name)
        zkb = open(os.path.join(gold, "circuit.zkey"), "rb").read()
        meta = json.load(open(os.path.join(gold, "meta.json")))
        wt = g.read_wtns(open(os.path.join(gold, "witness.wtns"), "rb").read())
        vals = b"".join(bn.int_to_le32(v) for v in wt["witness"])
        A, B1, B2, Cs, H, w_s, w_c, (hlo, hhi), full = _shard_view(zkb, vals, rank, world)
        h = np.frombuffer(co.compute_h(full, vals), dtype=np.uint8)          # replicated on every rank
        part = (co.msm_g1(H, h[hlo * 32:hhi * 32].copy()) + co.msm_g1(A, w_s.copy()) + co.msm_g1(B1, w_s.copy())
                + co.msm_g2(B2, w_s.copy()) + co.msm_g1(Cs, w_c.copy()))
        parts = zk.gather_partials(part, dist, torch.device("cpu"))            # the exchange under test
        if rank == 0:
            f = zk.open_existing(zkb, "zkey", 1)
            hd = zk.load_zkey_header(f)
            vk = {k: getattr(hd, k) for k in ("vk_alpha1", "vk_beta1", "vk_beta2", "vk_delta1", "vk_delta2")}
            proof = zk.assemble(vk, parts, int(meta["r"]), int(meta["s"]))
            q.put((proof.hex() == meta["proof_bytes"], len(parts)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name", ["r1cs_n64", "r1cs_nopub"])
def test_two_rank_sharded_proof_over_gloo(name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    ok, nparts = q.get(timeout=5)
    assert ok and nparts == 2


def _fc_worker(rank, world, port, corrupt, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rapidsnark_old_amd  # noqa: F401
        from rapidsnark_old_amd.dist import first_contact
        exchange = None
        if corrupt:
            def exchange(dst, src):
                dist.all_to_all_single(dst, src)
                if rank == 1:
                    dst[dst.numel() // 2 + 5] ^= 0x40          # one flipped bit in the chunk that came from rank 1
        try:
            info = first_contact(dist, torch.device("cpu"), rank, world, 3 * 1024 * 32, exchange=exchange)
            q.put((rank, "ok", info))
        except RuntimeError as exc:
            q.put((rank, "error", str(exc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True])
def test_first_contact_kit_checks_the_exchange_byte_for_byte(corrupt):
    """rapidsnark_old_amd.dist.first_contact (what bench.py --gpus N runs before it times anything): world_size 2 over gloo —
    the all_to_all of the chain's exchange buffer against the host-computed pattern, the 384-byte all_gather; a single
    flipped bit on one rank is reported by that rank with the offset and the source rank."""
    world = 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fc_worker, args=(r, world, port, corrupt, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (st, info)) for r, st, info in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(60)
    if not corrupt:
        for r in range(world):
            st, info = res[r]
            assert st == "ok" and info["all_to_all_bytes_ok"] and info["all_gather_ok"] and info["world"] == 2 and info["backend"] == "gloo"
            assert info["exchange_bytes_per_rank"] == 3 * 1024 * 32 and info["all_to_all_ms"] > 0
    else:
        assert res[0][0] == "error" and "another rank" in res[0][1]          # every rank stops, the faulty one says where
        st, msg = res[1]
        assert st == "error" and "1 wrong bytes on rank 1" in msg and "source rank 1" in msg
