"""The one exchange step of the multi-GPU path: every rank contributes its
384-byte record of partial MSM sums, rank 0 receives all of them.

One process per GPU; `dist` is torch.distributed with backend "nccl" (= RCCL over xGMI) on
the GPU box and "gloo" in the CPU tests.  Elliptic-curve addition is not a collective
reduction op, so this is an all_gather of a tiny payload followed by a local O(N) add in
zk_prove_finish / zk_assemble — never an all-reduce, and bandwidth is irrelevant."""
import numpy as np
import torch

PARTIAL_BYTES = 384   # sizeof(zk_msm_sums)


def gather_partials(partial: bytes, dist, device):
    """-> list of every rank's partial-sum record (on all ranks)."""
    world = dist.get_world_size()
    src = torch.frombuffer(bytearray(partial), dtype=torch.uint8).to(device)
    assert src.numel() == PARTIAL_BYTES
    out = [torch.empty(PARTIAL_BYTES, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(out, src)
    return [t.cpu().numpy().tobytes() for t in out]


class ShardedChain:
    """One rank of a proof whose chain (A.w/B.w rows + the six transforms) is PARTITIONED across the
    ranks (include/zkhip.h, zk_shard_*): the library computes, this class moves the blocks — four
    all_to_all_single per proof on two torch tensors registered with the prover.
    `dist` is torch.distributed (backend nccl = RCCL over xGMI); `exchange(dst, src)` can be replaced for
    tests (e.g. staging through the CPU for gloo)."""

    def __init__(self, lib, handle, dist, device, exchange=None):
        import ctypes as C
        from . import lib as L
        self.lib, self.h, self.dist, self.L, self.C = lib, handle, dist, L, C
        nloc, part = C.c_uint64(), C.c_uint32()
        L.check(lib.zk_shard_info(handle, C.byref(nloc), C.byref(part)))
        if not part.value:
            raise ValueError("prover was not created with ZK_FLAG_PARTITIONED_CHAIN")
        self.nloc = nloc.value
        # both buffers: [GPU][polynomial][chunk] as bytes — all_to_all_single splits them into world_size equal parts
        self.send = torch.zeros(3 * self.nloc * 32, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(3 * self.nloc * 32, dtype=torch.uint8, device=device)
        L.check(lib.zk_shard_set_exchange(handle, C.c_void_p(self.send.data_ptr()), C.c_void_p(self.recv.data_ptr())))
        self.exchange = exchange or self._all_to_all
        self._phase_events = []          # per proof: five (start, end) CUDA event pairs around the exchanges (self-diagnosing SCALE runs)
        self.time_phases = device.type == "cuda"

    PHASES = ("witness_all_gather", "all_to_all_1_to_cross_inverse", "all_to_all_2_to_local", "all_to_all_3_to_cross_forward", "all_to_all_4_to_finish")

    def _timed(self, slot, fn, *a):
        if not self.time_phases:
            return fn(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a)
        e1.record()
        self._cur[slot] = (e0, e1)
        return r

    def phase_times_ms(self):
        """Mean GPU time of every exchange phase over the proofs submitted so far (includes waiting for the slowest peer:
        the collective cannot start before every rank has enqueued it)."""
        torch.cuda.synchronize()
        acc, cnt = [0.0] * len(self.PHASES), [0] * len(self.PHASES)
        for ev in self._phase_events:
            for i, pair in enumerate(ev):
                if pair is not None:
                    acc[i] += pair[0].elapsed_time(pair[1])
                    cnt[i] += 1
        self._phase_events = []
        return {name: round(acc[i] / cnt[i], 4) for i, name in enumerate(self.PHASES) if cnt[i]}

    def _all_to_all(self, dst, src):
        self.dist.all_to_all_single(dst, src)

    def enable_sliced_upload(self, n_vars, rank, world, depth, device, gather=None):
        """Host witnesses: every rank uploads only ITS 1/world of the witness over PCIe and the ranks
        all_gather the rest over xGMI (every GPU needs the whole vector: the rows of A.w / B.w gather random
        columns).  Without this each rank stages and uploads all of it — world x the host-memory traffic, and
        at 8 ranks the 128 MiB upload of a 2^22 witness (staging 5 ms + DMA 2.6 ms) is as long as the proof
        period.  `depth` buffers rotate (one per proof in flight).  `gather(full, part)` replaces
        dist.all_gather_into_tensor in tests."""
        nbytes = n_vars * 32
        if nbytes % world:
            return False
        self._sl = (rank * (nbytes // world), (rank + 1) * (nbytes // world))
        self._wfull = [torch.empty(nbytes, dtype=torch.uint8, device=device) for _ in range(depth)]
        self._wpart = [torch.empty(nbytes // world, dtype=torch.uint8, device=device) for _ in range(depth)]
        self._wpin = [torch.empty(nbytes // world, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self._wk = 0
        self._gather = gather or (lambda full, part: self.dist.all_gather_into_tensor(full, part))
        return True

    def submit_host_sliced(self, wtns, r=None, s=None):
        """wtns: the whole witness as a numpy uint8 array in host memory (identical on every rank)."""
        k = self._wk % len(self._wfull)
        self._wk += 1
        lo, hi = self._sl
        self._wpin[k].copy_(torch.from_numpy(wtns[lo:hi]))            # pageable -> pinned, 1/world of the witness
        self._wpart[k].copy_(self._wpin[k], non_blocking=True)        # PCIe, on the current stream
        self._cur = [None] * len(self.PHASES)
        self._timed(0, self._gather, self._wfull[k], self._wpart[k])  # xGMI
        self.submit(d_wtns=self._wfull[k].data_ptr(), r=r, s=s, _keep_cur=True)

    def submit(self, wtns=None, d_wtns=None, r=None, s=None, _keep_cur=False):
        """Enqueue one proof: wtns = host numpy uint8 array (kept alive by the caller until collected) or
        d_wtns = device pointer."""
        C, L = self.C, self.L
        if not _keep_cur:
            self._cur = [None] * len(self.PHASES)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ra = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8) if r is not None else None
        sa = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8) if s is not None else None
        L.check(self.lib.zk_shard_begin(self.h, C.c_void_p(wtns.ctypes.data) if wtns is not None else None,
                                        C.c_void_p(d_wtns) if d_wtns is not None else None,
                                        C.c_void_p(ra.ctypes.data) if ra is not None else None,
                                        C.c_void_p(sa.ctypes.data) if sa is not None else None, stream))
        self._timed(1, self.exchange, self.recv, self.send)
        L.check(self.lib.zk_shard_step(self.h, L.ZK_STEP_CROSS_INVERSE, stream))
        self._timed(2, self.exchange, self.send, self.recv)
        L.check(self.lib.zk_shard_step(self.h, L.ZK_STEP_LOCAL, stream))
        self._timed(3, self.exchange, self.recv, self.send)
        L.check(self.lib.zk_shard_step(self.h, L.ZK_STEP_CROSS_FORWARD, stream))
        self._timed(4, self.exchange, self.send, self.recv)
        L.check(self.lib.zk_shard_step(self.h, L.ZK_STEP_FINISH, stream))
        if self.time_phases:
            self._phase_events.append(self._cur)
            if len(self._phase_events) > 256:
                self._phase_events.pop(0)


def first_contact(dist, device, rank, world, exchange_bytes, exchange=None, gather=None):
    """What the FIRST run on a real multi-GPU node should say about itself before anything is timed (nothing in this
    repository has ever met a second GPU: DESIGN.md section 7).  On every rank; -> dict (the same on every rank):
      peer_access           matrix [i][j] = torch.cuda.can_device_access_peer(i, j) over the devices this process sees
                            (what zk_multi_prover's peer writes need; the one-process-per-GPU path needs only RCCL)
      all_to_all_bytes_ok   ONE all_to_all_single on a buffer of the proof's real exchange size (exchange_bytes per rank, uint8
                            views as ShardedChain uses them), every byte a function of (source rank, destination rank, offset),
                            checked on the receiver against a HOST-computed reference, byte for byte
      all_gather_ok         the 384-byte partial-sum record path (gather_partials) with a rank-stamped payload
      all_to_all_ms / all_to_all_GBps_per_rank   second, timed run of the same exchange
    A collective that fails or a mismatch raises: the caller turns it into a non-zero exit with the reason in its JSON line."""
    import time
    out = {"world": world, "backend": dist.get_backend(), "exchange_bytes_per_rank": int(exchange_bytes)}
    if device.type == "cuda":
        nd = torch.cuda.device_count()
        out["visible_devices"] = nd
        out["peer_access"] = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(nd)] for i in range(nd)]
        out["device_name"] = torch.cuda.get_device_name(device)
    part = exchange_bytes // world
    if part == 0 or exchange_bytes % world:
        raise ValueError("exchange buffer of %d bytes does not split into %d parts" % (exchange_bytes, world))
    idx = np.arange(part, dtype=np.uint64)

    def pattern(src, dst):              # byte k of the chunk rank `src` sends to rank `dst` (host reference)
        return ((idx * 2654435761 + src * 131 + dst * 17 + (idx >> 11)) % 251).astype(np.uint8)

    send_host = np.concatenate([pattern(rank, d) for d in range(world)])
    want_host = np.concatenate([pattern(s_, rank) for s_ in range(world)])
    send = torch.from_numpy(send_host).to(device)
    recv = torch.zeros(exchange_bytes, dtype=torch.uint8, device=device)
    xch = exchange or (lambda dst, src: dist.all_to_all_single(dst, src))
    xch(recv, send)
    if device.type == "cuda":
        torch.cuda.synchronize()
    got = recv.cpu().numpy()
    bad = int((got != want_host).sum())
    # every rank learns whether ANY rank saw wrong bytes (so that all of them stop together instead of one leaving the
    # others inside the next collective)
    flag = torch.tensor([0 if bad else 1], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    out["all_to_all_bytes_ok"] = int(flag.item()) == 1
    if bad:
        first = int(np.nonzero(got != want_host)[0][0])
        raise RuntimeError("all_to_all_single delivered %d wrong bytes on rank %d (first at offset %d: chunk of source rank %d)" % (bad, rank, first, first // part))
    if not out["all_to_all_bytes_ok"]:
        raise RuntimeError("all_to_all_single delivered wrong bytes on another rank (this rank's %d bytes are right)" % exchange_bytes)
    t0 = time.perf_counter()
    xch(recv, send)
    if device.type == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["all_to_all_ms"] = round(dt * 1e3, 3)
    out["all_to_all_GBps_per_rank"] = round(exchange_bytes * (world - 1) / world / dt / 1e9, 2)
    # the 384-byte record path
    rec = bytes([(rank * 37 + i) % 256 for i in range(PARTIAL_BYTES)])
    if gather is not None:
        parts = gather(rec)
    else:
        parts = gather_partials(rec, dist, device)
    ok = all(parts[r_] == bytes([(r_ * 37 + i) % 256 for i in range(PARTIAL_BYTES)]) for r_ in range(world))
    out["all_gather_ok"] = ok
    if not ok:
        raise RuntimeError("all_gather of the partial-sum records returned wrong bytes on rank %d" % rank)
    return out
