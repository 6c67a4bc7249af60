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
