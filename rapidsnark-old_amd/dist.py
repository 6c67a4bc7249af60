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
    ranks (include/zkhip.h, zk_shard_*): the library computes, this class moves the blocks — four rounds
    of all_to_all per proof (three polynomials each) on two torch tensors registered with the prover.
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
        # [poly][block] as bytes; a block is split into world_size equal chunks by all_to_all_single
        self.abc = torch.zeros((3, self.nloc * 32), dtype=torch.uint8, device=device)
        self.xb = torch.zeros((3, self.nloc * 32), dtype=torch.uint8, device=device)
        L.check(lib.zk_shard_set_exchange(handle, C.c_void_p(self.abc.data_ptr()), C.c_void_p(self.xb.data_ptr())))
        self.exchange = exchange or self._all_to_all

    def _all_to_all(self, dst, src):
        for poly in range(3):
            self.dist.all_to_all_single(dst[poly], src[poly])

    def submit(self, wtns=None, d_wtns=None, r=None, s=None):
        """Enqueue one proof: wtns = host numpy uint8 array (kept alive by the caller until collected) or
        d_wtns = device pointer."""
        C, L = self.C, self.L
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ra = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8) if r is not None else None
        sa = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8) if s is not None else None
        L.check(self.lib.zk_shard_begin(self.h, C.c_void_p(wtns.ctypes.data) if wtns is not None else None,
                                        C.c_void_p(d_wtns) if d_wtns is not None else None,
                                        C.c_void_p(ra.ctypes.data) if ra is not None else None,
                                        C.c_void_p(sa.ctypes.data) if sa is not None else None, stream))
        self.exchange(self.xb, self.abc)
        L.check(self.lib.zk_shard_step(self.h, L.ZK_STEP_CROSS_INVERSE, stream))
        self.exchange(self.abc, self.xb)
        L.check(self.lib.zk_shard_step(self.h, L.ZK_STEP_LOCAL, stream))
        self.exchange(self.xb, self.abc)
        L.check(self.lib.zk_shard_step(self.h, L.ZK_STEP_CROSS_FORWARD, stream))
        self.exchange(self.abc, self.xb)
        L.check(self.lib.zk_shard_step(self.h, L.ZK_STEP_FINISH, stream))
