"""The one exchange step of the multi-GPU path: every rank contributes its
384-byte record of partial MSM sums, rank 0 receives all of them.

One process per GPU; `dist` is torch.distributed with backend "nccl" (= RCCL over xGMI) on
the GPU box and "gloo" in the CPU tests.  Elliptic-curve addition is not a collective
reduction op, so this is an all_gather of a tiny payload followed by a local O(N) add in
zk_prove_finish / zk_assemble — never an all-reduce, and bandwidth is irrelevant."""
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
