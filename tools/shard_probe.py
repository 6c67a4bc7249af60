"""Per-rank cost of a sharded proof on ONE GPU: rank 0's share of a world of G (no exchange).
    python tools/shard_probe.py [log2n=22] [worlds=1,2,4,8] [chain=both|replicated|partitioned] [window_bits=0 (the plan's)]
With the chain partitioned (ZK_FLAG_PARTITIONED_CHAIN) rank 0 runs its block's phases through zk_shard_*
with the all_to_all left out: its exchange buffers keep whatever they hold, so the RESULT is meaningless
but the work (kernels, sizes, launch counts) is exactly a rank's — what is missing is the four rounds of
all_to_all (2 x 7/8 of a block per transform over xGMI at G = 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth
from rapidsnark_old_amd.dist import ShardedChain

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
w = torch.from_numpy(synth.make_witness(k, seed=0)).cuda()
worlds = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 4, 8]
chain = sys.argv[3] if len(sys.argv) > 3 else "both"
wbits = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for G in worlds:
    for mode in ("replicated", "partitioned"):
        if chain not in ("both", mode) or (mode == "partitioned" and G not in (2, 4, 8)):
            continue
        part = mode == "partitioned"
        p = bench.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=G, window_bits=wbits, timings=True, precomp=True, partitioned_chain=part)
        if part:
            ch = ShardedChain(p.lib, p.h, None, torch.device("cuda:0"), exchange=lambda dst, src: None)
            submit = lambda: ch.submit(d_wtns=w.data_ptr())
        else:
            submit = lambda: p.submit_dev(w.data_ptr())
        for i in range(2):
            submit(); p.collect_msm()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(6):
            submit(); p.collect_msm()
        dt = (time.perf_counter() - t0) / 6 * 1e3
        tm = {a: round(b, 2) for a, b in p.timings().items()}
        submit()
        t0 = time.perf_counter()
        for i in range(10):
            submit(); p.collect_msm()
        p.collect_msm()
        dt2 = (time.perf_counter() - t0) / 11 * 1e3
        print("world %d %-11s chain: rank-0 share %.2f ms one at a time, %.2f ms two in flight   %s" % (G, mode, dt, dt2, tm), flush=True)
        p.lib.zk_prover_destroy(p.h)
