"""Per-rank cost of a sharded proof on ONE GPU: rank 0's share of a world of G (no exchange).
    python tools/shard_probe.py [log2n=22] [worlds=1,2,4,8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
w = torch.from_numpy(synth.make_witness(k, seed=0)).cuda()
worlds = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 4, 8]
for G in worlds:
    p = bench.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=G, window_bits=0, timings=True, precomp=True)
    for i in range(2):
        p.prove_msm_dev(w.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5):
        p.prove_msm_dev(w.data_ptr())
    dt = (time.perf_counter() - t0) / 5 * 1e3
    print("world %d: rank-0 share %.2f ms  %s" % (G, dt, {a: round(b, 2) for a, b in p.timings().items()}), flush=True)
    if True:
        p.submit_dev(w.data_ptr())
        t0 = time.perf_counter()
        for i in range(8):
            p.submit_dev(w.data_ptr())
            p.collect_msm()
        p.collect_msm()
        print("   two in flight: %.2f ms" % ((time.perf_counter() - t0) / 9 * 1e3), flush=True)
    L = p.L
    L.check(0)
    p.lib.zk_prover_destroy(p.h)
