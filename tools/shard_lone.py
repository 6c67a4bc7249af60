"""One rank's share of a sharded proof, alone on the GPU with pauses around it, for a rocprofv3 --kernel-trace timeline
(tools/lone_timeline.py cuts clusters at 30 ms of silence):
    python tools/shard_lone.py [log2n=22] [G=8] [proofs=4] [in_flight=1]
Rank 0's share with the chain partitioned and the all_to_all left out, exactly as tools/shard_probe.py runs it.
in_flight=N >= 2: no pauses, `proofs` shares with N in flight (the kernel table of a rank's steady state: rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth
from rapidsnark_old_amd.dist import ShardedChain

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
inflight = int(sys.argv[4]) if len(sys.argv) > 4 else 1
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
w = torch.from_numpy(synth.make_witness(k, seed=0)).cuda()
p = bench.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=G, window_bits=0, timings=False, precomp=True, partitioned_chain=True)
ch = ShardedChain(p.lib, p.h, None, torch.device("cuda:0"), exchange=lambda dst, src: None)
for i in range(3):
    ch.submit(d_wtns=w.data_ptr()); p.collect_msm()
torch.cuda.synchronize()
if inflight >= 2:
    for _ in range(inflight - 1):
        ch.submit(d_wtns=w.data_ptr())
    t0 = time.perf_counter()
    for i in range(reps):
        ch.submit(d_wtns=w.data_ptr()); p.collect_msm()
    for _ in range(inflight - 1):
        p.collect_msm()
    print("share of 2^%d / %d, %d in flight: %.2f ms each (%d + %d shares)" % (k, G, inflight, (time.perf_counter() - t0) / (reps + inflight - 1) * 1e3, reps, inflight - 1), flush=True)
    p.lib.zk_prover_destroy(p.h)
    torch.cuda.synchronize()
    sys.exit(0)
for i in range(reps):
    time.sleep(0.06)
    t0 = time.perf_counter()
    ch.submit(d_wtns=w.data_ptr()); p.collect_msm()
    print("share %d of 2^%d / %d: %.2f ms wall" % (i, k, G, (time.perf_counter() - t0) * 1e3), flush=True)
time.sleep(0.06)
