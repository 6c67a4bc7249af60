"""Host-side cost of one proof at small sizes: time inside submit vs collect (two in flight).
    python tools/cpu_overhead_probe.py [log2n=16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
p = bench.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True)
w = torch.from_numpy(synth.make_witness(k, seed=0)).cuda()
for _ in range(3):
    p.prove_dev(w.data_ptr())
ts = tc = 0.0
N = 40
p.submit_dev(w.data_ptr())
t0 = time.perf_counter()
for i in range(N):
    a = time.perf_counter(); p.submit_dev(w.data_ptr()); b = time.perf_counter(); p.collect(); c = time.perf_counter()
    ts += b - a; tc += c - b
tot = time.perf_counter() - t0
p.collect()
print("2^%d: per proof %.3f ms; in submit %.3f ms, in collect %.3f ms (collect includes waiting for the GPU)" % (k, tot / N * 1e3, ts / N * 1e3, tc / N * 1e3))
# collect with the GPU already idle = pure host tail
p.submit_dev(w.data_ptr()); torch.cuda.synchronize(); time.sleep(0.05)
a = time.perf_counter(); p.collect(); print("   host tail alone (GPU done): %.3f ms" % ((time.perf_counter() - a) * 1e3))
