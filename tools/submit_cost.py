"""What one submission costs the HOST at a small size: time inside submit_host, inside collect, and the period, with D proofs
in flight on one thread (submit, then collect the oldest once D are in flight).   python tools/submit_cost.py [log2n=14] [depth=8] [n=400]
If submit time ~ period, the size is bound by the launching thread, not by the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth, views

k = int(sys.argv[1]) if len(sys.argv) > 1 else 14
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 400
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
p = views.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True)
p.reserve(depth)
ws = [synth.make_witness(k, seed=i) for i in range(8)]
for rep in range(2):
    for i in range(depth):
        p.submit_host(ws[i % 8], 5, 7)
    for i in range(depth):
        p.collect()
torch.cuda.synchronize()
ts = tc = 0.0
fly = 0
t0 = time.perf_counter()
for i in range(n):
    a = time.perf_counter()
    p.submit_host(ws[i % 8], 5, 7)
    b = time.perf_counter()
    ts += b - a
    fly += 1
    if fly == depth:
        p.collect(); fly -= 1
        tc += time.perf_counter() - b
while fly:
    p.collect(); fly -= 1
dt = time.perf_counter() - t0
print("2^%d, %d in flight, %s: submit %.3f ms, collect (wait + host tail) %.3f ms, period %.3f ms per proof"
      % (k, depth, " ".join("%s=%s" % (e, os.environ[e]) for e in ("ZKHIP_GRAPH", "ZKHIP_LANES") if e in os.environ) or "defaults", 1e3 * ts / n, 1e3 * tc / n, 1e3 * dt / n))
