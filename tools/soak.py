"""Soak: many proofs through one prover with host witnesses, six in flight, every N-th proof checked against a reference proof
of the same witness made at the start; GPU memory and the rate are sampled along the way.  A leak, a slot that is never
retired or a drifting result shows up here, not in a 30-step bench.   python tools/soak.py [log2n=20] [seconds=120]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth, views

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
p = views.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True)
p.reserve(6)
ws = [synth.make_witness(k, seed=i) for i in range(8)]
rs = [(1000 + i, 77777 + 3 * i) for i in range(8)]
want = [p.prove_host(ws[i], *rs[i]) for i in range(8)]
assert len(set(want)) == 8
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
n = bad = fly = 0
pending = []
last = t0
while time.perf_counter() - t0 < seconds:
    i = n % 8
    p.submit_host(ws[i], *rs[i]); pending.append(i); fly += 1; n += 1
    if fly == 6:
        got = p.collect(); j = pending.pop(0); fly -= 1
        bad += got != want[j]
    if time.perf_counter() - last > seconds / 6:
        last = time.perf_counter()
        print("  %6d proofs, %.1f proofs/s so far, %d mismatches, GPU memory in use beyond the start: %+d MiB"
              % (n, n / (last - t0), bad, (free0 - torch.cuda.mem_get_info()[0]) >> 20), flush=True)
while fly:
    got = p.collect(); j = pending.pop(0); fly -= 1
    bad += got != want[j]
dt = time.perf_counter() - t0
torch.cuda.synchronize()
print("2^%d: %d proofs in %.1f s = %.1f proofs/s; EVERY proof compared with the synchronous proof of its (witness, r, s): %d mismatches; GPU memory drift %+d MiB"
      % (k, n, dt, n / dt, bad, (free0 - torch.cuda.mem_get_info()[0]) >> 20))
sys.exit(1 if bad else 0)
