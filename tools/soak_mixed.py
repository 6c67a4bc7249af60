"""Soak with a MIXED schedule: bursts of 1..8 submissions followed by drains, synchronous zk_prove calls in between, and (second
prover) batched submissions of 1..8 witnesses — the paths a lone small proof takes (B2's follow-ups on the finishing stream,
the (r, s) part of the tail before the wait, the tail pool of a batch) interleaved with the pipelined one.  EVERY proof is
compared with the synchronous proof of its (witness, r, s) made at the start.
    python tools/soak_mixed.py [log2n=16] [seconds=40] [seed=1]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth, views

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rnd = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
NW = 12
ws = [synth.make_witness(k, seed=i) for i in range(NW)]
rs = [(1000 + i, 77777 + 3 * i) for i in range(NW)]
p = views.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True)
want = [p.prove_host(ws[i], *rs[i]) for i in range(NW)]
assert len(set(want)) == NW
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
n = bad = nsync = nburst = 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < seconds / 2:
    what = rnd.random()
    if what < 0.3:                                   # one synchronous proof (nothing in flight: the lone path)
        i = rnd.randrange(NW)
        bad += p.prove_host(ws[i], *rs[i]) != want[i]; n += 1; nsync += 1
    else:                                            # a burst, drained completely or partly before the next one
        burst = [rnd.randrange(NW) for _ in range(rnd.randint(1, 8))]
        for i in burst:
            p.submit_host(ws[i], *rs[i])
        for i in burst:
            bad += p.collect() != want[i]; n += 1
        nburst += 1
    if rnd.random() < 0.05:
        time.sleep(0.002)
dt1 = time.perf_counter() - t0
print("2^%d unbatched prover: %d proofs in %.1f s (%d synchronous, %d bursts of 1..8), %d mismatches" % (k, n, dt1, nsync, nburst, bad), flush=True)

pb = views.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True, batch=8)
nb = badb = 0
t0 = time.perf_counter()
pend = []
while time.perf_counter() - t0 < seconds / 2:
    cnt = rnd.randint(1, 8)
    idx = [rnd.randrange(NW) for _ in range(cnt)]
    pb.submit_batch([ws[i] for i in idx], [rs[i] for i in idx]); pend.append(idx)
    if len(pend) > rnd.randint(0, 3):
        while pend:
            idx = pend.pop(0)
            got = pb.collect_batch(len(idx))
            badb += sum(g != want[i] for g, i in zip(got, idx)); nb += len(idx)
while pend:
    idx = pend.pop(0)
    got = pb.collect_batch(len(idx))
    badb += sum(g != want[i] for g, i in zip(got, idx)); nb += len(idx)
dt2 = time.perf_counter() - t0
torch.cuda.synchronize()
print("2^%d batch-of-8 prover: %d proofs in %.1f s in submissions of 1..8 witnesses, up to 4 submissions in flight, %d mismatches; GPU memory beyond the first prover's start: %+d MiB"
      % (k, nb, dt2, badb, (free0 - torch.cuda.mem_get_info()[0]) >> 20), flush=True)
sys.exit(1 if bad or badb else 0)
