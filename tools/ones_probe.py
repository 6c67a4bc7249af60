"""What do the scalars equal to 1 of a circuit witness cost beyond their additions?  Pipelined period (witness resident, six in
flight) of the circuit-shaped 2^k key with (a) the 80/15/5 witness, (b) the same witness with every 1 replaced by 0, (c) with
every 1 replaced by a random 16-bit value (same number of additions, spread over 2^15 buckets instead of piled into bucket 0):
(a) - (c) is what a dedicated plain-sum path for the ones could save at most, (a) - (b) what the ones cost in all.
    python tools/ones_probe.py [log2n=22] [sparse_witness=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth, views

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
sparse = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes(), shape="circuit")
nv = wl["nVars"]
p = views.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True, sparse_witness=sparse)
rng = np.random.default_rng(5)


def variants(seed):
    w = synth.make_witness(k, seed=seed, kind="realistic", n_vars=nv).reshape(nv, 32).copy()
    ones = (w[:, 1:].max(axis=1) == 0) & (w[:, 0] == 1)
    ones[0] = False
    b = w.copy(); b[ones, 0] = 0
    c = w.copy(); c[ones, :2] = rng.integers(1, 256, size=(int(ones.sum()), 2), dtype=np.uint8)
    return [torch.from_numpy(x.reshape(-1)).cuda() for x in (w, b, c)], int(ones.sum())


sets = [variants(s) for s in range(4)]
print("2^%d circuit-shaped, nVars %d, ones per witness ~%d, sparse-witness flag %d" % (k, nv, sets[0][1], sparse))
names = ["(a) 80/15/5 witness", "(b) ones -> 0", "(c) ones -> random 16-bit values"]
for rep in range(2):
    for v in range(3):
        ws = [s[0][v] for s in sets]
        for i in range(8):
            p.submit_dev(ws[i % 4].data_ptr())
        for i in range(8):
            p.collect()
        torch.cuda.synchronize()
        n, depth, fly = 40, 6, 0
        t0 = time.perf_counter()
        for i in range(n):
            p.submit_dev(ws[i % 4].data_ptr()); fly += 1
            if fly == depth:
                p.collect(); fly -= 1
        while fly:
            p.collect(); fly -= 1
        print("%-36s %.3f ms per proof" % (names[v], (time.perf_counter() - t0) / n * 1e3), flush=True)
