"""Small circuits: throughput of batched submissions (opts.batch = B: B witnesses per set of kernel launches) against
single ones, host witnesses, a collector thread, up to eight submissions in flight; every proof checked against the
unbatched prover.    python tools/batch_probe.py [log2n=14] [proofs=480] [batches=1,2,4,8] [window_bits=0]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 14
N = int(sys.argv[2]) if len(sys.argv) > 2 else 480
batches = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8").split(",")]
wb = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # window bits (0 = the library's choice)
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
wits = [synth.make_witness(k, seed=i) for i in range(8)]
rs = [(5 + i, 77 + i) for i in range(8)]
ref = bench.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True)
want = [ref.prove_host(w, *rs[i]) for i, w in enumerate(wits)]
ref.lib.zk_prover_destroy(ref.h)      # (an idle prover still owns a dozen streams: hardware queues are few)
for B in batches:
    p = bench.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=wb, timings=False, precomp=True, batch=B if B > 1 else 0)
    nsub = N // B
    depth = 8
    for rep in range(2):
        got = []
        sem_s, sem_c = threading.Semaphore(depth), threading.Semaphore(0)
        def collector():
            for _ in range(nsub):
                sem_c.acquire()
                got.extend(p.collect_batch(B))
                sem_s.release()
        th = threading.Thread(target=collector)
        th.start()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(nsub):
            sem_s.acquire()
            idx = [(j * B + t) % 8 for t in range(B)]
            p.submit_batch([wits[i] for i in idx], [rs[i] for i in idx])
            sem_c.release()
        th.join()
        dt = time.perf_counter() - t0
    ok = all(got[i] == want[i % 8] for i in range(nsub * B))
    print("2^%d%s batch %d: %.3f ms/proof (%.0f proofs/s); proofs equal the unbatched prover's: %s" % (k, (" c=%d" % wb) if wb else "", B, dt / (nsub * B) * 1e3, nsub * B / dt, ok))
    p.lib.zk_prover_destroy(p.h)
