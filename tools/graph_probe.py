"""Throughput of a small-circuit prover with and without captured HIP graphs (ZKHIP_GRAPH=1), eight in flight,
host witnesses, a collector thread like bench.py's; every proof checked against the synchronous path.
    python tools/graph_probe.py [log2n=14] [proofs=400]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 14
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
p = bench.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True)
wits = [synth.make_witness(k, seed=i) for i in range(4)]
want = [p.prove_host(w, 5 + i, 77 + i) for i, w in enumerate(wits)]
depth = 8
for rep in range(2):
    got = []
    def collector():
        for _ in range(N):
            sem_c.acquire()
            got.append(p.collect())
            sem_s.release()
    sem_s, sem_c = threading.Semaphore(depth), threading.Semaphore(0)
    th = threading.Thread(target=collector)
    th.start()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        sem_s.acquire()
        p.submit_host(wits[i % 4], 5 + i % 4, 77 + i % 4)
        sem_c.release()
    th.join()
    dt = time.perf_counter() - t0
    ok = all(got[i] == want[i % 4] for i in range(N))
    print("2^%d graph=%s: %.3f ms/proof (%.0f proofs/s), all proofs equal the synchronous ones: %s" % (k, os.environ.get("ZKHIP_GRAPH", "0"), dt / N * 1e3, N / dt, ok))
