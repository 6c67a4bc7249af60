"""What the FIRST proof of a fresh prover costs beyond a steady-state one (the one-shot CLI only ever runs the first):
    python tools/first_proof_probe.py [log2n=22] [precomp=0] [timings=1]
prints create time, then the wall time and the device stage times of proofs 1, 2, 3 (synchronous zk_prove, pageable witness)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth
import bench

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
precomp = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
ws = [synth.make_witness(k, seed=i + 1) for i in range(3)]
tm = len(sys.argv) <= 3 or bool(int(sys.argv[3]))
t0 = time.perf_counter()
p = bench.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=tm, precomp=precomp)
print("create %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for i in range(8):
    t0 = time.perf_counter()
    p.prove_host(ws[i % 3], 5, 7)
    dt = (time.perf_counter() - t0) * 1e3
    print("proof %d: %.2f ms   %s" % (i + 1, dt, p.timings() if tm else ""), flush=True)
