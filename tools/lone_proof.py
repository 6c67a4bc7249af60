"""ONE synchronous zk_prove at a time (host witness, the reference's main_prover.cpp:75), with pauses in between, so that a
rocprofv3 --kernel-trace of this process shows every lone proof as its own cluster (tools/lone_timeline.py reads it):
    python tools/lone_proof.py [log2n=22] [proofs=4] [precomp=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth
from rapidsnark_old_amd.views import ProverFromView

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
nproofs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
precomp = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
ws = [synth.make_witness(k, seed=i + 1) for i in range(3)]
p = ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=precomp)
for i in range(3):
    p.prove_host(ws[i % 3])
for i in range(nproofs):
    time.sleep(0.1)
    t0 = time.perf_counter()
    p.prove_host(ws[i % 3])
    print("lone proof %d: %.2f ms wall" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
