"""Large-size consistency run: plain vs window-precomputed tables must give the same proof bytes
(independent sort / table / reduction paths), plus timing.  usage: check_large.py <log2n>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth
import bench

k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
t = time.time()
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
print("generate %.1f s" % (time.time() - t), flush=True)
w = torch.from_numpy(synth.make_witness(k)).to("cuda:0")
torch.cuda.synchronize()
r, s = 0x123456789abcdef, (1 << 240) + 7
out = {}
for mode in (0, 1):
    t = time.time()
    p = bench.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=True, precomp=bool(mode))
    tc = time.time() - t
    proof = p.prove_dev(w.data_ptr(), r, s)
    ts = []
    for _ in range(3):
        t = time.time(); p.prove_dev(w.data_ptr(), r, s); ts.append(time.time() - t)
    # two proofs in flight (second ProofSlot) must give the same bytes
    p.submit_dev(w.data_ptr()); p.submit_dev(w.data_ptr())
    t = time.time(); p.collect(); p.submit_dev(w.data_ptr()); p.collect(); p.collect(); tp = (time.time() - t) / 2
    out[mode] = proof
    print("   pipelined period ~%.1f ms" % (tp * 1e3), flush=True)
    print("precomp=%d create %.2f s  prove %.1f ms  free HBM %.1f GB" % (mode, tc, min(ts) * 1e3, torch.cuda.mem_get_info()[0] / 1e9), flush=True)
    p.lib.zk_prover_destroy(p.h)
    del p
print("proofs identical:", out[0] == out[1])
print(zk.proof_to_json(out[1])[:120], "...")
sys.exit(0 if out[0] == out[1] else 1)
