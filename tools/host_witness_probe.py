"""PCIe-inclusive rate: zk_prove with the witness in (pageable / pinned) host memory vs zk_prove_dev.
    python tools/host_witness_probe.py [log2n=22]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth, lib as L

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
wl = synth.workload(k, zk.synth_chain_g1, zk.synth_chain_g2, zk.g1_mul, zk.g2_mul, synth.g1_gen_bytes(), synth.g2_gen_bytes())
p = bench.ProverFromView(zk, wl, device=0, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=True)
w = synth.make_witness(k, seed=0)
wd = torch.from_numpy(w).cuda()
wp = torch.from_numpy(w).pin_memory()
def timed(fn, n=6):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
out = L.zk_proof()
print("zk_prove_dev (HBM witness)      %.2f ms" % timed(lambda: p.prove_dev(wd.data_ptr())))
print("zk_prove (pageable host witness) %.2f ms" % timed(lambda: L.check(p.lib.zk_prove(p.h, C.c_void_p(w.ctypes.data), None, None, C.byref(out)))))
print("zk_prove (pinned host witness)   %.2f ms" % timed(lambda: L.check(p.lib.zk_prove(p.h, C.c_void_p(wp.data_ptr()), None, None, C.byref(out)))))
