"""Times operator-level MSMs at several sizes (debug/profiling helper)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth
g1 = synth.g1_gen_bytes()
p0, q = zk.g1_mul(g1, 123456789), zk.g1_mul(g1, 987654321)
for k in [int(a) for a in sys.argv[1:]] or [14, 16, 18]:
    n = 1 << k
    pts = zk.synth_chain_g1(n, p0, q)
    sc = synth.make_witness(k)
    t = time.time(); r = zk.msm_g1(pts, sc); dt = time.time() - t
    t = time.time(); r = zk.msm_g1(pts, sc); dt2 = time.time() - t
    print("k=%d msm_g1 %.1f ms (2nd %.1f ms)" % (k, dt * 1e3, dt2 * 1e3), flush=True)
