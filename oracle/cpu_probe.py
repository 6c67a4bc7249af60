"""Stage timing of the C restatement on the host cores (debug helper of the oracle; test infrastructure)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import c_oracle as co
from rapidsnark_old_amd import synth
try:
    co.build(march="native", out="_build/libzkoracle_native.so")
    co._LIB = co.load(os.path.join(os.path.dirname(co.__file__), "_build", "libzkoracle_native.so"))
except Exception as e:
    print("native build failed", e)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
print("threads", co.num_threads())
t = time.time(); wl = co.synth_workload(k); print("gen %.2f" % (time.time() - t))
v = co.ZkeyView(wl); w = synth.make_witness(k)
t = time.time(); h = co.compute_h(v, w); print("compute_h %.3f" % (time.time() - t))
n = 1 << k
x = np.frombuffer(h, dtype=np.uint8).copy()
t = time.time(); co.fr_fft(x); print("one fft %.3f" % (time.time() - t))
t = time.time(); co.msm_g1(wl["pointsA"], w); print("msm_g1 %.3f" % (time.time() - t))
t = time.time(); co.msm_g1(wl["pointsA"], w); print("msm_g1 again %.3f" % (time.time() - t))
t = time.time(); co.msm_g2(wl["pointsB2"], w); print("msm_g2 %.3f" % (time.time() - t))
t = time.time(); co.prove(v, w, 1, 2); print("prove %.3f" % (time.time() - t))
