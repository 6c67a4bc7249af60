#!/usr/bin/env python3
"""The coset-evaluation pipeline alone (zk_fr_abc_to_h at 2^k: a, b, c = a o b through ifft / coset shift / fft, then
a.b - c) — for rocprofv3 runs that look at the NTT kernels only.   python tools/ntt_probe.py [k=22] [repeats=3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rapidsnark_old_amd as zk
from rapidsnark_old_amd import synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(1)
a = synth.random_fr_bytes(rng, 1 << k).reshape(-1)
b = synth.random_fr_bytes(rng, 1 << k).reshape(-1)
for i in range(rep):
    t = time.perf_counter()
    h = zk.fr_abc_to_h(a, b)
    print("2^%d: zk_fr_abc_to_h %.1f ms (incl. tables, uploads)" % (k, (time.perf_counter() - t) * 1e3))
