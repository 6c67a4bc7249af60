"""ctypes wrapper of oracle/c/zk_oracle.c (libzkoracle.so) — TEST INFRASTRUCTURE ONLY.

The CPU restatement of rapidsnark's prove() algorithm: bit-exact comparator at sizes the
Python big-int oracle cannot reach, and the `cpu_baseline` of bench.py.  Never imported by
the product package.  PARITY UNPINNED at the reference boundary (see zk_oracle.c header);
pinned against oracle/bn254.py + trapdoor check by tests/test_oracle_c.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class oracle_zkey_view(C.Structure):
    _fields_ = [("nVars", C.c_uint32), ("nPublic", C.c_uint32), ("domainSize", C.c_uint32), ("nCoefs", C.c_uint64),
                ("vk_alpha1", C.c_void_p), ("vk_beta1", C.c_void_p), ("vk_beta2", C.c_void_p),
                ("vk_delta1", C.c_void_p), ("vk_delta2", C.c_void_p), ("coefs", C.c_void_p),
                ("pointsA", C.c_void_p), ("pointsB1", C.c_void_p), ("pointsB2", C.c_void_p),
                ("pointsC", C.c_void_p), ("pointsH", C.c_void_p)]


def cpu_has_adx():
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return " adx" in flags and " bmi2" in flags


def build(march=None, out=None, adx=False):
    """Compile the C restatement (gcc, OpenMP).  march='native' on the box that times it; adx=True: the field product
    with mulx / adcx / adox (-DZK_ORACLE_ADX; the CPU must have BMI2 + ADX)."""
    args = ["make", "-s", "-C", _HERE]
    if march:
        args.append("MARCH=%s" % march)
    if adx:
        args.append("EXTRA=-DZK_ORACLE_ADX")
    if out:
        args.append("OUT=%s" % out)
        if os.path.exists(os.path.join(_HERE, out)):
            os.remove(os.path.join(_HERE, out))
    subprocess.check_call(args)


def load(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or os.path.join(_HERE, "_build", "libzkoracle.so")
    if not os.path.exists(p):
        build()
    lib = C.CDLL(p)
    v = C.c_void_p
    lib.oracle_num_threads.restype = C.c_int
    if hasattr(lib, "oracle_variant"):
        lib.oracle_variant.restype = C.c_char_p
    lib.oracle_set_num_threads.argtypes = [C.c_int]
    lib.oracle_set_num_threads.restype = None
    for n in ("oracle_fr_mul_vec", "oracle_fq_mul_vec"):
        getattr(lib, n).argtypes = [v, v, v, C.c_uint64]
        getattr(lib, n).restype = None
    lib.oracle_fr_fft.argtypes = [v, C.c_uint64, C.c_int]
    lib.oracle_msm_g1.argtypes = [v, v, v, C.c_uint64]
    lib.oracle_msm_g2.argtypes = [v, v, v, C.c_uint64]
    lib.oracle_compute_h.argtypes = [C.POINTER(oracle_zkey_view), v, v]
    lib.oracle_prove_msm.argtypes = [C.POINTER(oracle_zkey_view), v, v]
    lib.oracle_prove.argtypes = [C.POINTER(oracle_zkey_view), v, v, v, v]
    lib.oracle_chain_g1.argtypes = [v, C.c_uint64, v, v, v]
    lib.oracle_chain_g2.argtypes = [v, C.c_uint64, v, v, v]
    lib.oracle_chainp_g1.argtypes = [v, C.c_uint64, v, v]
    lib.oracle_chainp_g2.argtypes = [v, C.c_uint64, v, v]
    lib.oracle_chainp_g1.restype = None
    lib.oracle_chainp_g2.restype = None
    lib.oracle_g1_mul.argtypes = [v, v, v]
    lib.oracle_g2_mul.argtypes = [v, v, v]
    for n in ("oracle_fr_fft", "oracle_msm_g1", "oracle_msm_g2", "oracle_compute_h", "oracle_prove_msm", "oracle_prove",
              "oracle_chain_g1", "oracle_chain_g2", "oracle_g1_mul", "oracle_g2_mul"):
        getattr(lib, n).restype = None
    if path is None:
        _LIB = lib
    return lib


def _a(b):
    return b if isinstance(b, np.ndarray) else np.frombuffer(b, dtype=np.uint8)


def _p(a):
    return C.c_void_p(a.ctypes.data)


def num_threads():
    return load().oracle_num_threads()


def set_num_threads(n):
    load().oracle_set_num_threads(int(n))


def fr_mul_vec(a, b):
    a, b = _a(a), _a(b)
    out = np.empty(a.size, dtype=np.uint8)
    load().oracle_fr_mul_vec(_p(out), _p(a), _p(b), a.size // 32)
    return out.tobytes()


def fq_mul_vec(a, b):
    a, b = _a(a), _a(b)
    out = np.empty(a.size, dtype=np.uint8)
    load().oracle_fq_mul_vec(_p(out), _p(a), _p(b), a.size // 32)
    return out.tobytes()


def fr_fft(data, inverse=False):
    a = _a(data).copy()
    load().oracle_fr_fft(_p(a), a.size // 32, 1 if inverse else 0)
    return a.tobytes()


def msm_g1(bases, scalars):
    bases, scalars = _a(bases), _a(scalars)
    out = np.zeros(64, dtype=np.uint8)
    load().oracle_msm_g1(_p(out), _p(bases), _p(scalars), scalars.size // 32)
    return out.tobytes()


def msm_g2(bases, scalars):
    bases, scalars = _a(bases), _a(scalars)
    out = np.zeros(128, dtype=np.uint8)
    load().oracle_msm_g2(_p(out), _p(bases), _p(scalars), scalars.size // 32)
    return out.tobytes()


class ZkeyView:
    """Parses a zkey image (same section rules as the reference readers) into the view the C code takes.
    Also accepts a dict of numpy arrays (synthetic workloads, oracle/synth.py)."""

    def __init__(self, zkey):
        self.keep = []
        v = oracle_zkey_view()
        if isinstance(zkey, dict):
            d = zkey
            v.nVars, v.nPublic, v.domainSize, v.nCoefs = d["nVars"], d["nPublic"], d["domainSize"], d["nCoefs"]
            for k in ("vk_alpha1", "vk_beta1", "vk_beta2", "vk_delta1", "vk_delta2", "coefs", "pointsA", "pointsB1",
                      "pointsB2", "pointsC", "pointsH"):
                a = _a(d[k])
                self.keep.append(a)
                setattr(v, k, a.ctypes.data)
        else:
            from . import groth16_ref as g
            import struct
            secs = g.read_binfile(bytes(zkey), b"zkey", 1)
            s2 = secs[2][0]
            n8q = struct.unpack_from("<I", s2, 0)[0]
            pos = 4 + n8q
            n8r = struct.unpack_from("<I", s2, pos)[0]
            pos += 4 + n8r
            v.nVars, v.nPublic, v.domainSize = struct.unpack_from("<III", s2, pos)
            pos += 12
            v.nCoefs = len(secs[4][0]) // (12 + n8r)

            def put(name, b):
                a = np.frombuffer(bytes(b), dtype=np.uint8)
                self.keep.append(a)
                setattr(v, name, a.ctypes.data if a.size else None)

            put("vk_alpha1", s2[pos:pos + 64]); pos += 64
            put("vk_beta1", s2[pos:pos + 64]); pos += 64
            put("vk_beta2", s2[pos:pos + 128]); pos += 128
            pos += 128                                    # gamma2 (unused by the prover)
            put("vk_delta1", s2[pos:pos + 64]); pos += 64
            put("vk_delta2", s2[pos:pos + 128])
            for name, sec in (("coefs", 4), ("pointsA", 5), ("pointsB1", 6), ("pointsB2", 7), ("pointsC", 8), ("pointsH", 9)):
                put(name, secs[sec][0])
        self.v = v


def compute_h(view, wtns_vals):
    w = _a(wtns_vals)
    out = np.empty(view.v.domainSize * 32, dtype=np.uint8)
    load().oracle_compute_h(C.byref(view.v), _p(w), _p(out))
    return out.tobytes()


def prove_msm(view, wtns_vals):
    w = _a(wtns_vals)
    out = np.zeros(384, dtype=np.uint8)
    load().oracle_prove_msm(C.byref(view.v), _p(w), _p(out))
    return out.tobytes()


def prove(view, wtns_vals, r, s):
    w = _a(wtns_vals)
    rb = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8)
    sb = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8)
    out = np.zeros(256, dtype=np.uint8)
    load().oracle_prove(C.byref(view.v), _p(w), _p(rb), _p(sb), _p(out))
    return out.tobytes()


def _k32(k):
    return np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).copy()


def chain_g1(n, gen_bytes, k0, kq):
    out = np.zeros(n * 64, dtype=np.uint8)
    g, a, b = _a(gen_bytes).copy(), _k32(k0), _k32(kq)      # keep alive across the call
    load().oracle_chain_g1(_p(out), n, _p(g), _p(a), _p(b))
    return out


def chain_g2(n, gen_bytes, k0, kq):
    out = np.zeros(n * 128, dtype=np.uint8)
    g, a, b = _a(gen_bytes).copy(), _k32(k0), _k32(kq)
    load().oracle_chain_g2(_p(out), n, _p(g), _p(a), _p(b))
    return out


def g1_mul(p, k):
    out = np.zeros(64, dtype=np.uint8)
    pp, kk = _a(p).copy(), _k32(k)
    load().oracle_g1_mul(_p(out), _p(pp), _p(kk))
    return out.tobytes()


def g2_mul(p, k):
    out = np.zeros(128, dtype=np.uint8)
    pp, kk = _a(p).copy(), _k32(k)
    load().oracle_g2_mul(_p(out), _p(pp), _p(kk))
    return out.tobytes()


def chainp_g1(n, p0, q):
    """out[i] = P0 + i*Q (same contract as the product's zk_synth_chain_g1)."""
    out = np.zeros(n * 64, dtype=np.uint8)
    a, b = _a(p0).copy(), _a(q).copy()
    load().oracle_chainp_g1(_p(out), n, _p(a), _p(b))
    return out


def chainp_g2(n, p0, q):
    out = np.zeros(n * 128, dtype=np.uint8)
    a, b = _a(p0).copy(), _a(q).copy()
    load().oracle_chainp_g2(_p(out), n, _p(a), _p(b))
    return out


def synth_workload(k, n_public=1, seed=0):
    """The synthetic zkey family of rapidsnark-old_amd/synth.py, generated entirely on the CPU."""
    import importlib
    from . import bn254 as bn
    synth = importlib.import_module("rapidsnark_old_amd.synth")
    return synth.workload(k, chainp_g1, chainp_g2, g1_mul, g2_mul, bn.g1_to_bytes(bn.G1.gen), bn.g2_to_bytes(bn.G2.gen),
                          n_public=n_public, seed=seed)
