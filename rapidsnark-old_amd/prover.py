"""Groth16::Prover mirror over the C-ABI (reference src/groth16.hpp:37-121,
src/main_prover.cpp:23-103).  Prover(zkey) == makeProver; prove(wtns) == Prover::prove."""
import ctypes as C

import numpy as np

from . import lib as L
from .binfile import open_existing
from .zkey import load_zkey_header
from .wtns import load_wtns_header

BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # main_prover.cpp:34


def _zkey_view(zkey, keep):
    """(header, zk_zkey_view) of a snarkjs .zkey given as path or bytes; `keep` collects the numpy views
    the pointers refer to (main_prover.cpp:42-72)."""
    f = open_existing(zkey, "zkey", 1)
    h = load_zkey_header(f)
    if h.rPrime != BN254_R:
        raise ValueError("zkey curve not supported")            # main_prover.cpp:46-48
    v = L.zk_zkey_view()
    v.nVars, v.nPublic, v.domainSize, v.nCoefs = h.nVars, h.nPublic, h.domainSize, h.nCoefs

    def ptr(b):
        a = np.frombuffer(b, dtype=np.uint8)
        keep.append(a)
        return a.ctypes.data if a.size else None

    v.vk_alpha1, v.vk_beta1, v.vk_beta2 = ptr(h.vk_alpha1), ptr(h.vk_beta1), ptr(h.vk_beta2)
    v.vk_delta1, v.vk_delta2 = ptr(h.vk_delta1), ptr(h.vk_delta2)
    for name, sec in (("coefs", 4), ("pointsA", 5), ("pointsB1", 6), ("pointsB2", 7), ("pointsC", 8), ("pointsH", 9)):
        data = f.getSectionData(sec)                              # main_prover.cpp:67-72
        setattr(v, name, ptr(data))
        setattr(v, name + "_bytes", len(data))
    return h, v


class Prover:
    def __init__(self, zkey, device=-1, shard_index=0, shard_count=1, window_bits=0, timings=False, precomp=False,
                 partitioned_chain=False, batch=0):
        """zkey: path or bytes of a snarkjs .zkey (version <= 1, main_prover.cpp:42).
        batch >= 2 (with precomp): submit_batch / collect_batch prove up to `batch` witnesses per submission."""
        self._lib = L.load_library()
        self._keep = []
        self.header, v = _zkey_view(zkey, self._keep)
        o = L.zk_opts(device, shard_index, shard_count, window_bits,
                      (L.ZK_FLAG_TIMINGS if timings else 0) | L.precomp_flags(precomp)
                      | (L.ZK_FLAG_PARTITIONED_CHAIN if partitioned_chain else 0), batch)
        self._h = C.c_void_p()
        L.check(self._lib.zk_prover_create(C.byref(self._h), C.byref(v), C.byref(o)))
        self._keep = []     # host image may be released after create (include/zkhip.h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.zk_prover_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    @staticmethod
    def _rs(x):
        if x is None:
            return None, None
        a = np.frombuffer(int(x).to_bytes(32, "little") if not isinstance(x, (bytes, bytearray)) else bytes(x), dtype=np.uint8)
        return a, C.c_void_p(a.ctypes.data)

    def _wtns_values(self, wtns):
        """Accepts a .wtns path/bytes (binfile) or raw nVars*32 value bytes."""
        if isinstance(wtns, str) or bytes(wtns[:4]) == b"wtns":
            f = open_existing(wtns, "wtns", 2)                        # main_prover.cpp:50
            wh = load_wtns_header(f)
            if wh.prime != BN254_R:
                raise ValueError("different wtns curve")             # main_prover.cpp:53-55
            if wh.nVars != self.header.nVars:
                raise ValueError("wtns has %d variables, zkey expects %d" % (wh.nVars, self.header.nVars))   # quirk Q8
            vals = f.getSectionData(2)
        else:
            vals = wtns
        a = np.frombuffer(vals, dtype=np.uint8)
        if a.size != self.header.nVars * 32:
            raise ValueError("witness size mismatch")
        return a

    def prove(self, wtns, r=None, s=None):
        """-> 256 proof bytes (A|B|C affine Montgomery, == Proof<Engine>)."""
        a = self._wtns_values(wtns)
        (ra, rp), (sa, sp) = self._rs(r), self._rs(s)
        out = L.zk_proof()
        L.check(self._lib.zk_prove(self._h, C.c_void_p(a.ctypes.data), rp, sp, C.byref(out)))
        return bytes(out)

    def prove_dev(self, d_wtns_ptr, r=None, s=None):
        (ra, rp), (sa, sp) = self._rs(r), self._rs(s)
        out = L.zk_proof()
        L.check(self._lib.zk_prove_dev(self._h, C.c_void_p(d_wtns_ptr), rp, sp, C.byref(out)))
        return bytes(out)

    def submit_dev(self, d_wtns_ptr, r=None, s=None):
        """Throughput mode (zk_prove_dev_submit): enqueue one proof; at most two in flight.  The device
        witness must stay valid until the matching collect()."""
        (ra, rp), (sa, sp) = self._rs(r), self._rs(s)
        L.check(self._lib.zk_prove_dev_submit(self._h, C.c_void_p(d_wtns_ptr), rp, sp))

    def submit(self, wtns, r=None, s=None):
        """Throughput mode with the witness in HOST memory (zk_prove_submit): .wtns path/bytes, raw value
        bytes or a numpy uint8 array (e.g. a lib.PinnedBuffer's .array, which is then read in place and
        must stay untouched until the matching collect())."""
        a = wtns if isinstance(wtns, np.ndarray) else self._wtns_values(wtns)
        if a.size != self.header.nVars * 32:
            raise ValueError("witness size mismatch")
        (ra, rp), (sa, sp) = self._rs(r), self._rs(s)
        L.check(self._lib.zk_prove_submit(self._h, C.c_void_p(a.ctypes.data), rp, sp))
        self._pending = getattr(self, "_pending", []) + [a]      # the library reads it until the proof is collected

    def submit_batch(self, wtns_list, rs=None):
        """zk_prove_batch_submit: several witnesses (each as for submit()) in ONE submission of a prover created
        with batch >= len(wtns_list); rs = list of (r, s) or None."""
        arrs = [w if isinstance(w, np.ndarray) else self._wtns_values(w) for w in wtns_list]
        for a in arrs:
            if a.size != self.header.nVars * 32:
                raise ValueError("witness size mismatch")
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        rp = sp = None
        keep = [arrs]
        if rs is not None:
            rb = np.frombuffer(b"".join(int(r).to_bytes(32, "little") for r, _ in rs), dtype=np.uint8).copy()
            sb = np.frombuffer(b"".join(int(s_).to_bytes(32, "little") for _, s_ in rs), dtype=np.uint8).copy()
            rp, sp = C.c_void_p(rb.ctypes.data), C.c_void_p(sb.ctypes.data)
            keep += [rb, sb]
        L.check(self._lib.zk_prove_batch_submit(self._h, ptrs, n, rp, sp))
        self._pending = getattr(self, "_pending", []) + [keep]

    def collect_batch(self, count):
        """-> list of `count` proofs of the OLDEST submission (zk_prove_batch_collect)."""
        out = (L.zk_proof * count)()
        L.check(self._lib.zk_prove_batch_collect(self._h, out, count))
        self._pending = getattr(self, "_pending", [])[1:]
        return [bytes(o) for o in out]

    def collect(self):
        """-> proof bytes of the OLDEST submitted proof (zk_prove_collect)."""
        out = L.zk_proof()
        L.check(self._lib.zk_prove_collect(self._h, C.byref(out)))
        self._pending = getattr(self, "_pending", [])[1:]
        return bytes(out)

    def collect_msm(self):
        """Sharded provers: partial sums of the OLDEST submitted proof (zk_prove_msm_collect)."""
        out = L.zk_msm_sums()
        L.check(self._lib.zk_prove_msm_collect(self._h, C.byref(out)))
        return bytes(out)

    def prove_msm(self, wtns):
        a = self._wtns_values(wtns)
        out = L.zk_msm_sums()
        L.check(self._lib.zk_prove_msm(self._h, C.c_void_p(a.ctypes.data), C.byref(out)))
        return bytes(out)

    def prove_msm_dev(self, d_wtns_ptr):
        out = L.zk_msm_sums()
        L.check(self._lib.zk_prove_msm_dev(self._h, C.c_void_p(d_wtns_ptr), C.byref(out)))
        return bytes(out)

    def prove_finish(self, partials, r=None, s=None):
        arr = (L.zk_msm_sums * len(partials))(*[L.zk_msm_sums.from_buffer_copy(p) for p in partials])
        (ra, rp), (sa, sp) = self._rs(r), self._rs(s)
        out = L.zk_proof()
        L.check(self._lib.zk_prove_finish(self._h, arr, len(partials), rp, sp, C.byref(out)))
        return bytes(out)

    def reserve(self, in_flight, host_witnesses=True):
        """zk_prover_reserve: allocate NOW every proof slot and lane a pipeline of `in_flight` proofs walks (a server's
        start-up check: out of device memory is raised here, not by a proof later)."""
        L.check(self._lib.zk_prover_reserve(self._h, in_flight, 1 if host_witnesses else 0))

    def info(self):
        """zk_prover_info: the launch plan chosen at create, as a dict."""
        return L.prover_info(self._lib, self._h)

    def timings(self):
        ms = (C.c_double * len(L.ZK_T_NAMES))()
        L.check(self._lib.zk_prover_timings(self._h, ms, len(L.ZK_T_NAMES)))
        return dict(zip(L.ZK_T_NAMES, list(ms)))


class MultiProver:
    """One proof on several GPUs of this process (zk_multi_prover): every MSM table sharded by point
    range, the A.w/B.w rows and the six transforms partitioned the same way (2, 4 or 8 devices), partial
    sums added on the host.  devices may repeat an ordinal (all shards on one GPU: test boxes)."""

    def __init__(self, zkey, devices, window_bits=0, precomp=False):
        self._lib = L.load_library()
        keep = []
        self.header, v = _zkey_view(zkey, keep)
        devs = (C.c_int32 * len(devices))(*devices)
        o = L.zk_opts(-1, 0, 1, window_bits, L.precomp_flags(precomp))
        self._h = C.c_void_p()
        L.check(self._lib.zk_multi_prover_create(C.byref(self._h), C.byref(v), devs, len(devices), C.byref(o)))
        ns, part = C.c_uint32(), C.c_uint32()
        L.check(self._lib.zk_multi_prover_info(self._h, C.byref(ns), C.byref(part)))
        self.n_shards, self.chain_partitioned = ns.value, bool(part.value)
        self._pending = []

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.zk_multi_prover_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _vals(self, wtns):
        a = wtns if isinstance(wtns, np.ndarray) else Prover._wtns_values(self, wtns)
        if a.size != self.header.nVars * 32:
            raise ValueError("witness size mismatch")
        return a

    def prove(self, wtns, r=None, s=None):
        a = self._vals(wtns)
        (ra, rp), (sa, sp) = Prover._rs(r), Prover._rs(s)
        out = L.zk_proof()
        L.check(self._lib.zk_multi_prove(self._h, C.c_void_p(a.ctypes.data), rp, sp, C.byref(out)))
        return bytes(out)

    def submit(self, wtns, r=None, s=None):
        a = self._vals(wtns)
        (ra, rp), (sa, sp) = Prover._rs(r), Prover._rs(s)
        L.check(self._lib.zk_multi_prove_submit(self._h, C.c_void_p(a.ctypes.data), rp, sp))
        self._pending.append(a)

    def collect(self):
        out = L.zk_proof()
        L.check(self._lib.zk_multi_prove_collect(self._h, C.byref(out)))
        self._pending = self._pending[1:]
        return bytes(out)


def prove_files(zkey_path, wtns_path, proof_path, public_path, r=None, s=None):
    """`prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>` (main_prover.cpp:23-103)."""
    p = Prover(zkey_path)
    try:
        vals = p._wtns_values(wtns_path)
        proof = p.prove(vals, r, s)
        with open(proof_path, "w") as f:
            f.write(L.proof_to_json(proof))
        with open(public_path, "w") as f:
            f.write(L.public_to_json(vals, p.header.nPublic))
    finally:
        p.close()
