"""ctypes wrappers over an IN-MEMORY zkey view (numpy arrays instead of a .zkey file) — what bench.py, the GPU tests and the
tools use to drive synthetic workloads through the C-ABI (include/zkhip.h): zk_prover_create on a zk_zkey_view, the
submit/collect pairs, the sharded entry points and zk_multi_prover.  Mirrors Groth16::makeProver's 15 arguments
(/root/reference/src/groth16.hpp:104-121) field by field."""
import numpy as np


def view_from_workload(L, wl):
    """zk_zkey_view over the numpy arrays of a workload dict -> (view, arrays to keep alive until create returns)."""
    v = L.zk_zkey_view()
    v.nVars, v.nPublic, v.domainSize, v.nCoefs = wl["nVars"], wl["nPublic"], wl["domainSize"], wl["nCoefs"]
    keep = []
    for name in ("vk_alpha1", "vk_beta1", "vk_beta2", "vk_delta1", "vk_delta2", "coefs", "pointsA", "pointsB1",
                 "pointsB2", "pointsC", "pointsH"):
        a = np.ascontiguousarray(wl[name])
        keep.append(a)
        setattr(v, name, a.ctypes.data)
        if hasattr(v, name + "_bytes"):
            setattr(v, name + "_bytes", a.size)
    return v, keep


class MultiProverFromView:
    """zk_multi_prover (ONE proof over several devices of this process, chain partitioned for 2/4/8) over an in-memory view."""

    def __init__(self, zk, wl, devices, precomp=False):
        import ctypes as C
        from rapidsnark_old_amd import lib as L
        self.L, self.C, self.lib = L, C, L.load_library()
        v, keep = view_from_workload(L, wl)
        devs = (C.c_int32 * len(devices))(*devices)
        o = L.zk_opts(-1, 0, 1, 0, L.precomp_flags(precomp))
        self.h = C.c_void_p()
        L.check(self.lib.zk_multi_prover_create(C.byref(self.h), C.byref(v), devs, len(devices), C.byref(o)))
        ns, part = C.c_uint32(), C.c_uint32()
        L.check(self.lib.zk_multi_prover_info(self.h, C.byref(ns), C.byref(part)))
        self.n_shards, self.chain_partitioned = ns.value, bool(part.value)

    def prove(self, w, r, s):
        out = self.L.zk_proof()
        ra, sa = ProverFromView._k32(r), ProverFromView._k32(s)
        self.L.check(self.lib.zk_multi_prove(self.h, self.C.c_void_p(w.ctypes.data), ra.ctypes.data, sa.ctypes.data, self.C.byref(out)))
        return bytes(out)

    def close(self):
        if self.h.value:
            self.lib.zk_multi_prover_destroy(self.h)
            self.h = self.C.c_void_p()


class ProverFromView:
    """zk.Prover over an in-memory view (numpy arrays) instead of a .zkey file."""

    def __init__(self, zk, wl, device, shard_index, shard_count, window_bits, timings, precomp=False, partitioned_chain=False, batch=0,
                 sparse_witness=False):
        import ctypes as C
        from rapidsnark_old_amd import lib as L
        self.L = L
        self.lib = L.load_library()
        v, self.keep = view_from_workload(L, wl)
        o = L.zk_opts(device, shard_index, shard_count, window_bits,
                      (L.ZK_FLAG_TIMINGS if timings else 0) | L.precomp_flags(precomp)
                      | (L.ZK_FLAG_PARTITIONED_CHAIN if partitioned_chain else 0) | (L.ZK_FLAG_SPARSE_WITNESS if sparse_witness else 0), batch)
        self.h = C.c_void_p()
        L.check(self.lib.zk_prover_create(C.byref(self.h), C.byref(v), C.byref(o)))
        self.keep = []
        self.C = C

    def prove_dev(self, ptr, r=None, s=None):
        out = self.L.zk_proof()
        ra = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8) if r is not None else None
        sa = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8) if s is not None else None
        self.L.check(self.lib.zk_prove_dev(self.h, self.C.c_void_p(ptr), ra.ctypes.data if ra is not None else None,
                                           sa.ctypes.data if sa is not None else None, self.C.byref(out)))
        return bytes(out)

    def submit_dev(self, ptr):
        self.L.check(self.lib.zk_prove_dev_submit(self.h, self.C.c_void_p(ptr), None, None))

    def submit_batch(self, ws, rs=None):
        """zk_prove_batch_submit: numpy uint8 witnesses (kept alive by the caller until collected)."""
        n = len(ws)
        ptrs = (self.C.c_void_p * n)(*[w.ctypes.data for w in ws])
        rb = sb = None
        if rs is not None:
            rb = np.frombuffer(b"".join(int(r).to_bytes(32, "little") for r, _ in rs), dtype=np.uint8).copy()
            sb = np.frombuffer(b"".join(int(s_).to_bytes(32, "little") for _, s_ in rs), dtype=np.uint8).copy()
        self.L.check(self.lib.zk_prove_batch_submit(self.h, ptrs, n, rb.ctypes.data if rb is not None else None,
                                                    sb.ctypes.data if sb is not None else None))

    def collect_batch(self, n):
        out = (self.L.zk_proof * n)()
        self.L.check(self.lib.zk_prove_batch_collect(self.h, out, n))
        return [bytes(o) for o in out]

    @staticmethod
    def _k32(x):
        return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint8) if x is not None else None

    def submit_host(self, w, r=None, s=None):
        """zk_prove_submit: witness in host memory (numpy uint8 array, nVars*32 bytes)."""
        ra, sa = self._k32(r), self._k32(s)
        self.L.check(self.lib.zk_prove_submit(self.h, self.C.c_void_p(w.ctypes.data), ra.ctypes.data if ra is not None else None,
                                              sa.ctypes.data if sa is not None else None))

    def prove_host(self, w, r=None, s=None):
        """zk_prove: the reference's Prover::prove(wtns) — host witness, synchronous."""
        out = self.L.zk_proof()
        ra, sa = self._k32(r), self._k32(s)
        self.L.check(self.lib.zk_prove(self.h, self.C.c_void_p(w.ctypes.data), ra.ctypes.data if ra is not None else None,
                                       sa.ctypes.data if sa is not None else None, self.C.byref(out)))
        return bytes(out)

    def collect(self):
        out = self.L.zk_proof()
        self.L.check(self.lib.zk_prove_collect(self.h, self.C.byref(out)))
        return bytes(out)

    def collect_msm(self):
        out = self.L.zk_msm_sums()
        self.L.check(self.lib.zk_prove_msm_collect(self.h, self.C.byref(out)))
        return bytes(out)

    def prove_msm_dev(self, ptr):
        out = self.L.zk_msm_sums()
        self.L.check(self.lib.zk_prove_msm_dev(self.h, self.C.c_void_p(ptr), self.C.byref(out)))
        return bytes(out)

    def prove_finish(self, parts, r=None, s=None):
        arr = (self.L.zk_msm_sums * len(parts))(*[self.L.zk_msm_sums.from_buffer_copy(p) for p in parts])
        out = self.L.zk_proof()
        ra, sa = self._k32(r), self._k32(s)
        self.L.check(self.lib.zk_prove_finish(self.h, arr, len(parts), ra.ctypes.data if ra is not None else None,
                                              sa.ctypes.data if sa is not None else None, self.C.byref(out)))
        return bytes(out)

    def reserve(self, in_flight, host_witnesses=True):
        """zk_prover_reserve: every slot / lane a pipeline of `in_flight` proofs walks, allocated now (raises on out of memory)."""
        self.L.check(self.lib.zk_prover_reserve(self.h, in_flight, 1 if host_witnesses else 0))

    def info(self):
        """zk_prover_info: the launch plan chosen at create, as a dict."""
        return self.L.prover_info(self.lib, self.h)

    def timings(self):
        ms = (self.C.c_double * len(self.L.ZK_T_NAMES))()
        self.L.check(self.lib.zk_prover_timings(self.h, ms, len(self.L.ZK_T_NAMES)))
        return dict(zip(self.L.ZK_T_NAMES, list(ms)))
