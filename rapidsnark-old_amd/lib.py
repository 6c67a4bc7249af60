"""ctypes binding of include/zkhip.h.  Fails loudly when libzkhip.so is absent — there is
no Python or CPU fallback for any compute entry point."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class ZkHipError(RuntimeError):
    pass


def library_path():
    # ZKHIP_LIB: another build of the same library (A/B measurements on one box); never a fallback
    return os.environ.get("ZKHIP_LIB") or os.path.join(_HERE, "libzkhip.so")


class zk_zkey_view(C.Structure):
    _fields_ = [("nVars", C.c_uint32), ("nPublic", C.c_uint32), ("domainSize", C.c_uint32), ("nCoefs", C.c_uint64),
                ("vk_alpha1", C.c_void_p), ("vk_beta1", C.c_void_p), ("vk_beta2", C.c_void_p),
                ("vk_delta1", C.c_void_p), ("vk_delta2", C.c_void_p), ("coefs", C.c_void_p),
                ("pointsA", C.c_void_p), ("pointsB1", C.c_void_p), ("pointsB2", C.c_void_p),
                ("pointsC", C.c_void_p), ("pointsH", C.c_void_p),
                ("coefs_bytes", C.c_uint64), ("pointsA_bytes", C.c_uint64), ("pointsB1_bytes", C.c_uint64),
                ("pointsB2_bytes", C.c_uint64), ("pointsC_bytes", C.c_uint64), ("pointsH_bytes", C.c_uint64)]


class zk_opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
                ("window_bits", C.c_uint32), ("flags", C.c_uint32), ("batch", C.c_uint32)]


class zk_proof(C.Structure):
    _fields_ = [("A", C.c_uint8 * 64), ("B", C.c_uint8 * 128), ("C", C.c_uint8 * 64)]


class zk_msm_sums(C.Structure):
    _fields_ = [("pih", C.c_uint8 * 64), ("pi_a", C.c_uint8 * 64), ("pib1", C.c_uint8 * 64),
                ("pi_b", C.c_uint8 * 128), ("pi_c", C.c_uint8 * 64)]


class zk_prover_plan(C.Structure):
    _fields_ = [("size", C.c_uint32), ("window_bits_h", C.c_uint32), ("windows_h", C.c_uint32), ("window_bits_w", C.c_uint32),
                ("windows_w", C.c_uint32), ("precomputed_tables", C.c_uint32), ("msm_a_b1_c_one_launch", C.c_uint32), ("lanes", C.c_uint32),
                ("follow_up_streams", C.c_uint32), ("max_in_flight", C.c_uint32), ("depth_host_witness", C.c_uint32),
                ("depth_resident_witness", C.c_uint32), ("batch", C.c_uint32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
                ("chain_partitioned", C.c_uint32), ("device_bytes_in_use", C.c_uint64), ("device_bytes_total", C.c_uint64),
                ("kernel_launches_last_proof", C.c_uint64), ("table_rows_h", C.c_uint32), ("table_rows_w", C.c_uint32),
                ("bucket_sets_h", C.c_uint32), ("bucket_sets_w", C.c_uint32)]


def prover_info(lib, handle):
    """zk_prover_info -> dict: the launch plan zk_prover_create chose (window bits, A|B1|C in one launch, lanes, depths)."""
    plan = zk_prover_plan()
    plan.size = C.sizeof(zk_prover_plan)
    check(lib.zk_prover_info(handle, C.byref(plan)))
    return {name: int(getattr(plan, name)) for name, _ in zk_prover_plan._fields_ if name != "size"}


ZK_FLAG_TIMINGS = 1
ZK_FLAG_PRECOMP = 2
ZK_FLAG_PARTITIONED_CHAIN = 4
ZK_FLAG_SPARSE_WITNESS = 8
ZK_FLAG_PRECOMP_HALF = 16


def precomp_flags(precomp):
    """The `precomp` option of the bindings -> ZK_FLAG_*: False / 0 = tables as in the zkey, True / 1 = a table row per
    window (ZK_FLAG_PRECOMP), 2 = a row per second window (ZK_FLAG_PRECOMP_HALF: 7 instead of 13 x the table memory)."""
    mode = int(precomp)
    if mode not in (0, 1, 2):
        raise ValueError("precomp: 0, 1 or 2")
    return (0, ZK_FLAG_PRECOMP, ZK_FLAG_PRECOMP | ZK_FLAG_PRECOMP_HALF)[mode]
ZK_STEP_CROSS_INVERSE, ZK_STEP_LOCAL, ZK_STEP_CROSS_FORWARD, ZK_STEP_FINISH = 1, 2, 3, 4
ZK_T_NAMES = ["spmv", "ntt_chain_wall", "sort_h", "msm_h_wall", "join_wait", "msm_reduce", "total_device", "g1_l1_kernel", "g2_l1_kernel", "wtns_h2d"]

# every symbol include/zkhip.h declares (tests check the library exports all of them)
EXPORTS = ["zk_last_error", "zk_device_count", "zk_prover_create", "zk_prover_destroy", "zk_prove", "zk_prove_dev",
           "zk_prove_dev_submit", "zk_prove_submit", "zk_prove_batch_submit", "zk_prove_batch_collect", "zk_host_alloc", "zk_host_free", "zk_prove_collect", "zk_prover_reserve", "zk_prover_info", "zk_prove_msm_collect", "zk_prove_msm_dev", "zk_prove_msm", "zk_prove_finish", "zk_prover_timings", "zk_fr_mul_vec",
           "zk_fq_mul_vec", "zk_fr_coef_accumulate", "zk_fr_ntt", "zk_fr_abc_to_h", "zk_msm_g1", "zk_msm_g2", "zk_proof_to_json",
           "zk_public_to_json", "zk_synth_chain_g1", "zk_synth_chain_g2", "zk_fixed_base_g1", "zk_fixed_base_g2", "zk_g1_mul", "zk_g2_mul", "zk_assemble",
           "zk_multi_prover_create", "zk_multi_prover_destroy", "zk_multi_prove", "zk_multi_prove_submit", "zk_multi_prove_collect",
           "zk_multi_prover_info", "zk_shard_info", "zk_shard_set_exchange", "zk_shard_begin", "zk_shard_step"]


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ZkHipError("libzkhip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "or `make -C rapidsnark-old_amd/csrc`; there is no CPU fallback" % path)
    # six streams per prover: the HIP runtime's default of 4 hardware queues aliases them and the
    # witness upload of proof k+1 then queues behind proof k (csrc/prover_create.hip); read at HIP init
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    try:
        # torch wheels bundle their own libamdhip64; load it FIRST so this process holds ONE HIP
        # runtime (two runtimes => the second one sees "No HIP GPUs are available").
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.zk_last_error.restype = C.c_char_p
    u8p = C.c_void_p
    lib.zk_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.zk_prover_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(zk_zkey_view), C.POINTER(zk_opts)]
    lib.zk_prover_destroy.argtypes = [C.c_void_p]
    lib.zk_prover_destroy.restype = None
    lib.zk_prove.argtypes = [C.c_void_p, u8p, u8p, u8p, C.POINTER(zk_proof)]
    lib.zk_prove_dev.argtypes = [C.c_void_p, C.c_void_p, u8p, u8p, C.POINTER(zk_proof)]
    lib.zk_prove_dev_submit.argtypes = [C.c_void_p, C.c_void_p, u8p, u8p]
    lib.zk_prove_submit.argtypes = [C.c_void_p, u8p, u8p, u8p]
    lib.zk_prove_batch_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, u8p, u8p]
    lib.zk_prove_batch_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.zk_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.zk_host_free.argtypes = [C.c_void_p]
    lib.zk_host_free.restype = None
    lib.zk_prove_collect.argtypes = [C.c_void_p, C.POINTER(zk_proof)]
    lib.zk_prove_msm_collect.argtypes = [C.c_void_p, C.POINTER(zk_msm_sums)]
    lib.zk_prove_msm.argtypes = [C.c_void_p, u8p, C.POINTER(zk_msm_sums)]
    lib.zk_prove_msm_dev.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(zk_msm_sums)]
    lib.zk_prove_finish.argtypes = [C.c_void_p, C.POINTER(zk_msm_sums), C.c_uint32, u8p, u8p, C.POINTER(zk_proof)]
    lib.zk_assemble.argtypes = [u8p, u8p, u8p, u8p, u8p, C.POINTER(zk_msm_sums), C.c_uint32, u8p, u8p, C.POINTER(zk_proof)]
    lib.zk_prover_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_uint32]
    if hasattr(lib, "zk_prover_reserve"):      # (ZKHIP_LIB may name an older build of the library: same-box A/B runs)
        lib.zk_prover_reserve.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    if hasattr(lib, "zk_prover_info"):
        lib.zk_prover_info.argtypes = [C.c_void_p, C.POINTER(zk_prover_plan)]
    lib.zk_multi_prover_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(zk_zkey_view), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(zk_opts)]
    lib.zk_multi_prover_destroy.argtypes = [C.c_void_p]
    lib.zk_multi_prover_destroy.restype = None
    lib.zk_multi_prove.argtypes = [C.c_void_p, u8p, u8p, u8p, C.POINTER(zk_proof)]
    lib.zk_multi_prove_submit.argtypes = [C.c_void_p, u8p, u8p, u8p]
    lib.zk_multi_prove_collect.argtypes = [C.c_void_p, C.POINTER(zk_proof)]
    lib.zk_multi_prover_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.zk_shard_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    lib.zk_shard_set_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.zk_shard_begin.argtypes = [C.c_void_p, u8p, C.c_void_p, u8p, u8p, C.c_void_p]
    lib.zk_shard_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    for name in ("zk_fr_mul_vec", "zk_fq_mul_vec"):
        getattr(lib, name).argtypes = [u8p, u8p, u8p, C.c_uint64]
    lib.zk_fr_coef_accumulate.argtypes = [u8p, u8p, u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint32]
    lib.zk_fr_ntt.argtypes = [u8p, C.c_uint64, C.c_int]
    lib.zk_fr_abc_to_h.argtypes = [u8p, u8p, u8p, C.c_uint64]
    lib.zk_msm_g1.argtypes = [u8p, u8p, u8p, C.c_uint64]
    lib.zk_msm_g2.argtypes = [u8p, u8p, u8p, C.c_uint64]
    lib.zk_synth_chain_g1.argtypes = [u8p, C.c_uint64, u8p, u8p]
    lib.zk_synth_chain_g2.argtypes = [u8p, C.c_uint64, u8p, u8p]
    lib.zk_fixed_base_g1.argtypes = [u8p, u8p, u8p, C.c_uint64]
    lib.zk_fixed_base_g2.argtypes = [u8p, u8p, u8p, C.c_uint64]
    lib.zk_g1_mul.argtypes = [u8p, u8p, u8p]
    lib.zk_g2_mul.argtypes = [u8p, u8p, u8p]
    lib.zk_proof_to_json.argtypes = [C.POINTER(zk_proof), C.c_char_p, C.c_size_t]
    lib.zk_proof_to_json.restype = C.c_size_t
    lib.zk_public_to_json.argtypes = [u8p, C.c_uint32, C.c_char_p, C.c_size_t]
    lib.zk_public_to_json.restype = C.c_size_t
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        raise ZkHipError(load_library().zk_last_error().decode("utf-8", "replace"))


def _buf(b, nbytes=None):
    """bytes / bytearray / numpy uint8 -> contiguous numpy uint8 array (kept alive by caller)."""
    a = np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b).view(np.uint8).reshape(-1)
    if nbytes is not None and a.size != nbytes:
        raise ValueError("expected %d bytes, got %d" % (nbytes, a.size))
    return a


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def device_count():
    n = C.c_int(0)
    check(load_library().zk_device_count(C.byref(n)))
    return n.value


class PinnedBuffer:
    """Page-locked host memory from zk_host_alloc, exposed as a numpy uint8 array (`.array`).  A witness
    placed here is uploaded by the DMA engine directly (zk_prove_submit does not stage it)."""

    def __init__(self, nbytes):
        self._ptr = C.c_void_p()
        check(load_library().zk_host_alloc(C.byref(self._ptr), nbytes))
        self.nbytes = nbytes
        self.array = np.ctypeslib.as_array(C.cast(self._ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def free(self):
        if self._ptr is not None and self._ptr.value:
            self.array = None
            load_library().zk_host_free(self._ptr)
            self._ptr = C.c_void_p()

    __del__ = free


def _mul_vec(fn, a, b):
    a, b = _buf(a), _buf(b)
    assert a.size == b.size and a.size % 32 == 0
    out = np.empty_like(a)
    check(fn(_ptr(out), _ptr(a), _ptr(b), a.size // 32))
    return out.tobytes()


def fr_mul_vec(a, b):
    return _mul_vec(load_library().zk_fr_mul_vec, a, b)


def fq_mul_vec(a, b):
    return _mul_vec(load_library().zk_fq_mul_vec, a, b)


def fr_coef_accumulate(coefs, n_coefs, domain_size, wtns):
    """(a, b) = (A.w, B.w) from the packed coefficient records (zkey section 4 image incl. its u32 count) —
    src/groth16.cpp:62-85; a, b as numpy uint8 arrays, Montgomery form."""
    coefs, wtns = _buf(coefs), _buf(wtns)
    a = np.empty(domain_size * 32, dtype=np.uint8)
    b = np.empty(domain_size * 32, dtype=np.uint8)
    check(load_library().zk_fr_coef_accumulate(_ptr(a), _ptr(b), _ptr(coefs), n_coefs, domain_size, _ptr(wtns), wtns.size // 32))
    return a, b


def fr_ntt(data, inverse=False):
    a = _buf(data).copy()
    check(load_library().zk_fr_ntt(_ptr(a), a.size // 32, 1 if inverse else 0))
    return a.tobytes()


def fr_abc_to_h(a, b):
    a, b = _buf(a), _buf(b)
    out = np.empty_like(a)
    check(load_library().zk_fr_abc_to_h(_ptr(out), _ptr(a), _ptr(b), a.size // 32))
    return out.tobytes()


def msm_g1(bases, scalars):
    bases, scalars = _buf(bases), _buf(scalars)
    n = scalars.size // 32
    assert bases.size == n * 64
    out = np.zeros(64, dtype=np.uint8)
    check(load_library().zk_msm_g1(_ptr(out), _ptr(bases), _ptr(scalars), n))
    return out.tobytes()


def msm_g2(bases, scalars):
    bases, scalars = _buf(bases), _buf(scalars)
    n = scalars.size // 32
    assert bases.size == n * 128
    out = np.zeros(128, dtype=np.uint8)
    check(load_library().zk_msm_g2(_ptr(out), _ptr(bases), _ptr(scalars), n))
    return out.tobytes()


def proof_to_json(proof_bytes):
    p = zk_proof.from_buffer_copy(bytes(proof_bytes))
    lib = load_library()
    n = lib.zk_proof_to_json(C.byref(p), None, 0)
    buf = C.create_string_buffer(n + 1)
    lib.zk_proof_to_json(C.byref(p), buf, n + 1)
    return buf.value.decode()


def public_to_json(wtns_values_bytes, n_public):
    a = _buf(wtns_values_bytes)
    lib = load_library()
    n = lib.zk_public_to_json(_ptr(a), n_public, None, 0)
    buf = C.create_string_buffer(n + 1)
    lib.zk_public_to_json(_ptr(a), n_public, buf, n + 1)
    return buf.value.decode()


def synth_chain_g1(n, p0, q):
    """out[i] = P0 + i*Q on the GPU -> numpy uint8 [n*64] (affine Montgomery)."""
    out = np.zeros(n * 64, dtype=np.uint8)
    a, b = _buf(p0).copy(), _buf(q).copy()
    check(load_library().zk_synth_chain_g1(_ptr(out), n, _ptr(a), _ptr(b)))
    return out


def synth_chain_g2(n, p0, q):
    out = np.zeros(n * 128, dtype=np.uint8)
    a, b = _buf(p0).copy(), _buf(q).copy()
    check(load_library().zk_synth_chain_g2(_ptr(out), n, _ptr(a), _ptr(b)))
    return out


def _scalars_le(scalars):
    if isinstance(scalars, np.ndarray) and scalars.dtype == np.uint8:
        return np.ascontiguousarray(scalars)
    return np.frombuffer(b"".join(int(k).to_bytes(32, "little") for k in scalars), dtype=np.uint8).copy()


def fixed_base_g1(base, scalars):
    """[k*base for k in scalars] on the GPU -> numpy uint8 [n*64] (affine Montgomery; 0 -> all-zero).
    scalars: iterable of ints < 2^256, or uint8 array n*32 (LE standard form)."""
    sc = _scalars_le(scalars)
    n = sc.size // 32
    out = np.zeros(n * 64, dtype=np.uint8)
    b = _buf(base).copy()
    check(load_library().zk_fixed_base_g1(_ptr(out), _ptr(b), _ptr(sc), n))
    return out


def fixed_base_g2(base, scalars):
    sc = _scalars_le(scalars)
    n = sc.size // 32
    out = np.zeros(n * 128, dtype=np.uint8)
    b = _buf(base).copy()
    check(load_library().zk_fixed_base_g2(_ptr(out), _ptr(b), _ptr(sc), n))
    return out


def g1_mul(p, k):
    """k*P on the host (product code, 64-bit limbs): Curve::mulByScalar."""
    out = np.zeros(64, dtype=np.uint8)
    a, kk = _buf(p).copy(), np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).copy()
    check(load_library().zk_g1_mul(_ptr(out), _ptr(a), _ptr(kk)))
    return out.tobytes()


def g2_mul(p, k):
    out = np.zeros(128, dtype=np.uint8)
    a, kk = _buf(p).copy(), np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).copy()
    check(load_library().zk_g2_mul(_ptr(out), _ptr(a), _ptr(kk)))
    return out.tobytes()


def assemble(vk, partials, r=None, s=None):
    """Host-only final assembly (src/groth16.cpp:209-253) over the partial MSM sums of all shards.
    vk: dict with vk_alpha1, vk_beta1, vk_beta2, vk_delta1, vk_delta2 (bytes)."""
    keep = [_buf(vk[k]).copy() for k in ("vk_alpha1", "vk_beta1", "vk_beta2", "vk_delta1", "vk_delta2")]
    arr = (zk_msm_sums * len(partials))(*[zk_msm_sums.from_buffer_copy(bytes(p)) for p in partials])
    ra = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8).copy() if r is not None else None
    sa = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8).copy() if s is not None else None
    out = zk_proof()
    check(load_library().zk_assemble(*[_ptr(a) for a in keep], arr, len(partials),
                                     _ptr(ra) if ra is not None else None, _ptr(sa) if sa is not None else None, C.byref(out)))
    return bytes(out)
