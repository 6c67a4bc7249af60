"""rapidsnark-old_amd — MI355X-native Groth16 hot path (drop-in for rapidsnark's prove()).

Python side = thin ctypes binding over the C-ABI in include/zkhip.h (libzkhip.so) plus a
mirror of the reference's binfile/zkey/wtns readers.  There is no Python/CPU compute
fallback: if libzkhip.so is missing or HIP has no device, calls raise.
"""
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # before the first HIP call of the process (lib.py: load_library)
from .binfile import BinFile, open_existing            # noqa: F401
from .zkey import ZkeyHeader, load_zkey_header          # noqa: F401
from .wtns import WtnsHeader, load_wtns_header          # noqa: F401
from .lib import (ZkHipError, load_library, library_path, fr_mul_vec, fq_mul_vec, fr_ntt, fr_coef_accumulate,   # noqa: F401
                  fr_abc_to_h, msm_g1, msm_g2, proof_to_json, public_to_json, device_count,
                  synth_chain_g1, synth_chain_g2, fixed_base_g1, fixed_base_g2, g1_mul, g2_mul, assemble, PinnedBuffer)
from .prover import Prover, MultiProver, prove_files                 # noqa: F401
from . import synth                                     # noqa: F401
from .dist import gather_partials, ShardedChain                        # noqa: F401
