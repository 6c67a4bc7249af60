import json
import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # read when HIP initialises (rapidsnark-old_amd/csrc/prover_create.hip)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        import rapidsnark_old_amd as zk
        return zk.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: the gpu-marked tests are skipped, not failed (the product
    has no CPU fallback to fall back on)."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items or _gpu_present():
        return
    skip = pytest.mark.skip(reason="needs a HIP device (libzkhip has no CPU fallback)")
    for it in gpu_items:
        it.add_marker(skip)


def golden_path(*parts):
    return os.path.join(GOLDEN, *parts)


def golden_bytes(*parts):
    with open(golden_path(*parts), "rb") as f:
        return f.read()


def golden_json(*parts):
    with open(golden_path(*parts)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def zk():
    """The product package; GPU tests call through its C-ABI binding only."""
    import rapidsnark_old_amd
    return rapidsnark_old_amd


CIRCUITS = ["multiplier2", "r1cs_n8", "r1cs_n64", "r1cs_nopub", "r1cs_n256"]
