import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
