import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """The shared objects normally travel with the tree (__graft_entry__.build()); if a checkout lacks them,
    build them once here -- the product itself never builds or falls back on its own."""
    import subprocess
    lib = os.path.join(ROOT, "xgnn_amd", "lib", "libggms_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "xgnn_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libggms_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session", autouse=True)
def scan_patience_of_this_run():
    """GGMS_TEST_SCAN_PATIENCE=0 pytest -m gpu: the whole GPU suite with look-backs that never wait -- every ordered scan
    that finds a predecessor's word missing computes it itself (include/ggms.h, ggms_debug_set_scan_patience).  Results
    must not change.  A test aid of the test session, not of the library: the product reads no environment for this."""
    v = os.environ.get("GGMS_TEST_SCAN_PATIENCE")
    if v is not None:
        from xgnn_amd import lib
        lib().ggms_debug_set_scan_patience(int(v))
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
