import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HOSTSIM = os.path.join(ROOT, "tests", "hostsim", "libhymls_mi_hostsim.so")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hostsim_lib():
    """TEST-ONLY host simulator of the device plan (tests/hostsim); never the product path."""
    if not os.path.exists(HOSTSIM):
        import subprocess
        subprocess.check_call(["make", "-j8", "-C", os.path.dirname(HOSTSIM)])
    import hymls_amd
    return hymls_amd.load_library(HOSTSIM)


@pytest.fixture(scope="session")
def gpu_lib():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import hymls_amd
    return hymls_amd.load_library()  # the HIP library; ImportError if missing (no fallback)
