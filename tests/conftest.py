import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()  # builds on first use (gcc)
    return oracle


@pytest.fixture(scope="session")
def codec():
    """The product module; GPU tests call through its C ABI."""
    import alice_codec_amd
    alice_codec_amd.load_library()
    return alice_codec_amd


@pytest.fixture(scope="session")
def gpu_codec(codec):
    if codec.device_count() < 1:
        pytest.fail("a -m gpu test ran without a HIP device: the product path has no CPU fallback")
    codec.set_device(0)
    return codec


def golden_cases():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
