import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import _abi
    lib = _abi.load_oracle()
    if lib is None:
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        lib = _abi.load_oracle()
    return lib


@pytest.fixture(scope="session")
def ref_strict():
    """The real reference C++ (strict-FP build), present when oracle/_ref was built in the dev container."""
    import _abi
    path = os.path.join(ROOT, "oracle", "_ref", "libvag_ref_strict.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return _abi.CpuLib(path, "vag_ref")


@pytest.fixture(scope="session")
def ref_fast():
    import _abi
    lib = _abi.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return lib
