import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import _oracle
    return _oracle.oracle()


@pytest.fixture(scope="session")
def ref():
    import _oracle
    r = _oracle.ref()
    if r is None:
        pytest.skip("oracle/_ref/libyams_ref.so not present (built only where /root/reference exists)")
    return r


@pytest.fixture(scope="session")
def accel_lib():
    """The product library, built in-tree; loaded WITHOUT torch for the CPU-side ABI tests."""
    from yams_amd import build as _b
    _b.build()
    from yams_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def acc():
    """A live accelerator context on cuda:0 sharing torch's HIP runtime and current stream."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from yams_amd.accel import Accel
    a = Accel(0, torch.cuda.current_stream().cuda_stream)
    yield a
    a.close()
