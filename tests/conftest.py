import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    # some -m gpu tests hand torch CUDA tensors to the library (ensemble exchange): PyTorch's bundled HIP runtime has to be
    # initialised before the library's; Context() does that when torch is already imported
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    from sclens_amd._lib import Context

    c = Context(0)
    yield c
    c.close()
