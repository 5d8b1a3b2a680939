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
    """One srx context on cuda:0.  Fails loudly (no skip, no fallback) when libsrx cannot run."""
    import torch
    torch.cuda.init()     # before libsrx touches the device: torch's HIP runtime does not come up after another one
    from speechrecognition_amd import host
    c = host.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
