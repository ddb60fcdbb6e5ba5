import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def pg():
    """The product package with its HIP library loaded (fails loudly if it is not built)."""
    import pagan2_msa_amd as p
    p.lib()
    return p
