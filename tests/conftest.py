import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The test process loads the LAB build of the library: the same objects as libflashjoin_hip.so, linked without the export list, so
# that the building blocks behind the C ABI (include/flashjoin_lab.h: stream joins, the pieces of the multi-GPU driver, the
# partition diagnostic) can be tested one by one.  The product library itself is exercised by the tests that start other processes
# with product_env() - the headline bench line, the twelve functions against the golden vectors, the C host example.
os.environ.setdefault("FJ_LIB_VARIANT", "lab")


def product_env(**extra):
    """Environment of a child process that must load the PRODUCT library."""
    env = {k: v for k, v in os.environ.items() if k != "FJ_LIB_VARIANT"}
    env.update(extra)
    return env


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
