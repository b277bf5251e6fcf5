import os
import sys

import pytest

# torch brings its own copy of the HIP runtime; it has to be the one the process loads first (the
# product library then binds to it by SONAME), or a later torch.cuda call finds "no HIP GPUs"
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger CPU-only case")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(HERE, "golden")
