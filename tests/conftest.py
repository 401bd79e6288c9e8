import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _oracle_on_a_few_threads():
    """The oracle instances the tests solve are small (at most a few MB of matrix): on a many-core host every OpenMP
    region costs more than the loop it splits, so cap the oracle at 16 threads for the test session (bench.py's
    cpu_baseline leg, which times the oracle on GB-sized samples, is not affected)."""
    try:
        import oracle as O
        if O.num_threads() > 16:
            O.set_num_threads(16)
    except Exception:
        pass
    yield
