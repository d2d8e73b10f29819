import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def three_products(monkeypatch):
    """The backward convolutions with THREE MFMA products per MAC (fp32-class gradients, 2e-7 per op) for the duration of a test:
    tests that pin that accuracy class -- per-op comparisons at 1e-6-class tolerances, fp64 accuracy budgets, A/B equalities of two
    launch forms -- ask for it; the default (hipops.BWD_PRODUCTS = 2) has its own tests (test_hip_ops.py::test_backward_two_products)."""
    from egaze_amd import hipops as H
    monkeypatch.setattr(H, "BWD_PRODUCTS", 3)
