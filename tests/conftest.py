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
def two_products(monkeypatch):
    """The OPT-IN backward arithmetic (EGAZE_BWD_PRODUCTS=2: two MFMA products per MAC, one operand of every backward product with
    11 significant bits) for the duration of a test.  The default is three products (fp32-class gradients, 2e-7 per op): every test
    without this fixture -- per-op comparisons at 1e-6-class tolerances, fp64 accuracy budgets, A/B equalities, the whole-model
    comparisons against oracle and goldens -- guards the arithmetic that ships and that bench.py times."""
    from egaze_amd import hipops as H
    monkeypatch.setattr(H, "BWD_PRODUCTS", 2)
