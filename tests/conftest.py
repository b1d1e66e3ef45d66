import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box via gpurun)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests SKIP (instead of failing their first assert) on a host without a ROCm device or without
    the built library.  `-m gpu` on the GPU box is unaffected: there both are present, and a missing library
    still fails loudly (the product path raises NphmAmdError)."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a ROCm GPU (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
