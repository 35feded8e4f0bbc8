import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Model-level parity tests run on the order-preserving kernels (see the fixture below and DESIGN 4.1): split-K is opted
# into by the tests that are about it (explicit tile ids at kernel level, test_split_k_autotuned_step at model level).
os.environ.setdefault('LOANS_SPLITK', '0')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, so a bare
    ``pytest tests`` works on the CPU container too."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


import pytest


@pytest.fixture
def deterministic_forward():
    """Multi-step trajectory comparisons: split-K (fp32 atomics in the small-batch forward / data gradient) makes the last
    bits of a step depend on the run, and at a handful of samples per batch one flipped ReLU / pooling decision moves a
    later loss by ~1e-3.  These tests pin the order-independent kernels; split-K itself is covered at kernel level."""
    from loans_amd import ops
    old, ops.SPLITK = ops.SPLITK, False
    yield
    ops.SPLITK = old
