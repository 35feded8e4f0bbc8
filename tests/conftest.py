import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Kernel-level tests and the multi-step trajectory tests run on the order-preserving kernels (see the fixtures below and
# DESIGN 4.1); the model-level parity files run every test TWICE -- once like that and once on the DEFAULT kernel selection
# (split-K / fine-tail forms offered to the autotuner: what bench.py and the trainer launch), fixture `kernel_selection`.
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


@pytest.fixture(params=['ordered-kernels', 'default-kernels'])
def kernel_selection(request):
    """Model-level parity on both kernel selections: 'ordered-kernels' = LOANS_SPLITK=0 (bit-reproducible forward),
    'default-kernels' = what a run without that variable launches (split-K and fine-tail tiles on offer; their atomics change
    the last bits from run to run, the parity tolerances hold all the same).  Tests that pin the ordered kernels themselves
    (fixture deterministic_forward: multi-step trajectories at a handful of samples) run once."""
    from loans_amd import ops
    if request.param == 'default-kernels' and 'deterministic_forward' in request.fixturenames:
        pytest.skip('this test pins the order-preserving kernels')
    old, ops.SPLITK = ops.SPLITK, request.param == 'default-kernels'
    yield request.param
    ops.SPLITK = old
