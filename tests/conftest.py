import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def rel_err(a, b):
    """max |a-b| / max |b| — the tolerance form used throughout (b is the reference)."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    denom = b.abs().max().clamp_min(1e-30)
    return ((a - b).abs().max() / denom).item()


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, "{}: rel err {:.3e} > {:.1e}".format(what, e, tol)


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2"""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def assert_grad_close(a, b, what="", l2_tol=5e-4, max_tol=5e-3):
    """Gradient comparison that tolerates activation kinks.

    PReLU / ReLU make the gradient discontinuous: a pre-activation within fp32 rounding of zero takes
    the other slope in one of two correct fp32 implementations (the reference's own fp32 gradients differ
    from a float64 run in exactly this way), which perturbs a handful of elements by O(1e-3) while the
    rest agree to 1e-6.  Measured on the small end-to-end case: reference fp32 vs float64 up to 1.2e-4 in
    max-norm, this build vs float64 up to 3.5e-4, traced to one flipped PReLU input out of 5e5.  So on
    networks with kinks gradients are held to 5e-4 in relative L2 and 5e-3 in max-norm; the strict 1e-4
    max-norm bar is applied where no kink can flip (single layers at small T, and end-to-end with all
    slopes set to 1: test_smooth_network_gradients_vs_float64)."""
    e2, em = rel_l2(a, b), rel_err(a, b)
    assert e2 <= l2_tol, "{}: rel L2 err {:.3e} > {:.1e}".format(what, e2, l2_tol)
    assert em <= max_tol, "{}: rel max err {:.3e} > {:.1e}".format(what, em, max_tol)
