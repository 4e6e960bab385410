import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


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
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "generator_fp32.npz"))


@pytest.fixture(scope="session")
def golden_default():
    return np.load(os.path.join(GOLDEN, "generator_default_init.npz"))


@pytest.fixture(scope="session")
def golden_metrics():
    return np.load(os.path.join(GOLDEN, "metrics.npz"))


@pytest.fixture(scope="session")
def synth_sd_np():
    from superresolution_amd import weights as W
    return W.synthetic_state_dict_numpy(0)


@pytest.fixture(scope="session")
def golden_inputs():
    from superresolution_amd import weights as W
    return np.concatenate([W.synthetic_lr_batch(1, 0, "dense"), W.synthetic_lr_batch(1, 0, "sparse")], 0)


def max_rel(a, ref):
    """max|a-ref| / max|ref| -- the fp32 criterion of BASELINE.json's north_star (outputs cross zero,
    so a per-element relative error is ill-posed)."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))
