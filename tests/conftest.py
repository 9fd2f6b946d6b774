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
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def rel_l2(a, b):
    import torch
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


@pytest.fixture
def kernel_env(monkeypatch):
    """Set kernel-selection overrides (PA2D_CONV_HALO, PA2D_MC_BIG, ...) for one test: libpa2d reads them once at load, so
    the table is re-read after every change and once more when the test's environment has been restored."""
    from transformerbasednavierstokesolver_amd import _lib

    def set_(**kv):
        for k, v in kv.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        _lib.load().pa2d_reload_env()

    yield set_
    monkeypatch.undo()
    _lib.load().pa2d_reload_env()
