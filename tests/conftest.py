import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle / restatements run on the host: on a 256-core GPU box PyTorch's small CPU ops crawl (minutes per
    # call) when every core joins each parallel region
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


def install_emulated_ops():
    """Replace every function of deepsvg_amd.ops by its plain-torch restatement (CPU host-logic tests only)."""
    import deepsvg_amd.ops as ops
    from tests import torch_ops_ref as ref
    saved = {}
    for name in dir(ref):
        if name.startswith("_") or not callable(getattr(ref, name)) or not hasattr(ops, name):
            continue
        saved[name] = getattr(ops, name)
        setattr(ops, name, getattr(ref, name))
    return saved


def restore_ops(saved):
    import deepsvg_amd.ops as ops
    for name, fn in saved.items():
        setattr(ops, name, fn)


@pytest.fixture
def emulated_ops():
    saved = install_emulated_ops()
    yield
    restore_ops(saved)


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
