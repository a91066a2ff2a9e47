import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's own DataLoader pin-memory thread warns once per batch about an argument torch itself passes (11,000 lines per GPU run)
    config.addinivalue_line("filterwarnings", "ignore:The argument 'device' of Tensor.(pin_memory|is_pinned):DeprecationWarning")


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


@pytest.fixture(autouse=True, scope="module")
def _give_cached_hbm_back_after_each_module():
    """The benchmark-shape tests leave > 200 GB in this process's caching allocator; tests that start OTHER processes on the same GPU
    (tests/test_dist_gpu.py: the plain `bench.py --gpus 2` command) need it back."""
    yield
    if "torch" in sys.modules:
        import gc
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            from emdr2_amd.model import kernels as K
            K.GRAD_SINK = None
            K.ATTN_STASH.store.clear()
            gc.collect()
            torch.cuda.empty_cache()
