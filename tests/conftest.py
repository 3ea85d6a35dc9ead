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
    """GPU tests must never silently pass on a box without a GPU: skip them there
    unless they were asked for explicitly (-m gpu), in which case they fail loudly."""
    import torch

    if torch.cuda.is_available():
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return  # explicit request: let them run and fail
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True, scope="module")
def _settle_gpu_state_between_modules():
    """Every test file of the GPU suite starts from a settled process: objects of the previous file that own GPU runtime
    state (captured HIP graphs and their private pools, process groups, DDP reducers) are destroyed HERE, at a known
    point, instead of whenever the cyclic collector next runs inside another file's kernels; the device is idle and the
    caching allocator's unused blocks are back with the driver (vendor libraries allocate outside it)."""
    yield
    import gc

    import torch
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
