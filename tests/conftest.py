import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU (or without the built HIP library) skips the gpu-marked tests instead of
    erroring inside their fixtures; on a GPU box a missing library still fails loudly (chipmunk_amd has no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import oracle
    oracle.build()
    yield


@pytest.fixture()
def fresh_config():
    """Restore GLOBAL_CONFIG and the shared LayerCounter around a test."""
    from chipmunk_amd.util import config as cfg
    from chipmunk_amd.util import layer_counter as lc
    cfg.reset_to_base()
    lc.singleton.__init__(0, 0)
    yield cfg.GLOBAL_CONFIG
    cfg.reset_to_base()
    lc.singleton.__init__(0, 0)
