import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a box without a GPU: skip instead of failing in whatever the test touches first (artgpu_create, torch.cuda, the CLI)."""
    if not any("gpu" in it.keywords for it in items):
        return
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if not have:
        skip = pytest.mark.skip(reason="no GPU")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from art_amd import capi

    ctx = capi.Context(0)
    yield ctx
    ctx.close()
