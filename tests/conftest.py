import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _emulated():
    """PK_EMU=1: run the -m gpu tests against the host emulation of the kernels (tools/hipemu; development aid for
    kernel logic when no GPU box is at hand -- never a parity claim, never set by the driver)."""
    return os.environ.get("PK_EMU") == "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if _emulated():
        sys.path.insert(0, os.path.join(ROOT, "tools", "hipemu"))
        import harness
        harness.install()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu() or _emulated():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
