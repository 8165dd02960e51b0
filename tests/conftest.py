import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The product path: libglove_hip.so through the C ABI.  No fallback: a missing library on
    a GPU box is an error, not a skip."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this container")
    from trainer import hip_api
    if not hip_api.LIB_PATH.exists():
        # a GPU box that received the sources without the built library: compile it here (hipcc is part of the image);
        # a failing build is an error like a missing library is
        import __graft_entry__
        __graft_entry__.build()
    return hip_api.GloveHip("cuda:0")
