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


@pytest.fixture(scope="session")
def plan_checker(hip):
    """tests/native/libglove_test_checks.so: device-side range checks of a plan, launched on the stream between the index
    build and the step (also inside a captured hipGraph).  check(plan, V, errors8) adds to the int32[8] device array."""
    import ctypes as C
    import subprocess
    import torch
    from trainer.hip_api import GlovePlan
    so = REPO / "tests" / "native" / "libglove_test_checks.so"
    if not so.exists():
        subprocess.run(["make", "-s", "-C", str(so.parent)], check=True)
    lib = C.CDLL(str(so))
    lib.glove_test_check_plan.restype = C.c_int
    lib.glove_test_check_plan.argtypes = [C.POINTER(GlovePlan), C.c_int32, C.c_void_p, C.c_void_p]

    def check(plan, V, errors):
        assert errors.dtype == torch.int32 and errors.numel() == 8 and errors.is_cuda
        rc = lib.glove_test_check_plan(C.byref(plan.struct()), V, errors.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    return check
