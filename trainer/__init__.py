"""`python -m trainer.estimator` — the reference's CLI name (reference README.md:92).

The implementation lives in `glove-tensorflow_amd/trainer/` (a directory name Python cannot
import directly because of the hyphen); this shim only puts it on the package path.
"""
import os as _os
from pathlib import Path as _Path

# multi-process GPU work on this platform needs dmabuf IPC (RCCL / tensor sharing fail with the legacy mode)
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__path__.append(str(_Path(__file__).resolve().parent.parent / "glove-tensorflow_amd" / "trainer"))
