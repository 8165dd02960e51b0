"""`python -m trainer.estimator` — the reference's CLI name (reference README.md:92).

The implementation lives in `glove-tensorflow_amd/trainer/` (a directory name Python cannot
import directly because of the hyphen); this shim only puts it on the package path.
"""
from pathlib import Path as _Path

__path__.append(str(_Path(__file__).resolve().parent.parent / "glove-tensorflow_amd" / "trainer"))
