#!/usr/bin/env python3
"""Times PREDICT (cosine + top-k over all V row embeddings) per 256-query tile."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer.hip_api import GloveHip  # noqa: E402

hip = GloveHip("cuda:0")
for V, d in ((10000, 64), (50000, 300), (400000, 300)):
    R = torch.randn(V, d, device="cuda:0")
    q = torch.arange(256, dtype=torch.int32, device="cuda:0")
    hip.topk_cosine(R, q, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        hip.topk_cosine(R, q, 20)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("V=%d d=%d: %.2f ms per 256 queries -> whole vocabulary in %.1f s" % (V, d, dt * 1e3, dt * V / 256))
