#!/usr/bin/env python3
"""Times glove_cooccurrence_i32 on a text8-sized synthetic corpus (17 M Zipf tokens, window 5)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer.hip_api import GloveHip  # noqa: E402

n, V, ctx = 17_005_207, 10_000, 5
dev = torch.device("cuda:0")
hip = GloveHip(dev)
g = torch.Generator(device=dev)
g.manual_seed(0)
cdf = torch.arange(1, V + 1, dtype=torch.float64, device=dev).pow(-1.0).cumsum(0)
tok = torch.searchsorted(cdf / cdf[-1], torch.rand(n, dtype=torch.float64, device=dev, generator=g)).clamp_(max=V - 1).int()
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    row, col, cnt, val = hip.cooccurrence(tok, V, ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("run %d: %.1f ms, %.3g tokens/s, %d distinct pairs, %d with count>=10" % (
        it, dt * 1e3, n / dt, row.numel(), int((cnt >= 10).sum())))
