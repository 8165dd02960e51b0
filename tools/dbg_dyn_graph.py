#!/usr/bin/env python3
"""Diagnostic (not a test): the index build captured in a hipGraph and replayed — do the plan's counts come out the same
on every replay (is the zeroing of `counts` part of the graph)?  No step kernel runs: nothing consumes the plan."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
import torch
from helpers import make_batch, to_dev
from trainer.hip_api import GloveHip, Plan

hip = GloveHip("cuda:0")
B, V, cap = int(sys.argv[1]) if len(sys.argv) > 1 else 131072, 10000, 16
batches = [to_dev(*make_batch(900 + k, B, V)) for k in range(3)]
staging = Plan(B, V, cap, "cuda:0")
ws = torch.empty(hip.lib.glove_plan_workspace_bytes(B, V), dtype=torch.uint8, device="cuda:0")
seen = torch.zeros(3, 8, dtype=torch.int32, device="cuda:0")


def burst():
    for k, bt in enumerate(batches):
        hip.build_plan(*bt, V, chunk_cap=cap, into=staging, ws=ws)
        seen[k].copy_(staging.counts)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    burst()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("eager      ", seen.tolist(), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    burst()
torch.cuda.synchronize()
print("after capture (not run)", staging.counts.tolist(), flush=True)
for r in range(3):
    if r == 2:
        staging.counts.fill_(-1)
        staging.heavy.fill_(-1)
    g.replay()
    torch.cuda.synchronize()
    print("replay %d   " % r, seen.tolist(), flush=True)
