#!/usr/bin/env python3
"""Time of one index build into a staging plan (device-side, back to back): tools/exp_build_time.py WORKLOAD B [records] [lib]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.hip_api import GloveHip, Plan  # noqa: E402

wl_name, B = sys.argv[1], int(sys.argv[2])
records = len(sys.argv) > 3 and sys.argv[3] == "records"
lib = sys.argv[4] if len(sys.argv) > 4 else None
dev = torch.device("cuda:0")
hip = GloveHip(dev, lib_path=lib) if lib else GloveHip(dev)
wl = synthetic.make_workload(wl_name, device=dev, work_device=dev)
V, nb, cap = wl["V"], 4, 32
batches = [tuple(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")) for b in range(nb)]
staging = Plan(B, V, cap, dev, records=records or None)
ws = torch.empty(hip.lib.glove_plan_workspace_bytes(B, V), dtype=torch.uint8, device=dev)
for rnd in range(4):
    for i in range(3):
        hip.build_plan(*batches[i % nb], V, chunk_cap=cap, into=staging, ws=ws)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(20):
        hip.build_plan(*batches[i % nb], V, chunk_cap=cap, into=staging, ws=ws)
    b.record()
    torch.cuda.synchronize()
    if rnd:
        print("%s B=%d records=%s lib=%s: %.1f us per build" % (wl_name, B, records, lib or "shipped", a.elapsed_time(b) * 1e3 / 20), flush=True)
