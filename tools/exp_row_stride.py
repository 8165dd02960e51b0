#!/usr/bin/env python3
"""Timing experiment: does the C4 step gain from table rows that start on 64- / 128-byte lines?  The same batches (ids, weights)
stepped on tables of d = 300 (1,200-byte rows: 9.4 lines of 128 B, any 16-byte phase), 304 (1,216 B = 19 x 64), 320 (1,280 B = 10 x 128).
    python tools/exp_row_stride.py [workload] [B]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from trainer import synthetic
from trainer.hip_api import DeviceTables, GloveHip, make_hyper

dev = torch.device("cuda:0")
hip = GloveHip(dev)
name = sys.argv[1] if len(sys.argv) > 1 else "zipf_v400k_d300"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
wl = synthetic.make_workload(name, device=dev, work_device=dev)
V = wl["V"]
nb = 8
for rnd in range(2):
    for d in (wl["d"], wl["d"] + 4, (wl["d"] + 31) // 32 * 32):
        tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
        tables.maybe_enable_twin()
        plans = [hip.build_plan(*(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")), V, chunk_cap=32, compact=True, d=tables.d)
                 for b in range(nb)]
        hyper = make_hyper(learning_rate=0.05, batch_size=B)
        ws = torch.empty(max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, tables.d) for p in plans), dtype=torch.uint8, device=dev)
        loss = torch.zeros(4, device=dev)
        for i in range(8):
            hip.step_adagrad(plans[i % nb], tables, hyper, loss, ws)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(40):
            hip.step_adagrad(plans[i % nb], tables, hyper, loss, ws)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 40
        ids = sum(p.host_counts[1] + p.host_counts[3] for p in plans) / nb
        print("%s B=%d d=%d (row %d B): %.1f us per step; %.0f ids: %.3f GB of rows read + written (4 per id)" % (
            name, B, d, 4 * tables.d, us, ids, ids * 16 * tables.d / 1e9), flush=True)
        del tables, plans, ws
        torch.cuda.empty_cache()
