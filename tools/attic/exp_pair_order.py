#!/usr/bin/env python3
"""Does the order of a row's pairs inside its chunks matter to the step?  The sorting builder keeps a row's pairs in arrival
order (random), the dealt epochs deliver them sorted by partner id (hot Zipf ids first in every chunk).  Same batches, same
plans otherwise; twin form; us per step.
Usage: tools/exp_pair_order.py [workload] [B]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, make_hyper  # noqa: E402

wl_name, B = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("zipf_v400k_d300", 1048576)
dev = torch.device("cuda:0")
hip = GloveHip(dev)
wl = synthetic.make_workload(wl_name, device=dev, work_device=dev)
V, d, nb, cap = wl["V"], wl["d"], 6, 32
batches = [tuple(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")) for b in range(nb)]


def ordered(bt, key):
    row, col, w, y = bt
    order = torch.argsort(key(row.long(), col.long()), stable=True)
    return tuple(t[order].contiguous() for t in bt)


def mixed(r, c):
    # partner order scrambled by a per-row rotation of a hash of the partner id: still one deterministic order per row
    h = (c * 2654435761 + r * 40503) & 0xFFFFF
    return r * (1 << 20) + h


variants = {
    "arrival order (sorting builder)": batches,
    "sorted by (row, col) (dealt epochs)": [ordered(bt, lambda r, c: r * V + c) for bt in batches],
    "sorted by (row, col descending)": [ordered(bt, lambda r, c: r * V + (V - 1 - c)) for bt in batches],
    "sorted by (row, hash of col)": [ordered(bt, mixed) for bt in batches],
}
# both sides: the col side's partner order follows from the arrival order too (stable sort by col): sorted by (row, col) input
# gives col chunks whose partners (rows) ascend as well
plans = {k: [hip.build_plan(*bt, V, chunk_cap=cap, compact=True, d=d) for bt in v] for k, v in variants.items()}
loss = torch.zeros(4, device=dev)
ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, B, (d + 3) // 4 * 4), dtype=torch.uint8, device=dev)
tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
tables.maybe_enable_twin()
hyper = make_hyper(learning_rate=0.05, batch_size=B)
for rnd in range(3):
    for name, ps in plans.items():
        for i in range(3):
            hip.step_adagrad(ps[i % nb], tables, hyper, loss, ws)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(18):
            hip.step_adagrad(ps[i % nb], tables, hyper, loss, ws)
        b.record()
        torch.cuda.synchronize()
        if rnd:
            print("%-40s %.1f us/step" % (name, a.elapsed_time(b) * 1e3 / 18), flush=True)
