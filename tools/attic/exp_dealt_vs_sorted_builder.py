#!/usr/bin/env python3
"""The same dealt batches indexed two ways — glove_plan_build_sorted (as the runner does) and the sorting builder on the
batch's pairs — then stepped: is a step on the runner's plans slower than on the sorting builder's, and why?
Usage: tools/exp_dealt_vs_sorted_builder.py [workload] [B]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.data_utils import NonzeroStream  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, PlanBlock, make_hyper  # noqa: E402
from trainer.stepper import HipBackend  # noqa: E402

wl_name, B = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("zipf_v400k_d300", 1048576)
dev = torch.device("cuda:0")
hip = GloveHip(dev)
wl = synthetic.make_workload(wl_name, device=dev, work_device=dev)
V, d, nb, cap = wl["V"], wl["d"], 6, 32
backend = HipBackend(dev)
backend.hip = hip
stream = NonzeroStream({k: wl[k] for k in ("row", "col", "w", "y")}, B, V, backend, dev, seed=0, static_plans=False)
stream.reshuffle_in_place()
torch.cuda.synchronize()
rs, cs = stream.epoch_sides()
kinds = {}
for records, words in ((True, False), (False, False), (False, True)):
    blk = PlanBlock([hip.staging_plan(B, V, cap, dev, records=records, run_words=words) for _ in range(nb)])
    ws = torch.empty(max(hip.lib.glove_plan_sorted_workspace_bytes(B, nb), 256), dtype=torch.uint8, device=dev)
    hip.build_plans_sorted(rs, cs, 0, blk, nb, V, ws)
    blk.fetch_counts()
    torch.cuda.synchronize()
    blk.adopt_counts(nb)
    kinds["build_sorted, %s" % ("records only" if records else "own pair arrays + run words" if words else "own pair arrays, no records")] = blk.plans
    kinds["_keep%d%d" % (records, words)] = [blk, ws]
batches = [tuple(t.contiguous() for t in stream.batch(b)) for b in range(nb)]
kinds["sorting builder, compact"] = [hip.build_plan(*bt, V, chunk_cap=cap, compact=True, d=d) for bt in batches]
kinds["sorting builder, staging + records"] = [hip.build_plan(*bt, V, chunk_cap=cap, records=True) for bt in batches]
slices = [tuple(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")) for b in range(nb)]
kinds["static slices of the stream, compact"] = [hip.build_plan(*bt, V, chunk_cap=cap, compact=True, d=d) for bt in slices]
for k, ps in kinds.items():
    if not k.startswith("_"):
        print("%-46s counts %s cap_chunks %d" % (k, ps[0].counts[:5].tolist(), ps[0].cap_chunks))
loss = torch.zeros(4, device=dev)
wsz = max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, (d + 3) // 4 * 4) for k, ps in kinds.items() if not k.startswith("_") for p in ps)
ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
tables.maybe_enable_twin()
hyper = make_hyper(learning_rate=0.05, batch_size=B)
for rnd in range(3):
    for name, ps in kinds.items():
        if name.startswith("_"):
            continue
        for i in range(3):
            hip.step_adagrad(ps[i % nb], tables, hyper, loss, ws)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(18):
            hip.step_adagrad(ps[i % nb], tables, hyper, loss, ws)
        b.record()
        torch.cuda.synchronize()
        if rnd:
            print("%-46s %.1f us/step" % (name, a.elapsed_time(b) * 1e3 / 18), flush=True)
