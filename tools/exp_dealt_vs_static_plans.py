#!/usr/bin/env python3
"""The same batches of one dealt epoch stepped from the runner's staging plans (glove_plan_build_sorted: capacity-sized
arrays, run words, pair fields borrowed from the epoch's arrays, counts adopted from the device) and from resident plans
(glove_plan_build + compact) — one process, same tables, interleaved rounds, one C call per run of steps either way.
Isolates what the plan's FORMAT costs the step (bench.py: C5 step kernels alone 370 us dealt against 343 static).

Usage: python tools/exp_dealt_vs_static_plans.py [--workload zipf_v2m_d128] [--batch-size 1048576]"""
import argparse
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.data_utils import NonzeroStream  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, make_hyper  # noqa: E402
from trainer.stepper import HipBackend, ReshufflingRunner  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="zipf_v2m_d128")
    ap.add_argument("--batch-size", type=int, default=1048576)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--batches", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hip = GloveHip(dev)
    wl = synthetic.make_workload(args.workload, device=dev, work_device=dev)
    V, d, B = wl["V"], wl["d"], args.batch_size
    backend = HipBackend(dev)
    backend.hip = hip
    tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
    backend.row_floats = tables.d
    stream = NonzeroStream({k: wl[k] for k in ("row", "col", "w", "y")}, B, V, backend, dev, seed=0, static_plans=False)
    hyper = make_hyper(batch_size=B, learning_rate=0.05)
    runner = ReshufflingRunner(hip, stream, tables, hyper, burst=64)
    n0 = min(runner.S, runner.nb, args.batches)
    slot = runner.slots[0]
    rs, cs = stream.epoch_sides()
    with torch.cuda.stream(stream.side):
        hip.build_plans_sorted(rs, cs, 0, slot, n0, V, runner.sorted_ws)
        if runner.host_counts:
            slot.fetch_counts()
    torch.cuda.synchronize()
    if runner.host_counts:
        slot.adopt_counts(n0)
    staging = slot.plans[:n0]
    resident = [hip.build_plan(*(t.contiguous() for t in stream.batch(b)), V, chunk_cap=runner.cap, compact=True, d=tables.d)
                for b in range(n0)]
    torch.cuda.synchronize()
    print("staging plan: records %s, run words %s, borrow %s; host_counts %s" % (
        runner.records, runner.run_words, runner.borrow, staging[0].host_counts))
    print("resident plan: records %s, run words %s; host_counts %s" % (
        resident[0].r_crec is not None, resident[0].r_chunk_hw is not None, resident[0].host_counts))
    variants = {"staging plans (dealt), C loop": staging, "resident plans (same batches), C loop": resident}
    loss = runner.loss_out
    graphs = {}
    for name, plans in list(variants.items()):
        hip.steps_adagrad(plans, tables, hyper, loss, ws=runner.step_ws)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            hip.steps_adagrad(plans, tables, hyper, loss, ws=runner.step_ws)
        gname = name.replace("C loop", "hipGraph replay")
        graphs[gname] = g
        variants[gname] = plans
    res = {k: [] for k in variants}
    for rnd in range(args.rounds + 1):
        for name, plans in variants.items():
            once = graphs[name].replay if name in graphs else (lambda: hip.steps_adagrad(plans, tables, hyper, loss, ws=runner.step_ws))
            once()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                once()
            b.record()
            torch.cuda.synchronize()
            if rnd > 0:
                res[name].append(a.elapsed_time(b) * 1e3 / (3 * len(plans)))
    print("%s B=%d d=%d, %d batches: us per step, median / min over %d rounds" % (args.workload, B, d, n0, args.rounds))
    for name, x in res.items():
        print("  %-50s %.1f / %.1f" % (name, statistics.median(x), min(x)))


if __name__ == "__main__":
    main()
