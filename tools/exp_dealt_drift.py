#!/usr/bin/env python3
"""bench.py reports the step kernels of a dealt C5 run "alone" at 370 us against 343 us for the static mode on the same box.
One process: the staging plans' steps and resident plans' steps (same batches) timed before the runner has run, after 200
steps of the runner (deals and index builds beside the steps), and again after an idle second.

Usage: python tools/exp_dealt_drift.py [--workload zipf_v2m_d128] [--batch-size 1048576]"""
import argparse
import sys
import time
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
    n0 = min(runner.S, runner.nb)
    loss = runner.loss_out

    def rebuilt_slot0():
        slot = runner.slots[0]
        rs, cs = stream.epoch_sides()
        with torch.cuda.stream(stream.side):
            hip.build_plans_sorted(rs, cs, 0, slot, n0, V, runner.sorted_ws)
            slot.fetch_counts()
        torch.cuda.synchronize()
        slot.adopt_counts(n0)
        return slot.plans[:n0]

    def resident_of_epoch():
        return [hip.build_plan(*(t.contiguous() for t in stream.batch(b)), V, chunk_cap=runner.cap, compact=True, d=tables.d)
                for b in range(n0)]

    def us_per_step(plans, reps=3):
        hip.steps_adagrad(plans, tables, hyper, loss, ws=runner.step_ws)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            hip.steps_adagrad(plans, tables, hyper, loss, ws=runner.step_ws)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / (reps * len(plans))

    def report(tag):
        st, rp = rebuilt_slot0(), resident_of_epoch()
        torch.cuda.synchronize()
        print("%-40s staging %.1f  resident %.1f  staging %.1f  resident %.1f us per step (%d batches)" % (
            tag, us_per_step(st), us_per_step(rp), us_per_step(st), us_per_step(rp), n0), flush=True)
        del rp

    report("before the runner has run")
    for rnd in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = 0
        while done < 200:
            done += runner.run(200 - done)
        torch.cuda.synchronize()
        print("runner: %.1f us per step over 200 steps" % ((time.perf_counter() - t0) * 1e6 / 200), flush=True)
        report("right after the runner's steps")
    time.sleep(2.0)
    report("after two idle seconds")


if __name__ == "__main__":
    main()
