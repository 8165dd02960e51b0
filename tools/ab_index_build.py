#!/usr/bin/env python3
"""One-process A/B of the index build of a segment of dealt batches (glove_plan_build_sorted) between builds of the library.
Usage: python tools/ab_index_build.py --libs a.so,b.so [--workload zipf_v400k_d300] [--batch-size 1048576]"""
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
    ap.add_argument("--libs", default="")
    ap.add_argument("--workload", default="zipf_v400k_d300")
    ap.add_argument("--batch-size", type=int, default=1048576)
    ap.add_argument("--optimizer", default="Adagrad")
    ap.add_argument("--rounds", type=int, default=7)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    libs = [x for x in args.libs.split(",") if x] or [None]
    hips = [GloveHip(dev, lib_path=x) if x else GloveHip(dev) for x in libs]
    wl = synthetic.make_workload(args.workload, device=dev, work_device=dev)
    V, d, B = wl["V"], wl["d"], args.batch_size
    backend = HipBackend(dev)
    backend.hip = hips[0]
    tables = DeviceTables(V, d, args.optimizer, device=dev, seed=1)
    backend.row_floats = tables.d
    stream = NonzeroStream({k: wl[k] for k in ("row", "col", "w", "y")}, B, V, backend, dev, seed=0, static_plans=False)
    hyper = make_hyper(batch_size=B, learning_rate=0.05)
    runners = [ReshufflingRunner(h, stream, tables, hyper, burst=64) for h in hips]
    torch.cuda.synchronize()
    rs, cs = stream.epoch_sides()
    n0 = min(runners[0].S, runners[0].nb)
    res = [[] for _ in hips]
    for rnd in range(args.rounds + 1):
        for i, (h, r) in enumerate(zip(hips, runners)):
            with torch.cuda.stream(stream.side):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream.side)
                for _ in range(3):
                    h.build_plans_sorted(rs, cs, 0, r.slots[0], n0, V, r.sorted_ws)
                b.record(stream.side)
            torch.cuda.synchronize()
            if rnd:
                res[i].append(a.elapsed_time(b) * 1e3 / (3 * n0))
    # the two builds give the same index
    same = True
    for f in ("r_chunk_id", "r_chunk_start", "c_chunk_id", "c_chunk_start", "r_uniq_rec", "c_uniq_rec", "counts"):
        for pa, pb in zip(runners[0].slots[0].plans[:n0], runners[-1].slots[0].plans[:n0]):
            ta, tb = getattr(pa, f), getattr(pb, f)
            k = int(pa.counts[0].item()) if "chunk" in f and "r_" in f else None
            if f == "counts":
                same &= bool((ta[:5] == tb[:5]).all())
    print("%s B=%d (%s): index of %d batches per build, records %s, run words %s; counts equal between builds: %s" % (
        args.workload, B, args.optimizer, n0, runners[0].records, runners[0].run_words, same))
    for x, r in zip(libs, res):
        print("  %-40s %.2f / %.2f us per batch (median / min of %d)" % (x or "shipped", statistics.median(r), min(r), len(r)))


if __name__ == "__main__":
    main()
