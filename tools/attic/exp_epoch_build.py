#!/usr/bin/env python3
"""The epoch pipeline alone on an idle GPU (no steps): masters once, then `--reps` x (deal one epoch, index every batch of it
a segment at a time).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel times behind DESIGN.md §3c, or alone
for HIP-event totals.   python3 tools/exp_epoch_build.py [--workload zipf_v400k_d300] [--batch-size 1048576] [--segment 13]"""
import argparse
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="zipf_v400k_d300")
    ap.add_argument("--batch-size", type=int, default=1048576)
    ap.add_argument("--segment", type=int, default=13)
    ap.add_argument("--chunk-cap", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-records", action="store_true")
    a = ap.parse_args()
    from trainer import synthetic
    from trainer.hip_api import GloveHip, Pairs, PlanBlock, auto_chunk_cap
    hip = GloveHip("cuda:0")
    wl = synthetic.make_workload(a.workload, seed=0, device="cuda:0", work_device="cuda:0")
    V, d, B = wl["V"], wl["d"], a.batch_size
    n = wl["row"].numel()
    nb = n // B
    cap = a.chunk_cap or auto_chunk_cap(B, V, (d + 3) // 4 * 4)
    t0 = time.perf_counter()
    m = hip.build_masters(wl["row"], wl["col"], wl["w"], wl["y"], V)
    torch.cuda.synchronize()
    print("masters: %.1f ms for %d pairs" % ((time.perf_counter() - t0) * 1e3, n))
    rs, cs = Pairs(n, "cuda:0"), Pairs(n, "cuda:0")
    dws = hip.deal_workspace(n, B, "cuda:0")
    S = min(a.segment, nb)
    block = PlanBlock([hip.staging_plan(B, V, cap, "cuda:0", records=not a.no_records) for _ in range(S)])
    ws = torch.empty(hip.lib.glove_plan_sorted_workspace_bytes(B, S), dtype=torch.uint8, device="cuda:0")

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e
    deal_us, build_us = [], []
    for r in range(a.reps):
        e0 = ev()
        hip.deal_epoch(m, B, 1234567 + r, rs, cs, dws)
        e1 = ev()
        for first in range(0, nb, S):
            hip.build_plans_sorted(rs, cs, first, block, min(S, nb - first), V, ws)
        e2 = ev()
        torch.cuda.synchronize()
        deal_us.append(e0.elapsed_time(e1) * 1e3)
        build_us.append(e1.elapsed_time(e2) * 1e3)
    deal_us.sort(); build_us.sort()
    print("%s B=%d V=%d: %d batches per epoch, cap %d, segment %d" % (a.workload, B, V, nb, cap, S))
    print("deal: %.1f us per epoch = %.2f us per batch;  index: %.1f us per epoch = %.2f us per batch" % (
        deal_us[len(deal_us) // 2], deal_us[len(deal_us) // 2] / nb, build_us[len(build_us) // 2], build_us[len(build_us) // 2] / nb))
    c = block.plans[0].counts.tolist()
    print("counts of batch %d: %s" % (nb - nb % S if nb % S else nb - S, c))


if __name__ == "__main__":
    main()
