#!/usr/bin/env python3
"""Does the order of the pairs INSIDE an id's run matter to the fused passes?

A dealt batch (glove_epoch_deal) arrives sorted by (row id, col id) on its row side and by (col id, row id) on its col side:
every run of equal ids walks its partners in ascending id order, and ids are frequency ranks (vocab.txt is sorted by count,
the synthetic generators likewise), so ALL lane groups gather the same few hot rows first.  A batch indexed from the stream
(glove_plan_build, stable sorts) keeps the stream's random order inside a run.  Same batches, same ids, same chunks, one
process, interleaved rounds: resident plans built from the batch as it streams, from the batch pre-sorted by (row, col) —
which is what the deal delivers — and from the batch pre-sorted by (row, scrambled col).

Usage: python tools/exp_pair_order.py [--workload zipf_v2m_d128] [--batch-size 1048576] [--batches 8]"""
import argparse
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, make_hyper  # noqa: E402


def scramble(x: torch.Tensor, bits: int) -> torch.Tensor:
    """bit reversal inside `bits` bits (a bijection of [0, 2^bits))"""
    x = x.long()
    out = torch.zeros_like(x)
    for b in range(bits):
        out |= ((x >> b) & 1) << (bits - 1 - b)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="zipf_v2m_d128")
    ap.add_argument("--batch-size", type=int, default=1048576)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=24)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hip = GloveHip(dev)
    wl = synthetic.make_workload(args.workload, device=dev, work_device=dev)
    V, d, B = wl["V"], wl["d"], args.batch_size
    nb = min(args.batches, wl["row"].numel() // B)
    bits = max(2, (V - 1).bit_length())
    tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
    tables.maybe_enable_twin()
    hyper = make_hyper(learning_rate=0.05, batch_size=B)
    loss = torch.zeros(4, device=dev)
    variants = {}
    for name in ("stream order", "sorted (row, col)", "sorted (row, scrambled col)"):
        plans = []
        for b in range(nb):
            r, c, w, y = (wl[k][b * B:(b + 1) * B] for k in ("row", "col", "w", "y"))
            if name != "stream order":
                minor = c.long() if name == "sorted (row, col)" else scramble(c, bits)
                order = torch.argsort(r.long() << 32 | minor, stable=True)
                r, c, w, y = r[order], c[order], w[order], y[order]
            plans.append(hip.build_plan(r.contiguous(), c.contiguous(), w.contiguous(), y.contiguous(), V, chunk_cap=0,
                                        compact=True, d=tables.d))
        variants[name] = plans
    ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, B, tables.d), dtype=torch.uint8, device=dev)
    res = {k: [] for k in variants}
    for rnd in range(args.rounds + 1):
        for name, plans in variants.items():
            for i in range(4):
                hip.step_adagrad(plans[i % nb], tables, hyper, loss, ws)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(args.reps):
                hip.step_adagrad(plans[i % nb], tables, hyper, loss, ws)
            b.record()
            torch.cuda.synchronize()
            if rnd > 0:
                res[name].append(a.elapsed_time(b) * 1e3 / args.reps)
    print("%s B=%d d=%d, %d resident batches: us per step, median / min over %d rounds" % (args.workload, B, d, nb, args.rounds))
    for name, x in res.items():
        print("  %-30s %.1f / %.1f" % (name, statistics.median(x), min(x)))


if __name__ == "__main__":
    main()
