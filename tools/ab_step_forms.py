#!/usr/bin/env python3
"""Which form of the sparse Adagrad step wins where: forms 1 (two launches), 3 (fused three launches) and
4 (fused on a twinned row table) on the SAME resident plans, one process, interleaved rounds (cdna guide rule 24), HIP events
around hipGraph replays.  python3 tools/ab_step_forms.py --workload text8_v50k_d300 --batch-size 131072"""
import argparse
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, auto_chunk_cap, make_hyper, row_width  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="text8_v50k_d300")
    ap.add_argument("--batch-size", type=int, default=131072)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--forms", default="1,3,4")
    ap.add_argument("--lib", default="", help="another build of libglove_hip.so (default: the one in the tree)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    hip = GloveHip(dev, lib_path=a.lib, any_abi=True) if a.lib else GloveHip(dev)
    wl = synthetic.make_workload(a.workload, device=dev, work_device=dev)
    V, d, B = wl["V"], wl["d"], a.batch_size
    nb = min(a.batches, wl["row"].numel() // B)
    dpad = row_width(V, d)          # the stored row stride (what workspace queries and thresholds take)
    cap = auto_chunk_cap(B, V, dpad)
    plans = [hip.build_plan(*(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")), V, chunk_cap=cap).compact(hip.lib, dpad, records=True)
             for b in range(nb)]
    ids = statistics.mean(p.host_counts[1] + p.host_counts[3] for p in plans)
    ws = torch.empty(max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, dpad) for p in plans), dtype=torch.uint8, device=dev)
    loss = torch.zeros(4, device=dev)
    runs = {}
    for form in [int(x) for x in a.forms.split(",")]:
        t = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
        if form == 4:
            t.enable_twin()
        h = make_hyper(learning_rate=0.05, batch_size=B, step_form=form)
        for p in plans:
            hip.step_adagrad(p, t, h, loss, ws)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for p in plans:
                hip.step_adagrad(p, t, h, loss, ws)
        runs[form] = (g, t, [])
    for rnd in range(a.rounds + 1):
        for form, (g, t, res) in runs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res.append(e0.elapsed_time(e1) * 1e3 / (3 * nb))
    print("%s B=%d V=%d d=%d cap=%d: %.0f distinct ids per step = %.0f MB of touched rows (ids x d x 16 B)" % (
        a.workload, B, V, d, cap, ids, ids * dpad * 16 / 1e6))
    for form, (_, _, res) in runs.items():
        print("  form %d: %.1f us per step (min %.1f)" % (form, statistics.median(res), min(res)))


if __name__ == "__main__":
    main()
