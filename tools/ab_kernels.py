#!/usr/bin/env python3
"""In-process A/B timing of kernel variants (cdna guide rule 24: interleaved rounds, one process).
Usage: python tools/ab_kernels.py [--workload text8_d64] [--batch-size 131072] [--caps 32,16]"""
import argparse
import ctypes as C
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, make_hyper  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="text8_d64")
    ap.add_argument("--batch-size", type=int, default=131072)
    ap.add_argument("--caps", default="32,16")
    ap.add_argument("--libs", default="", help="comma-separated builds of libglove_hip.so to compare (default: the shipped one)")
    ap.add_argument("--any-abi", action="store_true", help="accept builds of another ABI version (older commits)")
    ap.add_argument("--step-form", type=int, default=0)
    ap.add_argument("--only", default="", help="comma-separated subset of rowpass,colpass,passes,apply,step")
    ap.add_argument("--twin", action="store_true", help="twinned row table (step form 4 under auto)")
    ap.add_argument("--batches", type=int, default=8, help="resident batches cycled through (1: the plan stays cache-warm)")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--check", action="store_true", help="one step per build from equal tables: report the largest difference to the first build")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    libs = [x for x in args.libs.split(",") if x] or [None]
    hips = [GloveHip(dev, lib_path=x, any_abi=args.any_abi) if x else GloveHip(dev) for x in libs]
    hip = hips[0]
    wl = synthetic.make_workload(args.workload, device=dev, work_device=dev)
    V, d, B = wl["V"], wl["d"], args.batch_size
    nb = min(args.batches, wl["row"].numel() // B)
    tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
    if args.twin:
        tables.enable_twin()
    hyper = make_hyper(learning_rate=0.05, batch_size=B, step_form=args.step_form)
    loss = torch.zeros(4, device=dev)
    configs = []
    for cap in [int(c) for c in args.caps.split(",")]:
        for v in range(len(hips)):       # every build indexes the batches itself (the plan's record layout is the library's own)
            plans = [hips[v].build_plan(*(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")), V,
                                        chunk_cap=cap, compact=True, d=tables.d) for b in range(nb)]
            configs.append((cap, v, plans))
    ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, B, tables.d), dtype=torch.uint8, device=dev)
    res = {(c, v): {"rowpass": [], "colpass": [], "passes": [], "apply": [], "step": []} for c, v, _ in configs}
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for rnd in range(args.rounds + 1):
        for cap, v, plans in configs:
            hip = hips[v]
            for name in [n for n in ("rowpass", "colpass", "passes", "apply", "step") if not args.only or n in args.only.split(",")]:
                fn = {"rowpass": lambda p: hip.rowpass(p, tables, hyper, ws),
                      "passes": lambda p: hip.passes(p, tables, hyper, ws),
                      "colpass": lambda p: hip.colpass(p, tables, hyper, ws),
                      "apply": lambda p: hip.apply_adagrad(p, tables, hyper, loss, ws),
                      "step": lambda p: hip.step_adagrad(p, tables, hyper, loss, ws)}[name]
                for i in range(4):
                    fn(plans[i % nb])
                a, b = ev(), ev()
                a.record()
                for i in range(args.reps):
                    fn(plans[i % nb])
                b.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    res[(cap, v)][name].append(a.elapsed_time(b) * 1e3 / args.reps)
    if args.check:
        outs = []
        for v in range(len(hips)):
            t2 = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
            if args.twin:
                t2.enable_twin()
            plans = [c[2] for c in configs if c[1] == v][0]
            for b in range(min(nb, 2)):
                hips[v].step_adagrad(plans[b], t2, hyper, loss, ws)
            torch.cuda.synchronize()
            outs.append([x.clone() for x in (t2.R, t2.C, t2.br, t2.bc, t2.s1["R"], t2.s1["C"], loss)])
        for v in range(1, len(hips)):
            print("check lib=%s vs %s: max |diff| %s" % (libs[v], libs[0], ["%.3g" % (a - b).abs().max().item() for a, b in zip(outs[0], outs[v])]))
    print("%s B=%d d=%d  (us per launch incl. launch gaps; median / min over %d rounds)" % (args.workload, B, d, args.rounds))
    for (cap, v), r in res.items():
        print("cap=%-3d lib=%s  " % (cap, libs[v] or "shipped") + "  ".join(
            "%s %.2f/%.2f" % (k, statistics.median(x), min(x)) for k, x in r.items() if x))


if __name__ == "__main__":
    main()
