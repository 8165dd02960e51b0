#!/usr/bin/env python3
"""Gaps between consecutive kernels of one name in a rocprofv3 --kernel-trace CSV: where a chain of short step kernels loses
time (launch-to-launch gaps, graph boundaries, kernels of a side stream squeezed in between).
Usage: tools/kt_gaps.py <dir with *_kernel_trace.csv> <kernel name substring> [last N kernels]"""
import csv
import glob
import sys


def main():
    d, name = sys.argv[1], sys.argv[2]
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if name in r[2]]
    if last:
        idx = idx[-last:]
    if len(idx) < 2:
        print("fewer than two kernels match", name)
        return
    durs = [(rows[i][1] - rows[i][0]) / 1e3 for i in idx]
    gaps, between = [], {}
    for a, b in zip(idx, idx[1:]):
        gaps.append((rows[b][0] - rows[a][1]) / 1e3)
        for k in range(a + 1, b):
            n = rows[k][2].split("(")[0][-60:]
            o = between.setdefault(n, [0, 0.0])
            o[0] += 1
            o[1] += (rows[k][1] - rows[k][0]) / 1e3
    span = (rows[idx[-1]][1] - rows[idx[0]][0]) / 1e3
    n = len(idx)
    gs = sorted(gaps)
    print("%d kernels '%s': span %.1f us = %.2f us each; duration avg %.2f (min %.2f, max %.2f)" % (
        n, name, span, span / n, sum(durs) / n, min(durs), max(durs)))
    print("gaps start-after-end: median %.2f, mean %.2f, p90 %.2f, max %.2f; sum of gaps over 5 us: %.1f us in %d gaps" % (
        gs[len(gs) // 2], sum(gs) / len(gs), gs[int(len(gs) * 0.9)], gs[-1], sum(g for g in gs if g > 5), sum(1 for g in gs if g > 5)))
    hist = {}
    for g in gaps:
        b = "<1" if g < 1 else "1-2" if g < 2 else "2-3" if g < 3 else "3-5" if g < 5 else "5-10" if g < 10 else "10-30" if g < 30 else ">30"
        hist[b] = hist.get(b, 0) + 1
    print("gap histogram (us):", hist)
    print("kernels that started between two of them (name, calls, total us):")
    for k, v in sorted(between.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  %-60s %6d %10.1f" % (k, v[0], v[1]))


if __name__ == "__main__":
    main()
