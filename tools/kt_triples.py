#!/usr/bin/env python3
"""Per-launch durations of the fused step's three launches (row pass, col pass, apply) from a rocprofv3 --kernel-trace
CSV directory: the two pass launches share one kernel name, so the stats table cannot tell them apart.
Usage: kt_triples.py <dir>"""
import csv
import glob
import statistics as st
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "glove::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def kind(r):
    n = r["Kernel_Name"].split("(")[0]
    if "sidepass" in n:
        # the FUSE template argument: the run-merged schedule (the last argument in rounds 2 - 4: a bool in the first round-2
        # builds, 0 / 1 / 2 since; the fifth of six since round 5 added the loss head behind it)
        import re
        args = re.search(r"sidepass_kernel<([^>]*)>", n)
        a = [x.strip() for x in args.group(1).split(",")] if args else []
        fuse = a[4] if len(a) >= 5 else (a[-1] if a else "0")
        return "pass_fused" if fuse in ("true", "1", "2") else "pass"
    return "apply" if "apply_adagrad" in n else "triage" if "triage" in n else "other"


seq = [(kind(r), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3) for r in rows]
tri = []
for i in range(len(seq) - 2):
    if seq[i][0] == "pass_fused" and seq[i + 1][0] == "pass_fused":
        j = i + 2
        red = None
        if seq[j][0] == "triage" and j + 1 < len(seq):
            red, j = seq[j], j + 1
        if seq[j][0] == "apply":
            tri.append((seq[i], seq[i + 1], seq[j], red))
tri = tri[5:]
if tri:
    med = lambda xs: st.median(xs)
    reds = [t[3][1] for t in tri if t[3] is not None]
    print("%s: %d fused steps: row pass %.1f us, col pass %.1f us, id triage %s us, apply %.1f us, step (first start to last end) %.1f us" % (
        sys.argv[1], len(tri), med([t[0][1] for t in tri]), med([t[1][1] for t in tri]),
        ("%.1f" % med(reds)) if reds else "-", med([t[2][1] for t in tri]), med([t[2][3] - t[0][2] for t in tri])))
cl = [(seq[i][1], seq[i + 1][1]) for i in range(len(seq) - 1) if seq[i][0] == "pass" and seq[i + 1][0] == "apply"]
if cl:
    print("%s: %d two-launch steps: passes %.1f us, apply %.1f us" % (sys.argv[1], len(cl), st.median(c[0] for c in cl), st.median(c[1] for c in cl)))
