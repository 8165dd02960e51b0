#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a small text table
(the summaries committed under profiles/).  Usage: prof_summary.py <dir> [top_n]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    files = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)
    if not files:
        raise SystemExit("no *_kernel_stats.csv under " + d)
    rows = list(csv.DictReader(open(files[0])))
    print("%-72s %7s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for r in rows[:top]:
        print("%-72s %7s %12.1f %10.2f %10.2f %10.2f %7s" % (
            r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))


if __name__ == "__main__":
    main()
