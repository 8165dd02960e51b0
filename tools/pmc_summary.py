#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel name. Usage: pmc_summary.py <dir> [name-filter]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "glove::"
    files = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"]
        if flt not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print("    %-24s n=%-4d avg=%.1f" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
