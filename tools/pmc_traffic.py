#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per kernel from two rocprofv3 --pmc passes -> JSON for profiles/.
Per MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half of a wide
(16 B/lane) read stream, so the read side is doubled; WRITE_SIZE is exact for 16 B/lane stores.
Infinity-Cache hits are included (these are fabric-side requests, an upper bound on HBM bytes).
Usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [key=value ...]"""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "glove::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void glove::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    meta = dict(kv.split("=", 1) for kv in sys.argv[4:])
    out = {"meta": meta, "kernels": {}}
    step = ("sidepass", "rowpass", "colpass", "apply_adagrad")          # the kernels of one sparse-Adagrad step
    total = 0.0
    for k in sorted(set(fetch) | set(write)):
        rd, wr = 2.0 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        out["kernels"][k] = {"FETCH_SIZE_KiB_raw": fetch.get(k), "WRITE_SIZE_KiB_raw": write.get(k),
                             "read_bytes_corrected": rd, "write_bytes": wr}
        if any(k.startswith(x) for x in step):
            total += rd + wr
    out["traffic_bytes_per_step"] = total
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["traffic_bytes_per_step"]))


if __name__ == "__main__":
    main()
