#!/usr/bin/env python3
"""Which HIP streams share a hardware queue, and how much of the side stream's work ran beside a pass kernel.
Usage: python tools/attic/queue_overlap.py <rocprofv3 results.db>"""
import bisect
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("stream queue launches busy_ms first_kernel")
for r in cur.execute("select stream_id, queue_id, count(*), sum(end-start)/1e6, min(name) from kernels group by stream_id, queue_id"):
    print(r[0], r[1], r[2], round(r[3], 1), r[4][:70])
rows = list(cur.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
side = [(s, e) for n, s, e, st, q in rows if any(k in n for k in ("side_emit", "side_tiles", "csort_", "fill_records"))]
main = [(s, e) for n, s, e, st, q in rows if "sidepass" in n]
ms = [m[0] for m in main]
ov = tot = 0
for s, e in side:
    tot += e - s
    i = bisect.bisect_left(ms, s)
    for j in range(max(0, i - 3), min(len(main), i + 3)):
        a, b = main[j]
        ov += max(0, min(e, b) - max(s, a))
print("deal + index kernels: %.2f ms in all, %.2f ms of it beside a pass kernel" % (tot / 1e6, ov / 1e6))
