#!/usr/bin/env python3
"""Reads bench.py JSON lines on stdin and prints one short line each: label, us/step, G nonzeros/s, frac, chunk cap.
Usage: python bench.py --single ... | python tools/bench_brief.py LABEL"""
import json
import sys

label = " ".join(sys.argv[1:])
for line in sys.stdin:
    if line.startswith("{"):
        j = json.loads(line)
        r = j.get("roofline", {})
        print("%s: %.1f us/step, %.3f G nonzeros/s, frac %.3f, cap %s, kernels %s" % (
            label, j["ms_per_step"] * 1e3, j["value"] / 1e9, r.get("frac", 0.0), j["config"].get("chunk_cap"),
            {k: round(v, 1) for k, v in (r.get("kernel_us") or {}).items()}))
