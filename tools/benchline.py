#!/usr/bin/env python
"""Reads bench.py's JSON line on stdin and prints the few numbers compared between variants."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get("kernel_avg_ms", {})
print(" ".join(sys.argv[1:]), "ms/step", round(d["ms_per_step"], 3), "head_p50_us", round(d["get_head_p50_us"], 1),
      {n: round(v * 1e3, 1) for n, v in k.items() if v})
