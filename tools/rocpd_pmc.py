#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (counters_collection view).
usage: rocpd_pmc.py results.db COUNTER   -> prints {kernel: {"launches": n, "avg": value_per_launch}} as JSON
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; the caller applies the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE x 2 for wide coalesced reads)."""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? "
                  "group by kernel_name, dispatch_id", (sys.argv[2],)).fetchall()
agg = {}
for name, _, v in rows:
    short = name.split("(")[0].replace("posevo::", "").replace("void ", "")
    agg.setdefault(short, []).append(v)
out = {}
for k, vals in agg.items():
    vals.sort()
    out[k] = {"launches": len(vals), "avg": sum(vals) / len(vals), "median": vals[len(vals) // 2], "max": vals[-1]}
print(json.dumps(out, indent=1))
