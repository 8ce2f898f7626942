#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table.
usage: rocpd_stats.py results.db [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    d = (e - s) / 1e3  # ns -> us
    a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values())
lines = [f"{'kernel':70s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}"]
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    short = name if len(name) <= 70 else name[:67] + "..."
    lines.append(f"{short:70s} {a[0]:7d} {a[1]:12.1f} {a[1] / a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100 * a[1] / total:6.2f}")
text = "\n".join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
