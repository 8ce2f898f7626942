#!/usr/bin/env python
"""Kernel timeline of a few consecutive bench steps from a rocprofv3 rocpd database (kernel-trace):
start / end of every kernel relative to the first union of the window (k_bits_union, or the paired launch that carries it
since round 5: k_pair_union_tree), with overlaps visible.
usage: rocpd_timeline.py results.db [first_step] [n_steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = sorted(db.execute(f"select {name_col}, start, end from kernels").fetchall(), key=lambda r: r[1])


def short(n):
    n = n.split("(")[0].replace("posevo::", "").replace("void ", "")
    return n[:28]


unions = [i for i, r in enumerate(rows) if "k_pair_union_tree" in r[0]]
if len(unions) < 3:  # unpaired runs (POSEVO_PAIR=0, rounds 1-4)
    unions = [i for i, r in enumerate(rows) if "k_bits_union" in r[0]]
if len(unions) < 2:
    raise SystemExit("no step boundaries (k_pair_union_tree / k_bits_union) in this trace")
nsteps = min(nsteps, len(unions) - 1)
if len(unions) <= first + nsteps:
    first = max(0, len(unions) - nsteps - 1)
lo, hi = unions[first], unions[first + nsteps]
t0 = rows[lo][1]
print(f"{'kernel':30s} {'start_us':>10s} {'end_us':>10s} {'dur_us':>8s}")
for name, s, e in rows[lo:hi]:
    print(f"{short(name):30s} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}")
# also kernels that started before the window but end inside it
for name, s, e in rows[max(0, lo - 12):lo]:
    if e > t0:
        print(f"{'(earlier) ' + short(name):30s} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}")
per = [(rows[unions[k + 1]][1] - rows[unions[k]][1]) / 1e3 for k in range(first, first + nsteps)]
print("step periods (us):", ", ".join(f"{p:.1f}" for p in per))
