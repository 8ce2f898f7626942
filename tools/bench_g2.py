#!/usr/bin/env python
"""Kernel-time measurement of pe_g2_sum (G2 signature aggregation) on the bench workload's shape:
2048 committees x 512 signatures gathered from a table of distinct points.  Prints one JSON line.
    python tools/bench_g2.py [n_groups] [group_size] [table_size]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pos_evolution_amd as pea  # noqa: E402
from oracle import g1, g2  # noqa: E402  (checker only: one group is verified against the closed form)


def main():
    n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    table = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
    a, b = 0x1234567, 0x89ABCDE
    pts = np.stack([np.frombuffer(g2.to_bytes192(p), dtype=np.uint8) for p in g2.synthetic_points(table, a, b)])
    rng = np.random.default_rng(1)
    n = n_groups * size
    index = rng.integers(0, table, size=n, dtype=np.uint32)
    offsets = (np.arange(n_groups + 1, dtype=np.uint64) * size).astype(np.uint32)
    e = pea.Engine()
    out = e.g2_sum(pts, offsets, index=index)
    idx = index[:size].astype(object)
    assert out[0].tobytes() == g2.to_bytes192(g2.mul((size * a + int(idx.sum()) * b) % g1.R_ORDER, g2.G2))
    e.profile_enable(True)
    e.profile_reset()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        e.g2_sum(pts, offsets, index=index)
    wall = (time.perf_counter() - t0) / reps
    prof = e.profile()
    acc = prof["g2_accumulate"]["total_ms"] / max(1, prof["g2_accumulate"]["launches"])
    fin = prof["g2_normalise"]["total_ms"] / max(1, prof["g2_normalise"]["launches"])
    print(json.dumps({"workload": f"{n_groups} groups x {size} G2 points (table {table})", "points": n,
                      "k_g2_accumulate_ms": acc, "k_g2_finish_ms": fin,
                      "g2_adds_per_s_kernel": n / ((acc + fin) * 1e-3),
                      "mont_products_per_s": 36 * n / (acc * 1e-3), "wall_ms_per_call": wall * 1e3,
                      "checked_against_oracle": True}))


if __name__ == "__main__":
    main()
