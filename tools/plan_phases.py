#!/usr/bin/env python3
"""Where does k_att_plan spend its time?  Needs the variant library built by
    tools/build_variant.sh plantime -DPOSEVO_PLAN_TIMING
(POSEVO_LIB_PATH=build/variants/libposevo_plantime.so): runs bench.py's workload for a few streaming steps and prints the
wall-clock stamps (100 MHz) the kernel took at its phase boundaries in its LAST launch.  Needs a GPU."""
import ctypes as C
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    torch.cuda.init()
    import bench
    import pos_evolution_amd as pea

    steps = 12
    args = types.SimpleNamespace(validators_local=1 << 20, blocks=4096, committees=2048, parts=4, mixed_balances=False,
                                 host_arena=False, host_rows=False, with_shuffle=False, by_committee=False, world=1,
                                 shuffle_variant_from=steps)
    e = pea.Engine(device=0, max_committee_tables=steps + 3)
    w = bench.build_workload(e, args, 0, steps)
    e.set_pipeline_lag(4)
    e.reuse_outputs(6)
    lib = pea._abi.load()
    names = ["1 group ids", "2 resolve committees", "2b scans", "3 block size", "4 descriptors + counts", "5 rows per committee", "6 plan"]
    for mode in ("streaming", "alone"):
        for s in range(steps):
            if mode == "streaming":
                bench.run_step_single(e, w, w["steps"][s], pipelined=True, lagged=True, sync_head=False)
            else:
                bench.run_step_single(e, w, w["steps"][s], pipelined=False, lagged=False, sync_head=True)
        e.drain()
        st = (C.c_ulonglong * 16)()
        assert lib.pe_debug_plan_stamps(st) == 0
        t = [st[i] for i in range(8)]
        print(f"k_att_plan, last launch of {steps} {mode} steps: " +
              ", ".join(f"{names[i]} {(t[i + 1] - t[i]) / 100:.1f} us" for i in range(6)) + f"; total {(t[6] - t[0]) / 100:.1f} us")
        # the second round needs fresh epochs
        if mode == "streaming":
            e.close()
            e = pea.Engine(device=0, max_committee_tables=steps + 3)
            w = bench.build_workload(e, args, 0, steps)


if __name__ == "__main__":
    main()
