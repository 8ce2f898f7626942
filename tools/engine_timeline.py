#!/usr/bin/env python3
"""In-situ kernel timeline of the streaming step from the ENGINE's own events (pe_profile_enable(h, 2) /
pe_profile_timeline): every bracketed launch's start and end on the stream it ran on, without a profiler in the process
-- rocprofv3 serialises enough to stretch the 0.28 ms step to 0.4 (profiles/r03_timeline.txt), which hides exactly the
overlap one wants to see.

    python tools/engine_timeline.py [--steps 24] [--show 3] [--lag 4] [--validators 1048576] > timeline.txt

Builds bench.py's workload (BASELINE configs[3] shape by default), runs `--steps` streaming steps with rows + bits resident
in HBM, and prints the launches of the last `--show` steps in time order, the period between consecutive
k_g1_accumulate starts, and for each accumulate the gap since the previous one ended and which kernel ended last before
it started (the candidate for what it waited for).  Brackets: att_group = ingest + plan + members, att_validate =
validate_on_attestation / process_attestation asserts, bits_union, lmd, votes, tree, participation, g1_accumulate,
g1_tree, g1_normalise (the finish).  Needs a GPU."""
import argparse
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--show", type=int, default=3)
    ap.add_argument("--lag", type=int, default=4)
    ap.add_argument("--validators", type=int, default=1 << 20)
    ap.add_argument("--committees", type=int, default=2048)
    ap.add_argument("--blocks", type=int, default=4096)
    ap.add_argument("--parts", type=int, default=4)
    ap.add_argument("--cold", type=int, default=0,
                    help="K > 0: the picture of a K-step run that starts from a drained engine and ends with its drain (what a\n"
                         "short timed region pays for ramp and drain): every accumulation's start / end and the last kernel's end")
    ap.add_argument("--signed", action="store_true",
                    help="pe_aggregate_signed in pe_aggregate's place (one compressed signature per row, resident in HBM)")
    a = ap.parse_args()

    import torch

    torch.cuda.init()   # before the engine's own HIP runtime comes up: the other order leaves torch without a device
    import bench
    import pos_evolution_amd as pea

    args = types.SimpleNamespace(validators_local=a.validators, blocks=a.blocks, committees=a.committees, parts=a.parts,
                                 mixed_balances=False, host_arena=False, host_rows=False, with_shuffle=False,
                                 by_committee=False, world=1, shuffle_variant_from=a.steps)
    e = pea.Engine(device=0, max_committee_tables=a.steps + 3)
    w = bench.build_workload(e, args, 0, a.steps)
    e.set_pipeline_lag(a.lag)
    e.reuse_outputs(a.lag + 2)
    sigs = None
    if a.signed:
        import pos_evolution_amd.synth as synth
        from pos_evolution_amd import DeviceArena

        sg = synth.signature_points(e, len(w["steps"][0]["atts"]))
        sig_t = torch.from_numpy(sg.reshape(-1).copy()).cuda()
        sigs = DeviceArena(sig_t.data_ptr(), sig_t.numel(), keep=sig_t)
    warm = max(a.lag + 2, a.steps - a.show - a.lag - 2)
    if a.cold:
        warm = a.steps - a.cold
    for s in range(warm):
        bench.run_step_single(e, w, w["steps"][s], pipelined=True, lagged=True, sync_head=False, sigs=sigs)
    e.drain()
    import gc

    gc.collect()
    gc.disable()                   # as bench.py does over its timed steps: with torch loaded a full collection is tens of ms
                                   # (a 35-40 ms hole in two of this tool's pictures in round 6)
    e.profile_enable(2)
    e.profile_reset()              # time zero
    for s in range(warm, a.steps):
        bench.run_step_single(e, w, w["steps"][s], pipelined=True, lagged=True, sync_head=False, sigs=sigs)
    e.drain()
    tl = e.profile_timeline()
    e.profile_enable(0)
    acc = [r for r in tl if r[0] == "g1_accumulate"]
    if a.cold and acc:
        t00 = min(r[1] for r in tl)
        end = max(r[1] + r[2] for r in tl)
        print(f"# {a.cold} steps from a drained engine: first launch at 0, everything done at {(end - t00) * 1e3:.1f} us "
              f"= {(end - t00) * 1e3 / a.cold:.1f} us per step")
        print("step  accumulate_start  accumulate_end  dur")
        for i, (_, t0, dt) in enumerate(acc):
            print(f"{i:4d} {(t0 - t00) * 1e3:17.1f} {(t0 + dt - t00) * 1e3:15.1f} {dt * 1e3:6.1f}")
        tail = sorted(((r[1] + r[2] - t00) * 1e3, r[0]) for r in tl)[-6:]
        print("last kernels to end:", ", ".join(f"{n} {t:.0f}" for t, n in tail))
        print()
        print(f"{'kernel':16s} {'start':>9s} {'end':>9s} {'dur':>8s}")
        for name, t0, dt in tl:
            print(f"{name:16s} {(t0 - t00) * 1e3:9.1f} {(t0 + dt - t00) * 1e3:9.1f} {dt * 1e3:8.1f}")
        return 0
    if len(acc) < 2:
        print("no accumulations bracketed", file=sys.stderr)
        return 1
    first = acc[max(0, len(acc) - a.show - 1)][1]
    print(f"# {a.validators} validators, {a.committees} committees, {a.blocks} blocks, lag {a.lag}; times in us from the "
          f"start of the accumulation {len(acc) - a.show - 1} of {len(acc)} timed steps")
    print(f"{'kernel':16s} {'start':>9s} {'end':>9s} {'dur':>8s}")
    for name, t0, dt in tl:
        if t0 >= first - 1e-3:
            print(f"{name:16s} {(t0 - first) * 1e3:9.1f} {(t0 + dt - first) * 1e3:9.1f} {dt * 1e3:8.1f}")
    print()
    print("accumulate  period_us  gap_after_previous_us  last_kernel_to_end_before_it (end -> accumulate start, us)")
    for i in range(1, len(acc)):
        t0 = acc[i][1]
        prev_end = acc[i - 1][1] + acc[i - 1][2]
        before = [(r[1] + r[2], r[0]) for r in tl if r[1] + r[2] <= t0 + 1e-6 and r[0] != "g1_accumulate"]
        last = max(before) if before else (float("nan"), "-")
        print(f"{i:10d} {(t0 - acc[i - 1][1]) * 1e3:10.1f} {(t0 - prev_end) * 1e3:22.1f}  {last[1]} ({(t0 - last[0]) * 1e3:.1f})")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
