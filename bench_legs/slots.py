"""bench.py `slot_cadence`: the four functions at the cadence a client calls them, one step per slot."""
import functools
import os
import time

import numpy as np

from .verify import step_digest
from .workload import load_registry


def slot_cadence(pea, w, device, n_epochs, lag):
    """The same four functions at the cadence a client calls them (pe:934-944, 963, 1102, 1536): one step per SLOT -- on_tick
    (which resets the proposer boost, pe:943), then the aggregates of the slot that just ended (64 committees x 4 partial
    aggregates = 256 rows at configs[3]) through pe_aggregate -> pe_on_attestation_batch -> pe_get_head ->
    pe_process_attestation_batch, 32 steps per epoch, the participation rotation at the epoch boundary and the NEXT epoch's
    committee shuffle (pe_compute_committees_async) enqueued once per epoch, beside the slots' steps.  One warm-up epoch, then
    n_epochs - 1 timed ones through streaming pipelines (throughput, per-step period), then one more epoch with the head
    polled inside every step (the latency a client sees from its on_tick to the slot's head).  Every timed slot-step is
    replayed with synchronous host-row calls on a fresh engine and compared by digest.  -> the `slot_cadence` object."""
    import torch
    from pos_evolution_amd import RESIDENT, ROWS_RESIDENT, DeviceArena, DeviceRows
    from pos_evolution_amd._abi import pe_state_ctx

    spe, tree = w["spe"], w["tree"]
    epochs = w["steps"][:n_epochs + 1]           # + 1: the latency pass
    keep = []

    def make_engine():
        e = pea.Engine(device=device, max_committee_tables=len(epochs) + 4)
        e.store_init(0, 0, tree.roots[0].tobytes())
        for i in range(1, tree.roots.shape[0]):
            e.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
        load_registry(e, w)
        for st in epochs:
            e.set_committees(st["epoch"], st["comm"].offsets, st["comm"].members)
        return e

    slots = []
    for k, st in enumerate(epochs):
        order = np.argsort(st["atts"]["slot"], kind="stable")
        atts = np.ascontiguousarray(st["atts"][order])
        rows_t = torch.from_numpy(atts.view(np.uint8).reshape(-1).copy()).cuda()
        arena_t = torch.from_numpy(st["arena"]).cuda()
        keep += [rows_t, arena_t]
        arena_in = DeviceArena(arena_t.data_ptr(), arena_t.numel(), keep=arena_t)
        bounds = np.searchsorted(atts["slot"], st["epoch"] * spe + np.arange(spe + 1))
        cps = (st["comm"].offsets.size - 1) // spe
        for s in range(spe):
            lo, hi = int(bounds[s]), int(bounds[s + 1])
            S = st["epoch"] * spe + s + 1           # the slot whose tick makes slot S - 1's attestations valid (pe:1411)
            c = pe_state_ctx()
            c.slot = S
            c.chain_tip_root[:] = tree.roots[tree.roots.shape[0] - 1].tobytes()
            c.current_justified_root[:] = tree.roots[0].tobytes()
            c.previous_justified_root[:] = tree.roots[0].tobytes()
            c.base_reward_per_increment = 2264
            sl = dict(S=S, rotate=(S % spe == 0), atts=atts[lo:hi], arena=st["arena"], arena_in=arena_in, cap=cps, ctx=c,
                      rows_in=DeviceRows(rows_t.data_ptr() + 144 * lo, hi - lo, keep=rows_t))
            if s == 0 and k + 1 < len(epochs):     # MIN_SEED_LOOKAHEAD: epoch E's first slot can shuffle epoch E + 1
                nxt = epochs[k + 1]
                sl["shuffle"] = (nxt["epoch"], nxt["ep_seed"], w["bal"].size, nxt["comm"].offsets.size - 1, 90)
            slots.append(sl)
    torch.cuda.synchronize()

    def step(e, sl, resident, lagged, sync_head):
        e.on_tick(sl["S"] * 12)
        if sl["rotate"]:
            e.participation_rotate()
        if resident and "shuffle" in sl:
            e.compute_committees_async(*sl["shuffle"])
        if not resident:
            agg = e.aggregate(packed=(sl["atts"], sl["arena"]), want_aggregate_pubkeys=True)
            rows = agg["atts"]
            status, _, count = e.on_attestation_batch(packed=(rows, agg["out_arena"]))
            st2, num = e.process_attestation_batch(sl["ctx"], packed=(rows, agg["out_arena"]))
            return dict(agg=agg, status=status, count=count, pstatus=st2, numerators=num, head=e.get_head())
        with e.pipeline(lagged=lagged):
            agg = e.aggregate(packed=(sl["rows_in"], sl["arena_in"]), want_aggregate_pubkeys=True)
            status, _, count = e.on_attestation_batch(packed=(ROWS_RESIDENT, RESIDENT), cap=sl["cap"])
            head = e.get_head() if sync_head else e.get_head_async()
            st2, num = e.process_attestation_batch(sl["ctx"], packed=(ROWS_RESIDENT, RESIDENT), cap=sl["cap"])
        return dict(agg=agg, status=status, count=count, pstatus=st2, numerators=num, head=head)

    e = make_engine()
    e.set_pipeline_lag(lag)
    e.reuse_outputs(len(slots) + 2)
    n_warm, n_timed = spe, spe * (n_epochs - 1)
    got = [step(e, sl, True, True, False) for sl in slots[:n_warm]]
    e.drain()
    e.fill_ring()
    torch.cuda.synchronize()
    timeline_path = os.environ.get("POSEVO_SLOT_TIMELINE")   # diagnostic: the engine's own event timeline of the timed slots
    if timeline_path:
        e.profile_enable(2)
        e.profile_reset()
    t0 = time.perf_counter()
    stamps = [t0]
    for sl in slots[n_warm:n_warm + n_timed]:
        got.append(step(e, sl, True, True, False))
        stamps.append(time.perf_counter())
    e.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if timeline_path:
        tl = e.profile_timeline()
        e.profile_enable(0)
        with open(timeline_path, "w") as f:
            f.write(f"# slot cadence, {n_timed} slot-steps in {dt * 1e3:.2f} ms; kernel start end dur (us)\n")
            for name, a0, d0 in tl:
                f.write(f"{name:22s} {a0 * 1e3:10.1f} {(a0 + d0) * 1e3:10.1f} {d0 * 1e3:8.1f}\n")
    n_att = int(sum(int(np.asarray(r["count"]).sum()) for r in got[n_warm:]))
    # the latency pass: one wait per slot, the head polled inside the step
    lat = []
    for sl in slots[n_warm + n_timed:]:
        t = time.perf_counter()
        e.on_tick(sl["S"] * 12)
        if sl["rotate"]:
            e.participation_rotate()
        with e.pipeline(lagged=False):
            agg = e.aggregate(packed=(sl["rows_in"], sl["arena_in"]), want_aggregate_pubkeys=True)
            status, _, count = e.on_attestation_batch(packed=(ROWS_RESIDENT, RESIDENT), cap=sl["cap"])
            head = e.get_head()
            t_head = time.perf_counter()
            st2, num = e.process_attestation_batch(sl["ctx"], packed=(ROWS_RESIDENT, RESIDENT), cap=sl["cap"])
        got.append(dict(agg=agg, status=status, count=count, pstatus=st2, numerators=num, head=head))
        lat.append(((t_head - t) * 1e6, (time.perf_counter() - t) * 1e6))
    for r in got:
        r["head"] = bytes(r["head"])
    # on_attestation for ONE attestation (pe:963: the reference's handler takes them one at a time): a batch of one host row,
    # synchronous -- rows of a slot already applied (the same latest messages again: nothing changes in the store)
    single = []
    one = slots[-2]
    for i in range(min(100, len(one["atts"]))):
        t = time.perf_counter()
        st1, _, _ = e.on_attestation_batch(packed=(one["atts"][i:i + 1], one["arena"]))
        single.append((time.perf_counter() - t) * 1e6)
        assert int(st1[0]) == 0
    single.sort()
    e.close()
    # every slot-step again: synchronous calls over host rows on a fresh engine
    e2 = make_engine()
    same = [step_digest(step(e2, sl, False, False, True)) == step_digest(r) for sl, r in zip(slots, got)]
    e2.close()
    per = np.diff(np.array(stamps)) * 1e6
    to_head = np.sort(np.array([a for a, _ in lat]))
    whole = np.sort(np.array([b for _, b in lat]))
    rows_per_slot = int(np.mean([len(sl["atts"]) for sl in slots]))
    out = {
        "workload": (f"{n_timed} slot-steps ({n_epochs - 1} epochs x {spe}) after {n_warm} warm-up ones: per slot on_tick + "
                     f"{rows_per_slot} partial aggregates of {slots[0]['cap']} committees -> pe_aggregate (union + aggregate "
                     "pubkeys) -> pe_on_attestation_batch -> pe_get_head -> pe_process_attestation_batch; participation "
                     "rotated and the next epoch's committees shuffled (pe_compute_committees_async) once per epoch; rows + "
                     "bits resident in HBM, streaming pipelines"),
        "attestations_per_s": n_att / dt,
        "slot_step_us_mean": dt / n_timed * 1e6,
        "slot_step_us_p50": float(np.median(per)), "slot_step_us_p99": float(np.percentile(per, 99)),
        "slot_step_detail": "host stamps around each streaming slot-step (the host runs `lag` steps ahead of the device); "
                            "the mean includes the final drain",
        "tick_to_head_us_p50": float(to_head[len(to_head) // 2]), "tick_to_head_us_p99": float(to_head[-1]),
        "tick_to_all_outputs_us_p50": float(whole[len(whole) // 2]),
        "latency_detail": f"{len(lat)} further slots, one wait per slot: on_tick -> aggregate -> on_attestation -> pe_get_head "
                          "returns the slot's head (polled) -> process_attestation -> pe_pipeline_end",
        "on_attestation_single_us_p50": float(single[len(single) // 2]),
        "on_attestation_single_detail": "pe_on_attestation_batch with ONE attestation in host memory, synchronous (validate, "
                                        "upload, LMD update, wait): what forkchoice.on_attestation costs per call",
        "fraction_of_the_slot": dt / n_timed / 12.0,
        "slot_steps_verified": int(sum(same[n_warm:])), "slot_steps": len(slots) - n_warm,
        "warmup_slot_steps_verified": int(sum(same[:n_warm])),
    }
    assert all(same), f"slot-steps differ from their synchronous replay: {[i for i, x in enumerate(same) if not x][:8]}"
    return out

