"""The CPU legs of bench.py: the L1 C oracle timed on the host (`cpu_baseline`) and the L0 pyspec text at configs[0]."""
import functools
import os
import time

import numpy as np

def _flag_masks(w, st, rows):
    """get_attestation_participation_flag_indices (Appendix A.9) per aggregate row of the step, computed here in plain
    Python from the synthetic tree: the inputs of the C oracle's flag pass."""
    import pos_evolution_amd.synth as synth

    tree, spe = w["tree"], w["spe"]
    ctx = st["ctx"]
    tip = tree.roots.shape[0] - 1
    cur_epoch = int(ctx.slot) // spe
    masks = np.zeros(len(rows), dtype=np.uint8)
    which = np.zeros(len(rows), dtype=np.uint8)
    tgt_cache, head_cache = {}, {}
    for k, a in enumerate(rows):
        slot, ep = int(a["slot"]), int(a["target_epoch"])
        delay = int(ctx.slot) - slot
        if ep not in tgt_cache:
            tgt_cache[ep] = tree.roots[synth.ancestor_at(tree, tip, ep * spe)].tobytes()
        if slot not in head_cache:
            head_cache[slot] = tree.roots[synth.ancestor_at(tree, tip, slot)].tobytes()
        mt = a["target_root"].tobytes() == tgt_cache[ep]
        mh = mt and a["beacon_block_root"].tobytes() == head_cache[slot]
        masks[k] = (1 if delay <= 5 else 0) | (2 if mt and delay <= spe else 0) | (4 if mh and delay == 1 else 0)
        which[k] = 0 if ep == cur_epoch else 1
    return masks, which


def cpu_step_inputs(w, st):
    """Flat arrays of one step for the C oracle (built once, outside every timed region)."""
    tree, comm, atts = w["tree"], st["comm"], st["atts"]
    spe = w["spe"]
    n_comm = comm.offsets.size - 1
    cps = n_comm // spe
    pos = ((atts["slot"] % spe) * cps + atts["index"]).astype(np.int64)
    order = np.argsort(pos, kind="stable")
    group_start = np.concatenate([[0], np.cumsum(np.bincount(pos, minlength=n_comm))]).astype(np.uint32)
    sizes = (comm.offsets[1:] - comm.offsets[:-1]).astype(np.uint32)
    out_off = np.concatenate([[0], np.cumsum((sizes + 7) // 8)]).astype(np.uint32)
    first = order[group_start[:-1]]
    root_idx = {tree.roots[i].tobytes(): i for i in range(tree.roots.shape[0])}
    blk = np.array([root_idx[atts[i]["beacon_block_root"].tobytes()] for i in first], dtype=np.uint32)
    masks, which = _flag_masks(w, st, atts[first])
    return dict(n_comm=n_comm, order=order.astype(np.uint32), group_start=group_start, sizes=sizes, out_off=out_off,
                first=first, blk=blk, masks=masks, which=which, target_epoch=atts["target_epoch"][first].copy())


def cpu_step(w, st, inp, mt, vote_epoch, vote_block, epoch_bump=0):
    """One whole step on the CPU with the L1 C oracle: union + G1 sums + LMD + get_head + flags."""
    from oracle import cport

    tree, comm, arena = w["tree"], st["comm"], st["arena"]
    n_comm, sizes, out_off = inp["n_comm"], inp["sizes"], inp["out_off"]
    union, count = cport.bits_union(inp["group_start"], inp["order"], st["atts"]["bits_offset"], arena, sizes,
                                    out_off[:-1], int(out_off[-1]), mt=mt)
    aggpk = cport.g1_sum_attesters(comm.offsets[:-1], sizes, out_off[:-1], union, comm.members, w["pts"], mt=mt)
    cport.update_latest_messages(comm.offsets[:-1], sizes, out_off[:-1], inp["target_epoch"] + epoch_bump, inp["blk"],
                                 union, comm.members, w["flags"], vote_epoch, vote_block, mt=mt)
    head, weights = cport.get_head(tree.parent, np.ones(tree.parent.size, np.uint8), tree.roots, vote_block,
                                   w["bal"], w["flags"], 0, boost_idx=st.get("boost_idx", cport.NONE), mt=mt)
    pc, pp = np.zeros(w["bal"].size, dtype=np.uint8), np.zeros(w["bal"].size, dtype=np.uint8)
    num = cport.process_attestation_flags(comm.offsets[:-1], sizes, out_off[:-1], inp["masks"], inp["which"], union,
                                          comm.members, w["bal"], 10**9, int(st["ctx"].base_reward_per_increment),
                                          pc, pp, mt=mt)
    return dict(union=union, count=count, aggpk=aggpk, head=tree.roots[head].tobytes(), weights=weights,
                vote_block=vote_block.copy(), numerators=num, part_cur=pc, part_prev=pp)


def cpu_baseline(w, st, target_seconds=10.0):
    """The L1 C oracle ("port") timed on the GPU box's host: whole steps of the timed workload on ONE core and, with
    the OpenMP forms of the same loops, on ALL cores.  Bounded samples (~10 s each)."""
    from oracle import cport

    inp = cpu_step_inputs(w, st)
    V = w["bal"].size
    legs, result = {}, None
    for name, mt in (("one_core", False), ("all_cores", True)):
        if mt:
            # the cores this process may run on (a container's CPU set can be smaller than the box); the thread count is
            # calibrated on a slice of the G1 sums, the dominant part: SMT siblings and the interpreter's own thread make
            # "all logical CPUs" the slowest choice on the 256-thread hosts of this pool (profiles/r02_cpu_scaling.txt)
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            comm = st["comm"]
            sub = min(inp["n_comm"], 512)
            ones = np.full(int(inp["out_off"][sub]), 0xFF, dtype=np.uint8)
            best_t, best_dt = 1, None
            for t in sorted({max(1, avail // d) for d in (1, 2, 4, 8)} | {min(avail, 64), min(avail, 32)}):
                cport.set_threads(t)
                t0 = time.perf_counter()
                cport.g1_sum_attesters(comm.offsets[:sub], inp["sizes"][:sub], inp["out_off"][:sub], ones, comm.members,
                                       w["pts"], mt=True)
                d = time.perf_counter() - t0
                if best_dt is None or d < best_dt:
                    best_t, best_dt = t, d
            cport.set_threads(best_t)
        vote_epoch = np.zeros(V, dtype=np.uint64)
        vote_block = np.full(V, 0xFFFFFFFF, dtype=np.uint32)
        n_att, reps = 0, 0
        t0 = time.perf_counter()
        while True:
            r = cpu_step(w, st, inp, mt, vote_epoch, vote_block, epoch_bump=reps)
            n_att += int(r["count"].sum())
            reps += 1
            if result is None:
                result = r
            elif reps == 1:  # the all-cores leg's first step starts from the same empty table: same answers
                for k in ("union", "count", "aggpk", "weights", "vote_block", "numerators"):
                    assert np.array_equal(r[k], result[k]), f"all-cores oracle differs from the single-thread one: {k}"
            if time.perf_counter() - t0 > target_seconds:
                break
        dt = time.perf_counter() - t0
        legs[name] = dict(value=n_att / dt, steps=reps, seconds=dt, ms_per_step=dt / reps * 1e3,
                          cores=(cport.max_threads() if mt else 1))
    one = legs["one_core"]
    return dict(value=one["value"], unit="attestations/s", cores=1, kind="port",
                sample=f"{one['steps']} full steps (union + G1 sums + LMD + get_head + flags) of the timed workload, "
                       f"oracle/posevo_oracle.c, single thread, {one['seconds']:.1f} s",
                all_cores=dict(value=legs["all_cores"]["value"], unit="attestations/s", cores=legs["all_cores"]["cores"],
                               ms_per_step=legs["all_cores"]["ms_per_step"],
                               sample=f"{legs['all_cores']['steps']} full steps, the same loops under OpenMP "
                                      f"(po_*_mt), {legs['all_cores']['seconds']:.1f} s"),
                host_cores_available=os.cpu_count()), result


def pyspec_c1_baseline(target_seconds=8.0):
    """BASELINE configs[0]: 1 024 validators, 32 slots, one committee per slot -- the L0 oracle, i.e. the reference's
    own pyspec text (oracle/_ref) on one host core: on_attestation throughput and get_head latency."""
    from oracle import spec
    from tests.scenario import new_world, slot_committee_members

    w = new_world(1024, "mainnet")
    anchor = w.store.justified_checkpoint.root
    tip, n_att, t_att, t_head, n_head = anchor, 0, 0.0, 0.0, 0
    t_start = time.perf_counter()
    for slot in range(1, 33):
        w.tick_to_slot(slot)
        tip = w.block(tip, slot)
        if slot >= 2:
            voters = slot_committee_members(w.store, slot - 1)
            atts = w.attestation_for(voters, w.store.blocks[tip].parent_root, slot - 1)
            t = time.perf_counter()
            for a in atts:
                spec.on_attestation(w.store, a)
            t_att += time.perf_counter() - t
            n_att += len(voters)
        t = time.perf_counter()
        spec.get_head(w.store)
        t_head += time.perf_counter() - t
        n_head += 1
        if time.perf_counter() - t_start > target_seconds and n_head >= 4:
            break
    spec.use_preset("mainnet")
    return dict(config="BASELINE configs[0]: 1024 validators, 32 slots, one committee of 32 per slot",
                oracle=f"L0 = the reference's pyspec text ({spec.ORACLE_OF_RECORD})", cores=1,
                on_attestation_attestations_per_s=(n_att / t_att) if t_att else None,
                get_head_ms=t_head / n_head * 1e3, slots_run=n_head)

