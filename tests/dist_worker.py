"""Worker of tests/test_gpu_dist_custom.py: one rank of a job whose ranks SHARE one GPU.  The engine-owned sharded step
(pe_aggregate_sharded over rows in device memory -> on_attestation -> pe_get_head_sharded -> process_attestation, inside
streaming pipelines) runs over the caller's collectives (pe_dist_init_custom: gloo, staged through the host); every rank
compares every step, bit for bit, with an UNSHARDED twin engine over the whole registry.

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tests/dist_worker.py [host|device] [lagged|plain]
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rows_mode = sys.argv[1] if len(sys.argv) > 1 else "device"
    pipe_mode = sys.argv[2] if len(sys.argv) > 2 else "lagged"
    import torch
    import torch.distributed as dist

    import pos_evolution_amd as pea
    import pos_evolution_amd.synth as synth
    from pos_evolution_amd import RESIDENT, ROWS_RESIDENT, DeviceArena, DeviceRows
    from pos_evolution_amd._abi import pe_state_ctx
    from pos_evolution_amd.sharded import HostStagedCollectives
    from tests import helpers as H
    from tests.test_gpu_sharded import _local_attestations, _local_committees

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)  # every rank on the same device
    dist.init_process_group(backend="gloo")

    V, C, spe, steps = 30011, 64, 32, 5           # an odd registry: shards of unequal size, ragged local committees
    bounds = [r * V // world + (7 if 0 < r < world else 0) for r in range(world + 1)]
    lo, hi = bounds[rank], bounds[rank + 1]
    tree = synth.random_tree(200, 9, "bushy")
    bal = synth.balances(V, 9, mixed=True)
    flags = synth.validator_flags(V, 9, inactive_frac=0.01)
    whole = pea.Engine(device=0, max_committee_tables=steps + 1)
    shard = pea.Engine(device=0, max_committee_tables=steps + 1)
    pts, _ = H.oracle_points(V)
    H.load_tree(whole, tree)
    H.load_tree(shard, tree)
    whole.set_validators(bal, flags, pts)
    shard.set_validators(bal[lo:hi], flags[lo:hi], pts[lo:hi])
    boost = tree.roots[tree.roots.shape[0] - 1].tobytes()
    whole.set_proposer_boost(boost)
    shard.set_proposer_boost(boost)   # the boost needs the GLOBAL active balance: carried by the all-reduce

    coll = HostStagedCollectives()
    if len(sys.argv) > 3 and sys.argv[3].startswith("stubrccl"):
        return stub_rccl_sharded(shard, whole, rank, world, sys.argv[3], dict(V=V, C=C, spe=spe, steps=steps + 2, tree=tree, lo=lo, hi=hi))
    if rows_mode == "committee":
        return committee_sharded(shard, whole, coll, rank, world, pipe_mode, dict(V=V, C=C, spe=spe, steps=steps, tree=tree,
                                                                                bal=bal, flags=flags, pts=pts))
    shard.dist_init_custom(rank, world, coll.all_reduce_u64, coll.all_gather)
    shard.dist_set_max_groups(C)
    ep0 = int(tree.slot.max()) // spe + 1
    keep = []
    for s in range(steps):
        ep = ep0 + s
        comm = synth.random_committees(V, C, 100 + s)
        lc = _local_committees(comm, lo, hi)
        whole.set_committees(ep, comm.offsets, comm.members)
        shard.set_committees(ep, lc.offsets, lc.members)
        atts, arena, bit_rows = synth.epoch_attestations(comm, tree, ep, spe, seed=s, density=0.9, parts=2,
                                                         source=(0, tree.roots[0].tobytes()), vote_recent=32)
        la, larena = _local_attestations(atts, bit_rows, comm, lo, hi)
        ctx = pe_state_ctx()
        ctx.slot = (ep + 1) * spe
        ctx.chain_tip_root[:] = tree.roots[tree.roots.shape[0] - 1].tobytes()
        ctx.current_justified_root[:] = tree.roots[0].tobytes()
        ctx.previous_justified_root[:] = tree.roots[0].tobytes()
        ctx.base_reward_per_increment = 777
        # the unsharded twin, synchronous host-row calls
        for e in (whole, shard):
            e.on_tick((ep + 1) * spe * 12)
            e.participation_rotate()
        ref = whole.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
        st, _, cnt = whole.on_attestation_batch(packed=(ref["atts"], ref["out_arena"]))
        ref_head, ref_w = whole.get_head(), whole.get_weights()
        pst, num = whole.process_attestation_batch(ctx, packed=(ref["atts"], ref["out_arena"]))
        # the sharded step
        if rows_mode == "device":
            r = torch.from_numpy(la.view(np.uint8).reshape(-1)).cuda()
            b = torch.from_numpy(larena).cuda()
            packed = (DeviceRows(r.data_ptr(), len(la), keep=r), DeviceArena(b.data_ptr(), b.numel(), keep=b))
            keep.append((r, b))
        else:
            packed = (la, larena)
        if pipe_mode == "plain":
            agg = shard.aggregate_sharded(packed=packed)
            hrows = (ROWS_RESIDENT, RESIDENT) if rows_mode == "device" else (agg["atts"], agg["out_arena"])
            kw = dict(cap=C) if rows_mode == "device" else {}
            lst, _, lcnt = shard.on_attestation_batch(packed=hrows, **kw)
            head = shard.get_head_sharded()
            lpst, lnum = shard.process_attestation_batch(ctx, packed=hrows, **kw)
        else:
            with shard.pipeline(lagged=(pipe_mode == "lagged")):
                agg = shard.aggregate_sharded(packed=packed)
                hrows = (ROWS_RESIDENT, RESIDENT) if rows_mode == "device" else (agg["atts"], RESIDENT)
                kw = dict(cap=C) if rows_mode == "device" else {}
                lst, _, lcnt = shard.on_attestation_batch(packed=hrows, **kw)
                # lagged: the root arrives with the pipeline's outputs (pe_get_head_sharded_async), as bench.py's N > 1 step has it
                head = shard.get_head_sharded_async() if pipe_mode == "lagged" else shard.get_head_sharded()
                lpst, lnum = shard.process_attestation_batch(ctx, packed=hrows, **kw)
            shard.drain()   # compare step by step
            head = bytes(head)
        g = ref["n_groups"]
        assert agg["n_groups"] == g, (agg["n_groups"], g)
        assert head == ref_head, f"step {s}: heads differ"
        assert np.array_equal(shard.last_weights(), ref_w), f"step {s}: reduced weights differ from the unsharded ones"
        got_pk = np.asarray(agg["aggpk96"])[:g]
        bad = np.nonzero((got_pk != ref["aggpk96"]).any(axis=1))[0]
        assert bad.size == 0, f"step {s}: aggregate pubkeys differ in {bad.size} of {g} groups, first {bad[:8]}"
        assert (st == 0).all() and (np.asarray(lst)[:g] == 0).all() and (np.asarray(lpst)[:g] == 0).all()
        # the local unions are the global ones restricted to this rank's members; local counts add up over ranks
        cps = C // spe
        for k in range(g):
            a = ref["atts"][k]
            c = int((a["slot"] % spe) * cps + a["index"])
            m = comm.members[comm.offsets[c]:comm.offsets[c + 1]]
            sel = (m >= lo) & (m < hi)
            assert np.array_equal(np.asarray(agg["bits"][k]), np.asarray(ref["bits"][k])[sel]), f"step {s} group {k}: union"
        tot = torch.from_numpy(np.asarray(lcnt)[:g].astype(np.int64))
        dist.all_reduce(tot)
        assert np.array_equal(tot.numpy(), cnt[:g].astype(np.int64)), f"step {s}: counts"
        numt = torch.from_numpy(np.asarray(lnum)[:g].astype(np.int64))
        dist.all_reduce(numt)
        assert np.array_equal(numt.numpy(), num[:g].astype(np.int64)), f"step {s}: reward numerators"
        # shard-local state = the slice of the global one
        assert np.array_equal(shard.latest_messages()[1], whole.latest_messages()[1][lo:hi]), f"step {s}: latest messages"
        assert np.array_equal(shard.participation_get(0), whole.participation_get(0)[lo:hi]), f"step {s}: participation"
    assert coll.calls["all_reduce"] >= steps and coll.calls["all_gather"] == steps, coll.calls
    shard.dist_destroy()
    dist.barrier()
    digest = hashlib.sha256(ref_head).hexdigest()[:12]
    sys.stdout.write(f"DIST_WORKER_OK rank {rank} rows={rows_mode} pipe={pipe_mode} steps={steps} head={digest} "
                     f"collectives={coll.calls}\n")
    sys.stdout.flush()
    dist.destroy_process_group()


def stub_rccl_sharded(shard, whole, rank, world, mode, W):
    """The engine's OWN RCCL path (pe_dist_unique_id / pe_dist_init_ex: two communicators, or one with "stubrccl1") over
    tests/native/libstub_rccl.so (POSEVO_RCCL_PATH), streaming lagged steps over device rows with held / paired launches.
    The ranks run SKEWED: rank 1 drains after every step (its held launches and their collectives go out at once, alone),
    rank 0 keeps its pipelines in flight (its collectives go out with the next aggregate, between paired launches) -- the order
    of the collectives per communicator must be the same on both, or the synchronous stub dead-locks and times out.  Every
    step of every rank against the unsharded twin; the ranks' call logs must agree line by line."""
    import torch
    import torch.distributed as dist

    import pos_evolution_amd.synth as synth
    from pos_evolution_amd import RESIDENT, ROWS_RESIDENT, DeviceArena, DeviceRows
    from pos_evolution_amd._abi import pe_state_ctx
    from tests.test_gpu_sharded import _local_attestations, _local_committees

    V, C, spe, steps, tree, lo, hi = W["V"], W["C"], W["spe"], W["steps"], W["tree"], W["lo"], W["hi"]
    ids = [shard.dist_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    shard.dist_init_ex(ids[0], rank, world, single_comm=mode.endswith("1"))
    shard.dist_set_max_groups(C)
    ep0 = int(tree.slot.max()) // spe + 1
    keep, done = [], []
    for s in range(steps):
        ep = ep0 + s
        comm = synth.random_committees(V, C, 100 + s)
        lc = _local_committees(comm, lo, hi)
        whole.set_committees(ep, comm.offsets, comm.members)
        shard.set_committees(ep, lc.offsets, lc.members)
        atts, arena, bit_rows = synth.epoch_attestations(comm, tree, ep, spe, seed=s, density=0.9, parts=2,
                                                         source=(0, tree.roots[0].tobytes()), vote_recent=32)
        la, larena = _local_attestations(atts, bit_rows, comm, lo, hi)
        ctx = pe_state_ctx()
        ctx.slot = (ep + 1) * spe
        ctx.chain_tip_root[:] = tree.roots[tree.roots.shape[0] - 1].tobytes()
        ctx.current_justified_root[:] = tree.roots[0].tobytes()
        ctx.previous_justified_root[:] = tree.roots[0].tobytes()
        ctx.base_reward_per_increment = 777
        for e in (whole, shard):
            e.on_tick((ep + 1) * spe * 12)
            e.participation_rotate()
        ref = whole.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
        st, _, cnt = whole.on_attestation_batch(packed=(ref["atts"], ref["out_arena"]))
        ref_head, ref_w = whole.get_head(), whole.get_weights()
        whole.process_attestation_batch(ctx, packed=(ref["atts"], ref["out_arena"]))
        r = torch.from_numpy(la.view(np.uint8).reshape(-1)).cuda()
        b = torch.from_numpy(larena).cuda()
        keep.append((r, b))
        with shard.pipeline(lagged=True):
            agg = shard.aggregate_sharded(packed=(DeviceRows(r.data_ptr(), len(la), keep=r), DeviceArena(b.data_ptr(), b.numel(), keep=b)))
            lst, _, lcnt = shard.on_attestation_batch(packed=(ROWS_RESIDENT, RESIDENT), cap=C)
            head = shard.get_head_sharded_async()
            lpst, _ = shard.process_attestation_batch(ctx, packed=(ROWS_RESIDENT, RESIDENT), cap=C)
        done.append(dict(s=s, agg=agg, head=head, lst=lst, lpst=lpst, ref=ref, ref_head=ref_head, st=st))
        if rank == 1 or s == steps - 1:
            shard.drain()
            assert np.array_equal(shard.last_weights(), ref_w), f"step {s}: reduced weights differ from the unsharded ones"
            assert np.array_equal(shard.latest_messages()[1], whole.latest_messages()[1][lo:hi]), f"step {s}: latest messages"
    for d in done:
        g = d["ref"]["n_groups"]
        assert d["agg"]["n_groups"] == g and bytes(d["head"]) == d["ref_head"], f"step {d['s']}: groups / head"
        assert np.array_equal(np.asarray(d["agg"]["aggpk96"])[:g], d["ref"]["aggpk96"]), f"step {d['s']}: aggregate pubkeys"
        assert (np.asarray(d["lst"])[:g] == 0).all() and (np.asarray(d["lpst"])[:g] == 0).all() and (d["st"] == 0).all()
    shard.dist_destroy()
    dist.barrier()
    logs = [None] * world
    mine = open(os.environ["STUB_RCCL_LOG"] + f".{rank}").read().split("\n")
    dist.all_gather_object(logs, mine)
    # communicator ordinals are per process (this process also created none before): the logs must agree line by line
    assert all(lg == logs[0] for lg in logs), "the ranks issued their collectives in different order:\n" + "\n---\n".join("\n".join(x) for x in logs)
    n_ar = sum(1 for x in mine if " allreduce " in x)
    n_ag = sum(1 for x in mine if " allgather " in x)
    assert n_ar >= steps and n_ag == steps, (n_ar, n_ag)
    sys.stdout.write(f"DIST_WORKER_OK rank {rank} rows=device pipe=lagged coll={mode} steps={steps} all_reduce={n_ar} all_gather={n_ag}\n")
    sys.stdout.flush()
    dist.destroy_process_group()


def committee_sharded(shard, whole, coll, rank, world, pipe_mode, W):
    """Committee-sharded steps (pe_aggregate + pe_aggregate_exchange): every rank holds the whole registry and store, is
    handed the rows of the committees c with c % world == rank, and must end every step with the unsharded twin's store."""
    import torch
    import torch.distributed as dist

    import pos_evolution_amd.synth as synth
    from pos_evolution_amd import RESIDENT, ROWS_RESIDENT, DeviceArena, DeviceRows
    from pos_evolution_amd._abi import pe_state_ctx

    V, C, spe, steps, tree = W["V"], W["C"], W["spe"], W["steps"], W["tree"]
    shard.set_validators(W["bal"], W["flags"], W["pts"])   # the whole registry on every rank
    shard.dist_init_custom(rank, world, coll.all_reduce_u64, coll.all_gather)
    bound = C // world + 3                                 # uneven ownership: one rank serves a few committees more
    shard.dist_set_max_groups(bound)
    cps = C // spe
    ep0 = int(tree.slot.max()) // spe + 1
    keep = []
    for s in range(steps):
        ep = ep0 + s
        comm = synth.random_committees(V, C, 100 + s)
        for e in (whole, shard):
            e.set_committees(ep, comm.offsets, comm.members)
        atts, arena, bit_rows = synth.epoch_attestations(comm, tree, ep, spe, seed=s, density=0.9, parts=2,
                                                         source=(0, tree.roots[0].tobytes()), vote_recent=32)
        pos = ((atts["slot"] % spe) * cps + atts["index"]).astype(np.int64)
        own = (pos % world == rank) if s % 2 == 0 else (pos * world // C == rank)   # interleaved / contiguous ownership
        if rank == 0 and s == 1:
            own |= pos == C - 1          # ... and rank 0 serves one committee more than its share
        if rank == 1 and s == 1:
            own &= pos != C - 1
        la = atts[own].copy()
        larena, offs, nb = synth.pack_bit_rows([b for b, o in zip(bit_rows, own) if o])
        la["bits_offset"], la["n_bits"] = offs, nb
        ctx = pe_state_ctx()
        ctx.slot = (ep + 1) * spe
        ctx.chain_tip_root[:] = tree.roots[tree.roots.shape[0] - 1].tobytes()
        ctx.current_justified_root[:] = tree.roots[0].tobytes()
        ctx.previous_justified_root[:] = tree.roots[0].tobytes()
        ctx.base_reward_per_increment = 777
        for e in (whole, shard):
            e.on_tick((ep + 1) * spe * 12)
            e.participation_rotate()
        ref = whole.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
        st, _, cnt = whole.on_attestation_batch(packed=(ref["atts"], ref["out_arena"]))
        ref_head, ref_w = whole.get_head(), whole.get_weights()
        pst, num = whole.process_attestation_batch(ctx, packed=(ref["atts"], ref["out_arena"]))
        r = torch.from_numpy(la.view(np.uint8).reshape(-1)).cuda()
        b = torch.from_numpy(larena).cuda()
        keep.append((r, b))
        cap = world * bound

        def body():
            agg = shard.aggregate(packed=(DeviceRows(r.data_ptr(), len(la), keep=r), DeviceArena(b.data_ptr(), b.numel(), keep=b)),
                                  want_aggregate_pubkeys=True)
            gx = shard.aggregate_exchange(cap_groups=cap)
            lst, _, lcnt = shard.on_attestation_batch(packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
            head = shard.get_head()
            lpst, lnum = shard.process_attestation_batch(ctx, packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
            return agg, gx, lst, lcnt, head, lpst, lnum

        if pipe_mode == "plain":
            agg, gx, lst, lcnt, head, lpst, lnum = body()
        else:
            with shard.pipeline(lagged=(pipe_mode == "lagged")):
                agg, gx, lst, lcnt, head, lpst, lnum = body()
            shard.drain()
        g = ref["n_groups"]
        key = lambda a: (int(a["slot"]), int(a["index"]), a["beacon_block_root"].tobytes())
        ref_by = {key(a): k for k, a in enumerate(ref["atts"])}
        # this rank's committees: complete unions and aggregate pubkeys, no collective behind them
        assert agg["n_groups"] == len({key(a) for a in la}), (agg["n_groups"], len(la))
        for k in range(agg["n_groups"]):
            j = ref_by[key(agg["atts"][k])]
            assert np.array_equal(agg["bits"][k], ref["bits"][j]) and agg["count"][k] == ref["count"][j], f"step {s} local group {k}"
            assert np.array_equal(agg["aggpk96"][k], ref["aggpk96"][j]), f"step {s}: aggregate pubkey of local group {k}"
        # the gathered epoch: every aggregate exactly once, rank after rank
        assert gx["n_groups"] == g, (gx["n_groups"], g)
        seen = set()
        for k in range(g):
            j = ref_by[key(gx["atts"][k])]
            seen.add(j)
            assert np.array_equal(gx["bits"][k], ref["bits"][j]) and gx["count"][k] == ref["count"][j], f"step {s} gathered group {k}"
            assert lst[k] == 0 and lpst[k] == 0 and lcnt[k] == cnt[j] and lnum[k] == num[j], f"step {s} handlers, group {k}"
        assert len(seen) == g and (np.asarray(lst)[g:] == 0).all()
        # ... and the store of every rank is the unsharded one
        assert head == ref_head, f"step {s}: heads differ"
        assert np.array_equal(shard.last_weights(), ref_w), f"step {s}: weights"
        assert np.array_equal(shard.latest_messages()[1], whole.latest_messages()[1]), f"step {s}: latest messages"
        assert np.array_equal(shard.participation_get(0), whole.participation_get(0)), f"step {s}: participation"
    assert coll.calls["all_gather"] == steps and coll.calls["all_reduce"] == 0, coll.calls
    shard.dist_destroy()
    dist.barrier()
    sys.stdout.write(f"DIST_WORKER_OK rank {rank} rows=committee pipe={pipe_mode} steps={steps} head={ref_head.hex()[:12]} "
                     f"collectives={coll.calls}\n")
    sys.stdout.flush()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
