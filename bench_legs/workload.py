"""The synthetic workload of bench.py (BASELINE.json's configs) and one step of it through the C ABI."""
import functools
import os
import time

import numpy as np

def load_registry(e, w, points=True):
    """The workload's registry into an engine: balances, activity flags, pubkeys -- and the equivocation marks, which the
    store only learns through on_attester_slashing / pe_mark_equivocating (pe:1459-1461; w["flags"] carries them as bit
    0x04 for the oracle, pe_set_validators ignores that bit)."""
    e.set_validators(w["bal"], w["flags"], w["pts"] if points else None)
    if w.get("equivocating") is not None and len(w["equivocating"]):
        e.mark_equivocating(w["equivocating"])


def build_workload(e, args, rank, n_steps_total):
    import pos_evolution_amd.synth as synth

    V, B, C, spe = args.validators_local, args.blocks, args.committees, 32
    by_committee = getattr(args, "by_committee", False)
    # validator-range shards: a registry of its own per rank; committee shards: the SAME registry, tables and epoch of
    # attestations on every rank, of which the rank is handed the rows of its own committees
    seed = 4 if by_committee else 4 + rank  # config 4 of BASELINE.json
    tree = synth.random_tree(B, 4, getattr(args, "tree_kind", None) or "bushy")  # the tree is global: same on every rank
    e.store_init(0, 0, tree.roots[0].tobytes())
    for i in range(1, B):
        e.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
    bal = synth.balances(V, seed, mixed=args.mixed_balances)
    flags = synth.validator_flags(V, seed, inactive_frac=0.005)
    pts = synth.registry_points(e, V, lo=0 if by_committee else rank * V)
    # "1 % equivocating": equivocating_indices of the store (pe:897), masked out of the LMD update (pe:1438)
    # and of the weights (A.1).  The oracle reads them as flag bit 0x04.
    equivocating = None
    frac = getattr(args, "equivocating_frac", None) or 0.0
    if frac > 0:
        rng = np.random.Generator(np.random.PCG64(seed + 5000))
        equivocating = np.sort(rng.choice(V, size=int(V * frac), replace=False)).astype(np.uint64)
        flags[equivocating] |= 0x04
    w0 = dict(bal=bal, flags=flags, pts=pts, equivocating=equivocating)
    load_registry(e, w0)
    epoch0 = int(tree.slot.max()) // spe + 1
    steps = []
    is_parent = np.zeros(B, dtype=bool)
    is_parent[tree.parent[1:]] = True
    leaves = np.nonzero(~is_parent)[0]
    for s in range(n_steps_total):
        ep = epoch0 + s
        # the epoch's committees: the reference's swap-or-not shuffle (pe:495-534, 90 rounds) run on the GPU
        import hashlib
        ep_seed = hashlib.sha256(b"bench-seed" + seed.to_bytes(8, "little") + ep.to_bytes(8, "little")).digest()
        off, mem = e.compute_committees(ep, ep_seed, V, C, 90)   # every validator active: the identity index set
        comm = synth.Committees(off, mem)
        atts, arena, _ = synth.epoch_attestations(comm, tree, ep, spe, seed=seed, density=0.99, parts=args.parts,
                                                  source=(0, tree.roots[0].tobytes()), vote_recent=64,
                                                  vote_seed=4)  # committee c votes the same block on every shard
        st = dict(epoch=ep, comm=comm, atts=atts, arena=arena, ep_seed=ep_seed)
        if getattr(args, "boost", False):
            # SURVEY 8(d) "boost {unset, set on a leaf}": proposer_boost_root (pe:896) set after the step's on_tick (which
            # clears it, pe:943-944) on one of the most recent leaves, another one every step
            st["boost_idx"] = int(leaves[-1 - (s % min(8, leaves.size))])
        if by_committee:  # committees [rank * C / N, (rank + 1) * C / N) are this rank's (its attestation subnets)
            world = getattr(args, "world", 1)
            cps = C // spe
            pos = ((atts["slot"] % spe) * cps + atts["index"]).astype(np.int64)
            own = pos * world // C == rank
            st["own"] = own
            own_atts = atts[own].copy()
            n_words = (own_atts["n_bits"].astype(np.int64) + 7) // 8
            offs = np.concatenate([[0], np.cumsum(n_words)[:-1]]).astype(np.uint32)
            st["own_arena"] = np.concatenate([arena[o:o + k] for o, k in zip(own_atts["bits_offset"], n_words)])
            own_atts["bits_offset"] = offs
            st["own_atts"] = own_atts
            if getattr(args, "emulate_ranks", 0) > 1:   # every emulated rank's rows: recorded once, replayed in the timed run
                st["rank_rows"] = []
                for q in range(world):
                    sel = pos * world // C == q
                    a_q = atts[sel].copy()
                    k_q = (a_q["n_bits"].astype(np.int64) + 7) // 8
                    ar_q = np.concatenate([arena[o:o + k] for o, k in zip(a_q["bits_offset"], k_q)])
                    a_q["bits_offset"] = np.concatenate([[0], np.cumsum(k_q)[:-1]]).astype(np.uint32)
                    st["rank_rows"].append((a_q, ar_q))
        steps.append(st)
    w = dict(tree=tree, bal=bal, flags=flags, pts=pts, equivocating=equivocating, steps=steps, spe=spe,
             world=getattr(args, "world", 1))
    shuffle_from = 0 if args.with_shuffle else getattr(args, "shuffle_variant_from", n_steps_total)
    if shuffle_from < n_steps_total:
        # the NEXT epoch's committee table is shuffled inside each step (pe_compute_committees_async: same seed, same
        # table, rewritten in place -- the epoch it feeds has not started); the last step shuffles one epoch more
        for s, st in enumerate(steps):
            if s < shuffle_from:
                continue
            # one epoch of lookahead (MIN_SEED_LOOKAHEAD): step s shuffles the table of step s + 2
            nxt = (steps[s + 2] if s + 2 < len(steps) else
                   dict(epoch=st["epoch"] + 2, ep_seed=hashlib.sha256(b"tail" + bytes([s & 255])).digest()))
            st["next_shuffle"] = (nxt["epoch"], nxt["ep_seed"], V, C, 90)
    for st in steps:  # the working state's context of each step is an input like the attestations: built up front
        st["ctx"] = state_ctx(w, st["epoch"])
    if not args.host_arena:
        # the contract's headline condition: inputs resident in HBM when the timed region starts -- the aggregation bits
        # and (unless --host-rows) the attestation rows, which are then grouped, resolved and validated on the device
        # (PE_ROWS_RESIDENT).  --host-arena / --host-rows time the hand-over from host memory instead.
        import torch
        from pos_evolution_amd import DeviceArena, DeviceRows
        for st in steps:
            t = torch.from_numpy(st["own_arena"] if by_committee else st["arena"]).cuda()
            st["arena_in"] = DeviceArena(t.data_ptr(), t.numel(), keep=t)
            if not args.host_rows:
                rows = st["own_atts"] if by_committee else st["atts"]
                r = torch.from_numpy(rows.view(np.uint8).reshape(-1)).cuda()
                st["rows_in"] = DeviceRows(r.data_ptr(), len(rows), keep=r)
        torch.cuda.synchronize()
    return w


def state_ctx(w, ep):
    from pos_evolution_amd._abi import pe_state_ctx

    tree = w["tree"]
    c = pe_state_ctx()
    c.slot = (ep + 1) * w["spe"]
    c.chain_tip_root[:] = tree.roots[tree.roots.shape[0] - 1].tobytes()
    c.current_justified_root[:] = tree.roots[0].tobytes()
    c.previous_justified_root[:] = tree.roots[0].tobytes()
    c.base_reward_per_increment = 2264  # 1e9 * 64 // isqrt(32e9 * 2^20 * 0.995) for the 1M x 32 ETH registry
    return c


_BREAKDOWN = {} if os.environ.get("POSEVO_BREAKDOWN") else None


def _timed(name, fn, *a, **k):
    if _BREAKDOWN is None:
        return fn(*a, **k)
    t = time.perf_counter()
    r = fn(*a, **k)
    _BREAKDOWN[name] = _BREAKDOWN.get(name, 0.0) + time.perf_counter() - t
    return r


def run_step_single(e, w, st, pipelined=True, lagged=True, sync_head=True, sigs=None):
    """One epoch through the per-function C ABI (sigs: one compressed BLSSignature per row -> pe_aggregate_signed in
    pe_aggregate's place).  pipelined: the three batch calls enqueue and return, the aggregate's
    rows + OR-ed bits stay on the device for the two handlers (PE_BITS_RESIDENT), get_head polls its head word, and
    pe_pipeline_end waits ONCE for every output (include/posevo.h "pipelined calls").  Same results either way
    (tests/test_gpu_pipeline.py)."""
    from pos_evolution_amd import RESIDENT, ROWS_RESIDENT

    ep = st["epoch"]
    e.on_tick((ep + 1) * w["spe"] * 12)
    if "boost_idx" in st:
        e.set_proposer_boost(w["tree"].roots[st["boost_idx"]].tobytes())
    e.participation_rotate()
    aggregate = e.aggregate if sigs is None else functools.partial(e.aggregate_signed, sigs)
    if "rows_in" in st and pipelined:
        # rows + bits resident in HBM: the host enqueues a fixed sequence of launches and reads nothing of the rows
        cap = st["comm"].offsets.size - 1   # one AttestationData per committee in this workload: groups <= committees
        if "next_shuffle" in st:
            # --with-shuffle: the per-epoch swap-or-not shuffle (pe:495-534) on the clock.  The table it makes is the one
            # the NEXT step resolves its current-epoch rows against, so it goes out first: on its own stream it runs
            # beside this step's kernels
            _timed("compute_committees", e.compute_committees_async, *st["next_shuffle"])
        with e.pipeline(lagged=lagged):
            agg = _timed("aggregate", aggregate, packed=(st["rows_in"], st["arena_in"]), want_aggregate_pubkeys=True)
            status, _, count = _timed("on_attestation", e.on_attestation_batch, packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
            # the root arrives with the step's other outputs (two steps behind, like them): the loop never blocks on the
            # device inside a step; --sync-head polls for it as pe_get_head does
            head = _timed("get_head", e.get_head if sync_head else e.get_head_async)
            st2, num = _timed("process_attestation", e.process_attestation_batch, st["ctx"],
                              packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
        return dict(agg=agg, rows=None, status=status, count=count, pstatus=st2, numerators=num, head=head)
    if not pipelined:
        agg = _timed("aggregate", aggregate, packed=(st["atts"], st["arena"]), want_aggregate_pubkeys=True)
        rows = agg["atts"]
        status, _, count = _timed("on_attestation", e.on_attestation_batch, packed=(rows, agg["out_arena"]))
        st2, num = _timed("process_attestation", e.process_attestation_batch, st["ctx"],
                          packed=(rows, agg["out_arena"]))
        head = _timed("get_head", e.get_head)
        return dict(agg=agg, rows=rows, status=status, count=count, pstatus=st2, numerators=num, head=head)
    # lagged: this step's outputs are complete when the NEXT step's block exits (the last one at e.drain(), inside the
    # timed region): the G1 sums of step N run on the second stream while the host prepares step N+1
    with e.pipeline(lagged=lagged):
        agg = _timed("aggregate", aggregate, packed=(st["atts"], st.get("arena_in", st["arena"])),
                     want_aggregate_pubkeys=True)
        rows = agg["atts"]
        status, _, count = _timed("on_attestation", e.on_attestation_batch, packed=(rows, RESIDENT))
        # fork choice first (the head depends on the LMD update only), then the state transition's flag pass: the
        # step's G1 sums are launched behind k_tree, so the flag kernel and its host work overlap them
        head = _timed("get_head", e.get_head)
        st2, num = _timed("process_attestation", e.process_attestation_batch, st["ctx"], packed=(rows, RESIDENT))
    return dict(agg=agg, rows=rows, status=status, count=count, pstatus=st2, numerators=num, head=head)

