#!/usr/bin/env python
"""bench.py -- attestations aggregated/sec + get_head() p50 latency at 1M validators (BASELINE.json metric).

A "step" = one epoch's pass of the hot path over one batch of synthetic input, per GPU:
    4 partial aggregates per committee (8192 rows) --pe_aggregate--> 2048 aggregates: bitfield union +
    aggregate pubkey (BLS12-381 G1 sum of ~1M validator points)        [A1, A2, A3]
    --pe_on_attestation_batch--> LMD latest-message update (~1M)       [A4, A5]
    --pe_get_head--> weights from the 1M-entry vote table + descent    [H1-H6]
    --pe_process_attestation_batch--> participation flags + numerators [A6]
Every step targets a new epoch (fresh committee table, fresh votes, rotated participation), so no work is
cached or skipped.  Inputs (registry, tree, committee tables) are resident in HBM before the timed region;
the per-step attestation rows + bits (~1.7 MB) cross PCIe inside it, as the C ABI hands over host buffers.
One GPU: the calls go through the pipelined C ABI (include/posevo.h): the aggregate's rows + bits stay on the
device for the two handlers, and a step's G1 sums run on their own streams while the host prepares the next
step; every step's outputs are complete (e.drain()) before the timed region closes.  --no-lag / --no-pipeline
time the same step with one wait per step / per call.

N > 1 (launched by torch.distributed.run): validators are range-sharded -- strong scaling by default (BASELINE
configs[3]: the 1 048 576-validator registry over N GPUs), --scaling weak for a full registry per GPU; the
exchange steps are one RCCL all-gather of G1 XYZZ partials (192 B per committee) and one all-reduce of per-block weights.

Prints ONE JSON line on rank 0.
"""
import argparse
import functools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guide: ~6290 GB/s achievable)


def build_workload(e, args, rank, n_steps_total):
    import pos_evolution_amd.synth as synth

    V, B, C, spe = args.validators_local, args.blocks, args.committees, 32
    by_committee = getattr(args, "by_committee", False)
    # validator-range shards: a registry of its own per rank; committee shards: the SAME registry, tables and epoch of
    # attestations on every rank, of which the rank is handed the rows of its own committees
    seed = 4 if by_committee else 4 + rank  # config 4 of BASELINE.json
    tree = synth.random_tree(B, 4, "bushy")  # the tree is global: same on every rank
    e.store_init(0, 0, tree.roots[0].tobytes())
    for i in range(1, B):
        e.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
    bal = synth.balances(V, seed, mixed=args.mixed_balances)
    flags = synth.validator_flags(V, seed, inactive_frac=0.005)
    pts = synth.registry_points(e, V, lo=0 if by_committee else rank * V)
    e.set_validators(bal, flags, pts)
    epoch0 = int(tree.slot.max()) // spe + 1
    steps = []
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
    w = dict(tree=tree, bal=bal, flags=flags, pts=pts, steps=steps, spe=spe, world=getattr(args, "world", 1))
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


def run_step_sharded_pipelined(e, w, st, lagged=True):
    """The sharded step through the engine's own RCCL communicator (pe_dist_init): kernels, the all-gather of the G1
    partials and the all-reduce of the vote weights are enqueued on the engine's stream, the unions are handed on
    resident, and the host waits once per step (two steps behind when lagged)."""
    from pos_evolution_amd import RESIDENT, ROWS_RESIDENT

    ep = st["epoch"]
    e.on_tick((ep + 1) * w["spe"] * 12)
    e.participation_rotate()
    if "rows_in" in st:  # rows + bits resident in HBM: grouped, resolved and validated on the device, as on one GPU
        cap = len(st["comm"].offsets) - 1
        with e.pipeline(lagged=lagged):
            agg = e.aggregate_sharded(packed=(st["rows_in"], st["arena_in"]))         # all-gather of C x 192 B partials inside
            status, _, count = e.on_attestation_batch(packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
            head = e.get_head_sharded_async()                                         # all-reduce of (B + 512) x 8 B inside
            st2, num = e.process_attestation_batch(st["ctx"], packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
        return dict(agg=agg, rows=None, status=status, count=count, pstatus=st2, numerators=num, head=head)
    with e.pipeline(lagged=lagged):
        agg = e.aggregate_sharded(packed=(st["atts"], st.get("arena_in", st["arena"])))   # all-gather of C x 192 B partials inside
        rows = agg["atts"]
        status, _, count = e.on_attestation_batch(packed=(rows, RESIDENT))
        head = e.get_head_sharded()                                   # all-reduce of (B + 512) x 8 B inside
        st2, num = e.process_attestation_batch(st["ctx"], packed=(rows, RESIDENT))
    return dict(agg=agg, rows=rows, status=status, count=count, pstatus=st2, numerators=num, head=head)


class _SoloDist:
    """torch.distributed's all_gather_object for a job of one process (the emulated-ranks run checks its one real rank)."""

    @staticmethod
    def all_gather_object(out, obj):
        for i in range(len(out)):
            out[i] = obj


class ReplayCollectives:
    """pe_dist_init_custom callbacks for `bench.py --emulate-ranks N`: ONE process and one GPU carry the per-rank load of an
    N-rank committee-sharded job.  record(): a second engine runs pe_aggregate + pe_aggregate_exchange over every emulated
    rank's rows of every step and the packed aggregates each rank would send are kept in HBM.  In the timed run the
    all-gather is a device-to-device copy of that step's recording (+ the live buffer of rank 0): the exchange costs what a
    copy costs, everything else -- this rank's aggregation, the ingestion of all ranks' aggregates, the handlers over the
    whole epoch, the head -- is the real work of one rank."""

    def __init__(self, world):
        import ctypes as C

        self.C, self.world = C, world
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        self.saved = {}       # step -> torch uint8 tensor of world x bytes_per_rank
        self.mode, self.step, self.rank = "record", 0, 0

    def all_reduce_u64(self, buf, count, stream):
        return 1   # a committee-sharded step has no all-reduce

    def all_gather(self, send, recv, nbytes, stream):
        import torch

        if self.mode == "record":
            t = self.saved.get(self.step)
            if t is None:
                t = self.saved[self.step] = torch.zeros(self.world * nbytes, dtype=torch.uint8, device="cuda")
            rc = self.hip.hipMemcpyAsync(t.data_ptr() + self.rank * nbytes, send, nbytes, 3, stream)
            rc |= self.hip.hipMemsetAsync(recv, 0, nbytes * self.world, stream)   # nothing is ingested while recording
            return rc
        t = self.saved[self.step]
        assert t.numel() == self.world * nbytes
        rc = self.hip.hipMemcpyAsync(recv, t.data_ptr(), nbytes * self.world, 3, stream)
        rc |= self.hip.hipMemcpyAsync(recv, send, nbytes, 3, stream)            # rank 0's slot: what it packed just now
        return rc

    def record(self, pea, args, w, device, cap):
        import torch
        from pos_evolution_amd import DeviceArena, DeviceRows

        tree = w["tree"]
        e2 = pea.Engine(device=device, max_committee_tables=len(w["steps"]) + 2)
        e2.store_init(0, 0, tree.roots[0].tobytes())
        for i in range(1, tree.roots.shape[0]):
            e2.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
        e2.set_validators(w["bal"], w["flags"], w["pts"])
        e2.dist_init_custom(0, self.world, self.all_reduce_u64, self.all_gather)
        e2.dist_set_max_groups((args.committees + self.world - 1) // self.world + 8)
        self.mode = "record"
        for s, st in enumerate(w["steps"]):
            e2.set_committees(st["epoch"], st["comm"].offsets, st["comm"].members)
            e2.on_tick((st["epoch"] + 1) * w["spe"] * 12)
            self.step = s
            for q, (a_q, ar_q) in enumerate(st["rank_rows"]):
                self.rank = q
                r = torch.from_numpy(a_q.view(np.uint8).reshape(-1)).cuda()
                b = torch.from_numpy(ar_q).cuda()
                e2.aggregate(packed=(DeviceRows(r.data_ptr(), len(a_q), keep=r), DeviceArena(b.data_ptr(), b.numel(), keep=b)))
                e2.aggregate_exchange(cap_groups=cap)
            del st["rank_rows"]
        e2.dist_destroy()
        e2.close()
        torch.cuda.synchronize()
        self.mode, self.rank = "replay", 0


class _Lazy:
    """An array that exists when it is first used (outputs of a lagged pipeline are sliced by a count that is itself an output)."""

    def __init__(self, fn):
        self.fn = fn

    def sum(self):
        return np.asarray(self.fn()).sum()


def run_step_committee(e, w, st, lagged=True):
    """The committee-sharded step (SURVEY.md 8e Option B): pe_aggregate over this rank's committees (unions + aggregate
    pubkeys, no G1 collective) -> pe_aggregate_exchange (one all-gather of the aggregates) -> the handlers over the whole
    epoch on this rank's full copy of the store -> the plain get_head."""
    from pos_evolution_amd import RESIDENT, ROWS_RESIDENT

    ep = st["epoch"]
    e.on_tick((ep + 1) * w["spe"] * 12)
    e.participation_rotate()
    cap = len(st["comm"].offsets) - 1 + 8 * w.get("world", 1)
    with e.pipeline(lagged=lagged):
        agg = e.aggregate(packed=(st["rows_in"], st["arena_in"]), want_aggregate_pubkeys=True)
        gx = e.aggregate_exchange(cap_groups=cap)
        status, _, count = e.on_attestation_batch(packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
        head = e.get_head_async()
        st2, num = e.process_attestation_batch(st["ctx"], packed=(ROWS_RESIDENT, RESIDENT), cap=cap)
    # "count" is read when the step has completed: this rank's own aggregates (the ranks' sums add up to the epoch)
    return dict(agg=agg, gx=gx, rows=None, status=status, count=_Lazy(lambda: agg["count"]), count_all=count, pstatus=st2,
                numerators=num, head=head)


def run_step_sharded(e, w, st, sh):
    """The sharded step with torch.distributed carrying the two collectives (synchronous calls)."""
    ep = st["epoch"]
    e.on_tick((ep + 1) * w["spe"] * 12)
    e.participation_rotate()
    agg = sh.aggregate(packed=(st["atts"], st["arena"]))    # all-gather of C x 192 B XYZZ partials inside
    rows = agg["atts"]
    status, _, count = e.on_attestation_batch(packed=(rows, agg["out_arena"]))
    st2, num = e.process_attestation_batch(st["ctx"], packed=(rows, agg["out_arena"]))
    head = sh.get_head()                                    # all-reduce of (B + 512) x 8 B inside
    return dict(agg=agg, rows=rows, status=status, count=count, pstatus=st2, numerators=num, head=head)


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
                                   w["bal"], w["flags"], 0, mt=mt)
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


def whole_step_check(pea, w, st, chk, device):
    """Step 0 through a fresh engine (pipelined + resident, as timed) against the oracle's answers for the same step:
    union bits, counts, every aggregate pubkey, the LMD table, the head, all per-block weights, the reward numerators
    and both participation arrays."""
    e2 = pea.Engine(device=device)
    tree = w["tree"]
    e2.store_init(0, 0, tree.roots[0].tobytes())
    for i in range(1, tree.roots.shape[0]):
        e2.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
    e2.set_validators(w["bal"], w["flags"], w["pts"])
    e2.set_committees(st["epoch"], st["comm"].offsets, st["comm"].members)
    r = run_step_single(e2, w, st, pipelined=True, lagged=True)
    e2.drain()
    rows = r["agg"]["atts"]
    C = st["comm"].offsets.size - 1
    cps = C // w["spe"]
    pos = ((rows["slot"] % w["spe"]) * cps + rows["index"]).astype(np.int64)   # committee id of every aggregate row
    inv = np.argsort(pos)                                                       # oracle arrays are in committee order
    assert np.array_equal(pos[inv], np.arange(C)), "one aggregate per committee expected"
    out = {}
    agg = r["agg"]
    union_e = np.concatenate([np.packbits(agg["bits"][g], bitorder="little") for g in inv])
    out["union_bits"] = bool(np.array_equal(union_e, chk["union"]))
    out["counts"] = bool(np.array_equal(agg["count"][inv], chk["count"]) and np.array_equal(r["count"][:C][inv], chk["count"]))
    out["aggregate_pubkeys"] = bool(np.array_equal(agg["aggpk96"][inv], chk["aggpk"]))
    out["latest_messages"] = bool(np.array_equal(e2.latest_messages()[1], chk["vote_block"]))
    out["head"] = bytes(r["head"]) == chk["head"]
    out["weights"] = bool(np.array_equal(e2.get_weights(), chk["weights"]))
    out["reward_numerators"] = bool(np.array_equal(r["numerators"][:C][inv], chk["numerators"]))
    out["participation"] = bool(np.array_equal(e2.participation_get(0), chk["part_cur"]) and
                                np.array_equal(e2.participation_get(1), chk["part_prev"]))
    out["statuses_ok"] = bool((r["status"] == 0).all() and (r["pstatus"] == 0).all())
    e2.close()
    return out


def sharded_step_check(e, w, st, r, rank, world, dist, args):
    """The FIRST step of an N > 1 run (fresh store on every rank) against the oracle, before the clock starts.
    Rank-local (this rank's shard against the C oracle): union bits, counts, the LMD table, reward numerators, both
    participation arrays.  Global (computed on every rank from the gathered shards' oracle results, so that every rank
    also checks what the exchange delivered to IT): the head and all per-block weights against cport.get_head over the
    concatenated vote tables / balances / flags, and every aggregate pubkey against the closed form of the synthetic
    registry (P_v = A + v * B  =>  sum over S = |S| * A + (sum of S) * B: ranks exchange counts and index sums, no
    million-point CPU sum is needed).  -> dict of booleans, AND-ed over ranks."""
    import pos_evolution_amd.synth as synth
    from oracle import cport

    tree, comm, arena, spe = w["tree"], st["comm"], st["arena"], w["spe"]
    inp = cpu_step_inputs(w, st)
    V = w["bal"].size
    sizes, out_off, n_comm = inp["sizes"], inp["out_off"], inp["n_comm"]
    union, count = cport.bits_union(inp["group_start"], inp["order"], st["atts"]["bits_offset"], arena, sizes,
                                    out_off[:-1], int(out_off[-1]), mt=True)
    vote_epoch = np.zeros(V, dtype=np.uint64)
    vote_block = np.full(V, 0xFFFFFFFF, dtype=np.uint32)
    cport.update_latest_messages(comm.offsets[:-1], sizes, out_off[:-1], inp["target_epoch"], inp["blk"], union,
                                 comm.members, w["flags"], vote_epoch, vote_block, mt=True)
    pc, pp = np.zeros(V, dtype=np.uint8), np.zeros(V, dtype=np.uint8)
    num = cport.process_attestation_flags(comm.offsets[:-1], sizes, out_off[:-1], inp["masks"], inp["which"], union,
                                          comm.members, w["bal"], 10**9, int(st["ctx"].base_reward_per_increment),
                                          pc, pp, mt=True)
    agg = r["agg"]
    g = int(agg["n_groups"])
    rows = agg["atts"][:g]
    cps = n_comm // spe
    pos = ((rows["slot"] % spe) * cps + rows["index"]).astype(np.int64)
    inv = np.argsort(pos)
    out = {"one_aggregate_per_committee": bool(g == n_comm and np.array_equal(pos[inv], np.arange(n_comm)))}
    if out["one_aggregate_per_committee"]:
        union_e = np.concatenate([np.packbits(agg["bits"][k], bitorder="little") for k in inv])
        out["union_bits"] = bool(np.array_equal(union_e, union))
        out["counts"] = bool(np.array_equal(np.asarray(agg["count"])[:g][inv], count) and
                             np.array_equal(np.asarray(r["count"])[:g][inv], count))
        out["reward_numerators"] = bool(np.array_equal(np.asarray(r["numerators"])[:g][inv], num))
    out["latest_messages"] = bool(np.array_equal(e.latest_messages()[1], vote_block))
    out["participation"] = bool(np.array_equal(e.participation_get(0), pc) and np.array_equal(e.participation_get(1), pp))
    out["statuses_ok"] = bool((np.asarray(r["status"])[:g] == 0).all() and (np.asarray(r["pstatus"])[:g] == 0).all())
    # ---- global: per committee the number of attesters of this shard and the sum of their GLOBAL indices
    lo = rank * V
    cnt_c = np.zeros(n_comm, dtype=np.int64)
    sum_c = np.zeros(n_comm, dtype=object)
    for c in range(n_comm):
        bits = np.unpackbits(union[out_off[c]:out_off[c + 1]], bitorder="little")[:sizes[c]].astype(bool)
        m = comm.members[comm.offsets[c]:comm.offsets[c + 1]][bits]
        cnt_c[c] = m.size
        sum_c[c] = int(m.astype(np.uint64).sum()) + lo * int(m.size)
    shards = [None] * world
    dist.all_gather_object(shards, dict(cnt=cnt_c, sum=sum_c, vote_block=vote_block, bal=w["bal"], flags=w["flags"]))
    vb = np.concatenate([s_["vote_block"] for s_ in shards])
    bal = np.concatenate([s_["bal"] for s_ in shards])
    flags = np.concatenate([s_["flags"] for s_ in shards])
    head_o, weights_o = cport.get_head(tree.parent, np.ones(tree.parent.size, np.uint8), tree.roots, vb, bal, flags, 0, mt=True)
    out["head"] = bytes(r["head"]) == tree.roots[head_o].tobytes()
    out["weights"] = bool(np.array_equal(e.last_weights(), weights_o))
    if out["one_aggregate_per_committee"]:
        pk = np.asarray(agg["aggpk96"])[:g][inv]
        tot_cnt = sum(s_["cnt"] for s_ in shards)
        tot_sum = sum(s_["sum"] for s_ in shards)
        out["aggregate_pubkeys"] = all(pk[c].tobytes() == synth.registry_closed_form_cs(int(tot_cnt[c]), int(tot_sum[c]))
                                       for c in range(n_comm))
    allr = [None] * world
    dist.all_gather_object(allr, out)
    keys = set().union(*[set(o) for o in allr])
    return {k: bool(all(o.get(k, False) for o in allr)) for k in sorted(keys)}


def committee_step_check(e, w, st, r, rank, world, dist, args):
    """The first step of a committee-sharded run against the oracle: every rank holds the whole store, so every rank checks
    the WHOLE epoch's outcome on its own copy -- LMD table, head, all weights, both participation arrays, the gathered
    unions / counts / reward numerators -- plus the aggregate pubkeys and unions of the committees it served itself."""
    inp = cpu_step_inputs(w, st)
    V = w["bal"].size
    chk = cpu_step(w, st, inp, True, np.zeros(V, dtype=np.uint64), np.full(V, 0xFFFFFFFF, dtype=np.uint32))
    spe, comm = w["spe"], st["comm"]
    C = comm.offsets.size - 1
    cps = C // spe
    off = inp["out_off"]
    union_of = lambda c: np.unpackbits(chk["union"][off[c]:off[c + 1]], bitorder="little")[:inp["sizes"][c]].astype(bool)
    out = {}
    agg, gx = r["agg"], r["gx"]
    pos_own = ((agg["atts"]["slot"] % spe) * cps + agg["atts"]["index"]).astype(np.int64)
    out["own_committees"] = bool(np.array_equal(np.sort(pos_own), np.nonzero(np.arange(C) * world // C == rank)[0]))
    out["own_union_bits"] = all(np.array_equal(agg["bits"][k], union_of(c)) for k, c in enumerate(pos_own))
    out["own_counts"] = bool(np.array_equal(agg["count"], chk["count"][pos_own]))
    out["own_aggregate_pubkeys"] = bool(np.array_equal(agg["aggpk96"], chk["aggpk"][pos_own]))
    g = int(gx["n_groups"])
    pos_all = ((gx["atts"]["slot"] % spe) * cps + gx["atts"]["index"]).astype(np.int64)
    out["gathered_every_committee_once"] = bool(g == C and np.array_equal(np.sort(pos_all), np.arange(C)))
    if out["gathered_every_committee_once"]:
        out["gathered_union_bits"] = all(np.array_equal(gx["bits"][k], union_of(c)) for k, c in enumerate(pos_all))
        out["gathered_counts"] = bool(np.array_equal(gx["count"], chk["count"][pos_all]) and
                                      np.array_equal(np.asarray(r["count_all"])[:g], chk["count"][pos_all]))
        out["reward_numerators"] = bool(np.array_equal(np.asarray(r["numerators"])[:g], chk["numerators"][pos_all]))
    out["statuses_ok"] = bool((np.asarray(r["status"])[:g] == 0).all() and (np.asarray(r["pstatus"])[:g] == 0).all())
    out["latest_messages"] = bool(np.array_equal(e.latest_messages()[1], chk["vote_block"]))
    out["head"] = bytes(r["head"]) == chk["head"]
    out["weights"] = bool(np.array_equal(e.last_weights(), chk["weights"]))
    out["participation"] = bool(np.array_equal(e.participation_get(0), chk["part_cur"]) and
                                np.array_equal(e.participation_get(1), chk["part_prev"]))
    allr = [None] * world
    dist.all_gather_object(allr, out)
    keys = set().union(*[set(o) for o in allr])
    return {k: bool(all(o.get(k, False) for o in allr)) for k in sorted(keys)}


def step_digest(r):
    """sha256 over everything one step hands back: head, statuses, counts, reward numerators, the aggregate rows, the
    OR-ed bits, the aggregate pubkeys, the grouping."""
    import hashlib

    agg = r["agg"]
    g = int(agg["n_groups"])
    h = hashlib.sha256()
    h.update(bytes(r["head"]))
    for a in (r["status"][:g], r["count"][:g], r["pstatus"][:g], r["numerators"][:g], agg["atts"][:g], agg["out_arena"],
              agg["aggpk96"][:g], agg["count"][:g], agg["group_of"]):
        h.update(np.ascontiguousarray(a).tobytes())
    if "_raw" in agg and "sig96c" in agg["_raw"]:   # pe_aggregate_signed: the aggregate signatures, per-row statuses
        h.update(np.ascontiguousarray(agg["sig96c"]).tobytes())
        h.update(np.ascontiguousarray(agg["sig_status"]).tobytes())
    return h.digest()


def replay_and_verify(pea, w, device, results, total):
    """Every step of the run (warm-up included: the store state carries over) again on a fresh engine with SYNCHRONOUS
    calls over HOST rows -- the path the -m gpu tests hold against the oracle call by call -- and the digest of each
    step's outputs compared with what the timed run returned.  -> number of steps whose outputs are identical."""
    e2 = pea.Engine(device=device, max_committee_tables=total + 1)
    tree = w["tree"]
    e2.store_init(0, 0, tree.roots[0].tobytes())
    for i in range(1, tree.roots.shape[0]):
        e2.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
    e2.set_validators(w["bal"], w["flags"], w["pts"])
    same = []
    for s, st in enumerate(w["steps"][:total]):
        e2.set_committees(st["epoch"], st["comm"].offsets, st["comm"].members)
        host_st = {k: v for k, v in st.items() if k not in ("rows_in", "arena_in")}
        r = run_step_single(e2, w, host_st, pipelined=False)
        same.append(step_digest(r) == step_digest(results[s]))
    e2.close()
    return same


def signed_steps(pea, w, device, n_warm, n_timed, lag):
    """The step with the signature leg of the aggregation (pe:659, pe:717, pe:1536: bls.Aggregate over the members'
    BLSSignatures): pe_aggregate_signed in pe_aggregate's place -- one 96-byte compressed signature per partial aggregate
    (8192 a step at configs[3]), resident in HBM like the rows, decompressed on the device (one Fp2 square root each), summed
    per group and handed back compressed.  Same streaming pipelines as the headline steps, on a fresh engine; afterwards every
    step is replayed with synchronous host-row calls (digest equality, signatures and per-row statuses included) and a
    sample of step 0's aggregate signatures is held against the oracle's closed form.  -> the `with_signatures` object."""
    import torch
    from oracle import g2   # the checker of the sampled aggregate signatures
    from pos_evolution_amd import DeviceArena
    import pos_evolution_amd.synth as synth

    steps = w["steps"][:n_warm + n_timed]
    tree = w["tree"]

    def make_engine():
        e = pea.Engine(device=device, max_committee_tables=len(steps) + 2)
        e.store_init(0, 0, tree.roots[0].tobytes())
        for i in range(1, tree.roots.shape[0]):
            e.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
        e.set_validators(w["bal"], w["flags"], w["pts"])
        for st in steps:
            e.set_committees(st["epoch"], st["comm"].offsets, st["comm"].members)
        return e

    e = make_engine()
    n_rows = len(steps[0]["atts"])
    assert all(len(st["atts"]) == n_rows for st in steps)
    a, b = 0xABCDEF12345, 0x1357
    sigs = synth.signature_points(e, n_rows, a, b)            # row i signs with (a + i * b) * G2
    sig_t = torch.from_numpy(sigs.reshape(-1).copy()).cuda()
    sig_dev = DeviceArena(sig_t.data_ptr(), sig_t.numel(), keep=sig_t)
    e.set_pipeline_lag(lag)
    e.reuse_outputs(len(steps) + 2)
    got = [run_step_single(e, w, st, lagged=True, sync_head=False, sigs=sig_dev) for st in steps[:n_warm]]
    e.drain()
    e.fill_ring()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in steps[n_warm:]:
        got.append(run_step_single(e, w, st, lagged=True, sync_head=False, sigs=sig_dev))
    e.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_att = int(sum(int(np.asarray(r["count"]).sum()) for r in got[n_warm:]))
    for r in got:
        r["head"] = bytes(r["head"])
    bad_rows = int(sum(int((np.asarray(r["agg"]["sig_status"]) != 0).sum()) for r in got))
    e.close()
    # a sample of step 0's groups against the closed form (|S| a + b sum(i)) G2 of their member rows
    agg0 = got[0]["agg"]
    gof = np.asarray(agg0["group_of"])[:n_rows]
    ng = int(agg0["n_groups"])
    sample = sorted(set(int(x) for x in np.linspace(0, ng - 1, 32)))
    ok = True
    for k in sample:
        rows = np.nonzero(gof == k)[0]
        want = g2.compress(g2.mul((len(rows) * a + b * int(rows.sum())) % g2.R_ORDER, g2.G2))
        ok = ok and bytes(agg0["sig96c"][k]) == want
    # every step again: synchronous calls over host rows and host signatures on a fresh engine
    e2 = make_engine()
    same = []
    for st, r in zip(steps, got):
        host_st = {k: v for k, v in st.items() if k not in ("rows_in", "arena_in")}
        same.append(step_digest(run_step_single(e2, w, host_st, pipelined=False, sigs=sigs)) == step_digest(r))
    e2.close()
    assert ok, "aggregate signatures differ from the oracle's closed form"
    assert all(same), f"signed steps differ from their synchronous replay: {[i for i, x in enumerate(same) if not x][:8]}"
    assert bad_rows == 0
    return {
        "ms_per_step_with_signatures": dt / n_timed * 1e3,
        "attestations_per_s": n_att / dt,
        "signatures_per_step": n_rows,
        "steps": n_timed, "warmup": n_warm,
        "detail": ("pe_aggregate_signed in pe_aggregate's place: one compressed BLSSignature (96 B, resident in HBM) per partial "
                   "aggregate -> k_g2_decompress (an Fp2 square root each) -> per-group G2 sums -> compressed aggregate "
                   "signatures, on the state-transition stream beside the aggregate pubkeys and the fork choice; the rest of "
                   "the step as the headline's; streaming pipelines, drain included"),
        "steps_verified": int(sum(same[n_warm:])),
        "aggregate_signatures_checked_against_oracle": len(sample),
    }


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
        e.set_validators(w["bal"], w["flags"], w["pts"])
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
    t0 = time.perf_counter()
    stamps = [t0]
    for sl in slots[n_warm:n_warm + n_timed]:
        got.append(step(e, sl, True, True, False))
        stamps.append(time.perf_counter())
    e.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
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


# BASELINE.json configs[1..4] as written there (configs[0] is the CPU-only plumbing case: pyspec_c1 below)
SHAPES = {
    "configs1": dict(validators=1 << 16, committees=2048, blocks=512, mixed_balances=False),
    "configs2": dict(validators=1 << 18, committees=2048, blocks=4096, mixed_balances=False),
    "configs3": dict(validators=1 << 20, committees=2048, blocks=4096, mixed_balances=False),
    "configs4": dict(validators=1 << 22, committees=2048, blocks=8192, mixed_balances=True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--shape", choices=sorted(SHAPES), default="configs3",
                    help="the BASELINE.json config the job runs, as written there: configs3 (default) = configs[3], 1 048 576 "
                         "validators, 2048 committees x 512, 4096-block tree; configs4 = configs[4], 4 194 304 validators "
                         "(2048 x 2048), EIP-7251 mixed balances, 8192-block tree; configs1 / configs2 = the single-GPU "
                         "parity shapes.  With --gpus N the SAME registry is divided over the N GPUs (--scaling strong)")
    ap.add_argument("--validators", type=int, default=None,
                    help="registry size (default: the shape's): the whole job's with --scaling strong, per GPU with "
                         "--scaling weak")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default) = the named config as written: ONE registry, range-sharded over the N GPUs (V/N "
                         "validators and C committees x (V/N)/C local members per rank); weak = a registry shard of "
                         "--validators per GPU (N x V validators in all: per-GPU work fixed, the two exchange steps grow "
                         "with N) -- not a BASELINE config for N > 1")
    ap.add_argument("--blocks", type=int, default=None)
    ap.add_argument("--committees", type=int, default=None)
    ap.add_argument("--parts", type=int, default=4, help="partial aggregates per committee")
    ap.add_argument("--mixed-balances", action="store_true", default=None)
    ap.add_argument("--head-calls", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded-mode", choices=["engine", "torch", "committee"], default="engine",
                    help="N > 1: validator-range shards with the collectives issued by the engine inside streaming pipelines "
                         "(engine) or by torch.distributed between synchronous calls (torch); committee = committee shards "
                         "(SURVEY 8e Option B): the whole registry on every rank, a rank aggregates its own committees, one "
                         "all-gather of the aggregates, no G1 collective and no weight all-reduce (strong scaling of one "
                         "registry)")
    ap.add_argument("--host-arena", action="store_true",
                    help="hand the aggregation bits over from pageable host memory (PCIe-inclusive) instead of HBM")
    ap.add_argument("--host-rows", action="store_true",
                    help="attestation rows in host memory: grouped and validated by the host inside the timed step (the "
                         "round-2 path) instead of resident in HBM and handled on the device")
    ap.add_argument("--sync-head", action="store_true",
                    help="poll for every step's head inside the step (pe_get_head) instead of receiving it with the "
                         "step's other outputs (pe_get_head_async)")
    ap.add_argument("--no-verify-steps", action="store_true",
                    help="skip the replay that checks every timed step's outputs (steps_verified)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one wait per call instead of one per step (A/B of the pipelined C ABI)")
    ap.add_argument("--no-lag", action="store_true",
                    help="complete every step's outputs at the end of that step (pe_pipeline_end instead of _end_lagged)")
    ap.add_argument("--with-shuffle", action="store_true",
                    help="run the NEXT epoch's committee shuffle (pe_compute_committees_async: 90-round swap-or-not over "
                         "the registry + inverse committee map) inside every timed step, on the state-transition stream")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="ONE process / one GPU carrying the per-rank load of an N-rank committee-sharded job: the other "
                         "ranks' aggregates are replayed from a recording, the all-gather is a device-to-device copy "
                         "(ReplayCollectives).  A measurement of the per-rank step, not of a collective")
    ap.add_argument("--no-shuffle-variant", action="store_true",
                    help="skip the extra steps that report ms_per_step_with_shuffle")
    ap.add_argument("--no-signed-steps", action="store_true",
                    help="skip the steps with the signature leg (pe_aggregate_signed): the `with_signatures` object")
    ap.add_argument("--no-slot-cadence", action="store_true",
                    help="skip the per-slot run that reports `slot_cadence` (one GPU)")
    ap.add_argument("--no-oracle-check", action="store_true",
                    help="N > 1: skip the check of the run's first step against the oracle")
    ap.add_argument("--lag", type=int, default=4,
                    help="lag depth of the streaming pipelines (pe_pipeline_set_lag): a step's outputs are complete when "
                         "the lag-th next step has been enqueued")
    args = ap.parse_args()
    for key, val in SHAPES[args.shape].items():   # what the shape fixes, unless given explicitly
        if getattr(args, key) is None:
            setattr(args, key, val)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    # Dry run of the N > 1 path on a box with fewer GPUs than ranks (ranks share a device, collectives over gloo):
    #   POSEVO_DIST_BACKEND=gloo POSEVO_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2
    # It validates the sharded code path end to end; its numbers are not a scaling measurement.
    backend = os.environ.get("POSEVO_DIST_BACKEND", "nccl")
    # `backend` = what carries the ENGINE's exchange (nccl: RCCL owned by the engine; anything else: host-staged dry run).
    # torch.distributed itself only carries ids, barriers and the oracle check's gathers, so it runs over gloo: torch's own
    # NCCL process group brings high-priority streams into the process, and with those hardware queues beside the engine's
    # the next step's fork-choice chain is scheduled BEHIND the running accumulation -- 0.59 instead of 0.37 ms/step with
    # one rank over RCCL (gpurun_out/r03C).  --sharded-mode torch needs torch's NCCL for the exchange itself.
    torch_backend = os.environ.get("POSEVO_TORCH_BACKEND",
                                   backend if (args.sharded_mode == "torch" or backend != "nccl") else "gloo")
    if os.environ.get("POSEVO_SHARE_GPU"):
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("POSEVO_FORCE_DIST"):
        import torch.distributed as dist

        if torch_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=torch_backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node N"

    import pos_evolution_amd as pea

    # validators this rank owns: the whole registry on one GPU; V/N (strong) or V (weak) of it on N
    emulate = args.emulate_ranks if args.emulate_ranks > 1 else 0
    assert not (emulate and world > 1), "--emulate-ranks is a one-process run"
    args.by_committee = bool(emulate) or (args.sharded_mode == "committee" and
                                          (world > 1 or bool(os.environ.get("POSEVO_FORCE_DIST"))))
    args.world = emulate or world
    if args.by_committee:
        args.scaling = "strong"   # one registry, replicated; the epoch's committees are what is divided
    args.validators_local = (args.validators if args.by_committee else
                             args.validators // world if (world > 1 and args.scaling == "strong") else args.validators)
    assert args.validators_local % args.committees == 0, "validators per rank must be a multiple of the committee count"
    total = args.warmup + args.steps
    lag = 1 if args.no_lag else args.lag
    # the reported variant with the per-epoch shuffle on the clock: extra steps behind the timed ones (one GPU, streaming)
    n_var = 0 if (args.with_shuffle or args.no_shuffle_variant or world > 1 or args.no_pipeline or args.host_rows
                  or args.host_arena or emulate) else min(args.steps, 60)
    if emulate:
        args.no_cpu_baseline = True   # the CPU legs belong to the one-GPU line
    args.shuffle_variant_from = total
    total_all = total + n_var

    def setup(single_comm):
        """Engine + workload + (N > 1) the exchange: RCCL owned by the engine (two communicators, or one after a timeout),
        the caller's collectives staged through the host when the backend is not RCCL (dry runs on a shared GPU), or
        torch.distributed between synchronous calls (--sharded-mode torch)."""
        e = pea.Engine(device=local_rank, max_committee_tables=total_all + 3)
        w = build_workload(e, args, rank, total_all)
        ex, engine_rccl, how = None, False, None
        if emulate:
            coll = ReplayCollectives(emulate)
            cap = args.committees + 8 * emulate
            coll.record(pea, args, w, local_rank, cap)
            e.dist_init_custom(0, emulate, coll.all_reduce_u64, coll.all_gather)
            e.dist_set_max_groups((args.committees + emulate - 1) // emulate + 8)
            w["replay"] = coll
            engine_rccl = True
            ex = True
            how = (f"EMULATED: one process carries rank 0 of {emulate}; the other ranks' aggregates are replayed from a "
                   "recording and the all-gather is a device-to-device copy (bench.py ReplayCollectives)")
        if dist is not None:
            from pos_evolution_amd.sharded import HostStagedCollectives, ShardedForkChoice
            if args.sharded_mode in ("engine", "committee") and not args.no_pipeline:
                ok_t = torch.tensor([1], device="cuda" if torch_backend == "nccl" else "cpu")
                try:
                    if backend == "nccl":  # the engine's own communicators; torch.distributed only carries the 256-byte id
                        ex = ShardedForkChoice(e, n_groups_max=args.committees, use_engine_rccl=True, single_comm=single_comm)
                        how = "RCCL owned by the engine, " + ("one communicator (fallback after a timeout)" if single_comm
                                                              else "two communicators")
                    else:                  # the same engine-owned step over the caller's collectives (pe_dist_init_custom)
                        ex = ShardedForkChoice(e, n_groups_max=args.committees, use_engine_rccl=True,
                                               collectives=HostStagedCollectives())
                        how = f"pe_dist_init_custom: {backend} staged through the host (dry run, not a scaling measurement)"
                    engine_rccl = True
                    e.dist_set_max_groups((args.committees + world - 1) // world + 8 if args.by_committee else args.committees)
                except Exception as err:  # e.g. no librccl to dlopen: the torch-carried exchange does the same job
                    print(f"[bench] engine-owned exchange unavailable ({err}); using torch.distributed", file=sys.stderr)
                    ok_t[0] = 0
                dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)   # every rank takes the same path
                if int(ok_t[0]) == 0:
                    if engine_rccl:
                        e.dist_destroy()
                    ex, engine_rccl = None, False
            if ex is None:
                ex = ShardedForkChoice(e, n_groups_max=args.committees)
                how = "torch.distributed between synchronous calls"
        return e, w, ex, engine_rccl, how

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(e, w, ex, engine_rccl):
        step_no = [0]

        def step(st):
            if emulate:
                w["replay"].step = step_no[0]   # steps run in workload order: warm-up, timed, nothing else
                step_no[0] += 1
            if engine_rccl and args.by_committee:
                return run_step_committee(e, w, st, lagged=not args.no_lag)
            if engine_rccl:
                return run_step_sharded_pipelined(e, w, st, lagged=not args.no_lag)
            return run_step_sharded(e, w, st, ex) if ex else run_step_single(e, w, st, pipelined=not args.no_pipeline,
                                                                             lagged=not args.no_lag, sync_head=args.sync_head)

        verify = ex is None and not args.no_pipeline and not args.no_verify_steps
        if (ex is None or engine_rccl) and not args.no_pipeline:
            # a streaming caller reuses its output buffers (results are consumed `lag` steps behind); to verify every
            # timed step afterwards the ring is as deep as the run, allocated and touched before the clock starts
            if not args.no_lag:
                e.set_pipeline_lag(args.lag)
            e.reuse_outputs(total_all + 2 if verify else lag + 2)
        kept, sharded_chk = [], None
        import gc

        def quiet_the_host():
            # Python's cyclic collector is paused over the timed steps: with torch loaded a full collection walks ~10^6
            # objects (tens of ms) and, landing inside a 20-step window, would be charged to the engine as +2 ms per step.
            # Collected HERE -- in front of the last warm-up steps, not between them and the clock: tens of milliseconds of
            # idle GPU in front of a 6 ms timed region is a cold start, which is not what W warm-up steps are for.
            gc.collect()
            gc.disable()
            e.profile_enable(True)

        for s in range(args.warmup):
            if s == 1:
                quiet_the_host()
            kept.append(step(w["steps"][s]))
            if s == 0:
                e.drain()
                e.fill_ring()
                if emulate and not args.no_oracle_check:
                    sharded_chk = committee_step_check(e, w, w["steps"][0], kept[0], 0, emulate, _SoloDist, args)
                if dist is not None and not args.no_oracle_check:
                    # N > 1: the first step of the run (a fresh store on every rank) against the oracle, before the clock
                    sharded_chk = (committee_step_check if args.by_committee else sharded_step_check)(
                        e, w, w["steps"][0], kept[0], rank, world, dist, args)
        if args.warmup < 2:
            quiet_the_host()
        e.drain()
        e.profile_reset()   # the warm-up launches are not part of the per-kernel averages
        barrier()
        t0 = time.perf_counter()
        inflight = []
        stamps = [t0]
        n_att_local = n_rejected = 0
        for s in range(args.warmup, total):
            inflight.append(step(w["steps"][s]))
            if verify:
                kept.append(inflight[-1])
            stamps.append(time.perf_counter())
            if len(inflight) > lag:  # complete by now (a lagged step completes when the lag-th next one's block exits)
                done = inflight.pop(0)
                n_att_local += int(done["count"].sum())
                n_rejected += int((done["status"] != 0).sum()) + int((done["pstatus"] != 0).sum())
        e.drain()  # the last (lagged) steps' outputs: inside the timed region
        for done in inflight:
            n_att_local += int(done["count"].sum())
            n_rejected += int((done["status"] != 0).sum()) + int((done["pstatus"] != 0).sum())
        last = inflight[-1]
        last["head"] = bytes(last["head"])
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        prof = e.profile()
        e.profile_enable(False)
        assert n_rejected == 0, "synthetic attestations were rejected"
        dt_var = None
        if n_var:  # the same steps with the next epoch's shuffle enqueued inside each of them
            gc.collect()
            gc.disable()
            barrier()
            t1 = time.perf_counter()
            more = [step(w["steps"][s]) for s in range(total, total_all)]
            e.drain()
            barrier()
            dt_var = time.perf_counter() - t1
            gc.enable()
            assert all(int((m["status"] != 0).sum()) + int((m["pstatus"] != 0).sum()) == 0 for m in more)
            if verify:
                kept.extend(more)
        return dict(dt=dt, stamps=stamps, n_att_local=n_att_local, last=last, prof=prof, kept=kept, verify=verify,
                    sharded_chk=sharded_chk, dt_var=dt_var)

    e, w, ex, engine_rccl, exchange_how = setup(single_comm=False)
    dist_fallback = None
    try:
        R = run(e, w, ex, engine_rccl)
    except pea.EngineError as err:
        # A hung exchange surfaces as PE_ERR_TIMEOUT on every rank (the engine's bounded waits; the communicators are
        # aborted).  Start over with both collectives on ONE communicator and one stream, where they cannot be ordered
        # differently on different ranks.
        if not (engine_rccl and backend == "nccl" and err.status == pea._abi.PE_ERR_TIMEOUT):
            raise
        print(f"[bench] rank {rank}: {err}; restarting with a single communicator", file=sys.stderr)
        e.dist_destroy()
        e.close()
        dist.barrier()
        dist_fallback = "single communicator after PE_ERR_TIMEOUT with two"
        e, w, ex, engine_rccl, exchange_how = setup(single_comm=True)
        R = run(e, w, ex, engine_rccl)
    dt, stamps, n_att_local, last, prof, kept, verify = (R[k] for k in ("dt", "stamps", "n_att_local", "last", "prof",
                                                                        "kept", "verify"))

    # get_head latency: full recomputation from the vote table, after the timed region
    lat = []
    for _ in range(20):
        e.get_head() if (ex is None or args.by_committee) else ex.get_head()
    for _ in range(args.head_calls):
        t = time.perf_counter()
        e.get_head() if (ex is None or args.by_committee) else ex.get_head()
        lat.append((time.perf_counter() - t) * 1e6)
    lat = np.sort(np.array(lat))

    if dist is not None:
        t = torch.tensor([dt, float(n_att_local)], dtype=torch.float64, device="cuda" if torch_backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, n_att = float(tmax[0]), float(t[1])
        heads = [None] * world
        dist.all_gather_object(heads, last["head"])
        assert all(h == heads[0] for h in heads), "ranks disagree on the head"
    else:
        n_att = float(n_att_local)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel: k_g1_accumulate, algorithmic bytes per launch ----
    C = args.committees
    VL = args.validators_local
    att_per_launch = n_att_local / args.steps
    alg_bytes = 100.125 * att_per_launch + C * (96 + (VL / C) / 8)
    acc = prof["g1_accumulate"]
    acc_ms = acc["total_ms"] / max(acc["launches"], 1)
    achieved = alg_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms else 0.0
    # HBM bytes per launch from the PMC passes (tools/profile_round.sh): recorded per shape, quoted only for the shape
    # that was measured
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            ent = json.load(open(tpath)).get("shapes", {}).get(f"{VL},{C},{args.blocks}")
            if ent:
                traffic, traffic_src = ent.get("k_g1_accumulate_bytes_per_launch"), ent.get("source")
        except Exception:
            traffic = None
    # VALU view of the same kernel (round 4: the S29 field form, fp381_s29.h).  A lane takes its first point as it is and
    # adds the others with the general mixed add: 8 products of 392 multiply-adds + 2 squarings of 301 = 3738
    # v_mad_[iu]64_[iu]32 and nothing that carries; the hand-over to the tree's words costs 4 products per lane.  Lanes as the
    # engine plans them in streaming steps: one wave per SIMD, k = max(4, ceil(members / 65536)) members per lane.
    MACS_MUL, MACS_SQR = 392, 301
    MACS_ADD = 8 * MACS_MUL + 2 * MACS_SQR
    k_run = max(4, -(-VL // 65536))
    lane_runs = C * -(-(VL // C) // k_run)
    mixed_adds = max(att_per_launch - lane_runs, 0.0)
    macs = mixed_adds * MACS_ADD + lane_runs * 4 * MACS_MUL
    # the multiplier's issue rate (tools/ubench_valu, profiles/r01_ubench_valu_fpmul.log: v_mad_u64_u32 every 2.496 ns per SIMD
    # at two waves, 2.454 at four; v_mad_i64_i32 is the same unit): 1024 SIMDs x 64 lanes
    MAC_PEAK = 1024 * 64 / 2.496e-9
    valu_peak = MAC_PEAK / MACS_ADD              # mixed adds per second if the SIMDs issued nothing but multiply-adds
    valu_ach = mixed_adds / (acc_ms * 1e-3) if acc_ms else 0.0
    # the tree over the lanes' partials (12 x 32-bit form, 288 multiply-adds + 288 carry adds per product, 14 products per
    # add): one add per lane but one per committee; it runs on the same SIMDs beside the NEXT accumulation
    tree_macs = 14.0 * 288 * max(lane_runs - C, 0)
    votes = prof["votes"]
    votes_ms = votes["total_ms"] / max(votes["launches"], 1)
    votes_bytes = 13.0 * VL + 32.0 * args.blocks
    kernel_ms = {k: (v["total_ms"] / v["launches"] if v["launches"] else None) for k, v in prof.items()}
    per_step = np.diff(np.array(stamps)) * 1e3
    V_total = VL if args.by_committee else VL * world   # committee shards: one registry, on every rank
    named = {(1 << 16, 2048, 512, False): 1, (1 << 18, 2048, 4096, False): 2, (1 << 20, 2048, 4096, False): 3,
             (1 << 22, 2048, 8192, True): 4}.get((V_total, C, args.blocks, bool(args.mixed_balances)))
    if named and world > 1 and args.scaling == "weak":
        named = None
    shape = (f"BASELINE configs[{named}]" + (f" over {world} GPUs" if world > 1 else " on one GPU") if named else
             f"custom shape (weak scaling: a configs[3]-sized registry shard per GPU, {world} x {VL} validators)"
             if (world > 1 and args.scaling == "weak" and (VL, C, args.blocks) == (1 << 20, 2048, 4096)) else "custom shape")
    scaling = args.scaling  # strong (default): the named config's registry, whole on one GPU, divided over N
    mode = ("sharded, streaming pipelines, collectives issued by the engine between its kernels (" + exchange_how + ")"
            if engine_rccl else
            "sharded, synchronous calls, collectives through torch.distributed" if world > 1 else
            "synchronous calls" if args.no_pipeline else
            "pipelined calls (one wait per step)" if args.no_lag else
            "streaming pipelines (pe_pipeline_begin_streaming / _end_lagged: a step's G1 sums overlap the next step)")

    out = {
        "metric": "attestations aggregated/sec + get_head() p50 latency at 1M validators",
        "value": n_att / dt,
        "unit": "attestations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "int32",
        "dtype_detail": ("381-bit Fp, exact integer arithmetic: the accumulation (dominant kernel) in 14 signed 29-bit limbs "
                         "held in int32 with int64 column sums (fp381_s29.h); tree / finish in 12 x u32 Montgomery limbs; "
                         "u64 Gwei weights"),
        "data": "synthetic",
        "config": {
            "workload": shape + f": {V_total} validators on {world} GPU(s) ({VL} per GPU), "
                        f"{C} committees x {V_total // C}, {args.parts} partial aggregates/committee, "
                        f"99% participation, {args.blocks}-block tree, one epoch per step: pe_aggregate (union + "
                        f"aggregate pubkeys) -> pe_on_attestation_batch -> pe_get_head -> pe_process_attestation_batch",
            "validators_total": V_total, "validators_per_gpu": VL, "blocks": args.blocks, "committees": C,
            "parallelism": (f"committee shards x{world}: registry and store replicated, each rank aggregates C / {world} "
                            f"committees ({scaling} scaling)" if args.by_committee else
                            f"validator-range shards x{world} ({scaling} scaling)" if world > 1 else "single GPU"),
            "call_mode": mode,
            "g1_field_form": "s29 (accumulation) + 12x32 (tree, finish)",
            "inputs": (("attestation rows in host memory (grouped and validated by the host inside the timed step); "
                        if (args.host_rows or args.host_arena or (world > 1 and not engine_rccl)) else
                        "attestation rows resident in HBM before the timed region (grouped, resolved and validated on "
                        "the device: PE_ROWS_RESIDENT); ")
                       + ("aggregation bits in pageable host memory, copied over PCIe inside the timed step"
                          if args.host_arena else "aggregation bits resident in HBM before the timed region")),
            "per_epoch_setup_outside_the_step": ("nothing: the next epoch's committee shuffle (pe_compute_committees_async) "
                                                 "runs inside every step (--with-shuffle)" if args.with_shuffle else
                                                 "pe_compute_committees (GPU swap-or-not shuffle + inverse committee map) "
                                                 "runs once per epoch when the workload is built, not in the step; "
                                                 "--with-shuffle puts it on the clock"),
        },
        "step_ms_p50": float(np.median(per_step)), "step_ms_min": float(per_step.min()),
        "step_ms_p90": float(np.percentile(per_step, 90)),
        "aggregation_budget": {"seconds": 4.0, "source": "pe:1536 (the last third of a 12 s slot)",
                               "step_fraction_of_budget": dt / args.steps / 4.0},
        "get_head_p50_us": float(lat[len(lat) // 2]),
        "get_head_p99_us": float(lat[min(len(lat) - 1, int(len(lat) * 0.99))]),
        "roofline": {
            "kernel": "k_g1_accumulate", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": acc_ms, "launches": acc["launches"],
            "note": "integer-VALU bound (3738 multiply-adds per 100 B gathered), not HBM bound: "
                    "see roofline_valu and DESIGN.md; the votes kernel below is the HBM-streaming one",
        },
        "roofline_valu": {
            "kernel": "k_g1_accumulate", "bound": "integer VALU: v_mad_i64_i32 / v_mad_u64_u32 issue (S29 field form)",
            "achieved": valu_ach / 1e9, "peak": valu_peak / 1e9, "unit": "G mixed adds/s", "frac": valu_ach / valu_peak,
            "mixed_adds_per_launch": mixed_adds, "multiply_adds_per_mixed_add": MACS_ADD,
            "multiply_adds_per_launch": macs,
            "peak_source": "instruction ceiling: one 32 x 32 -> 64 multiply-add per 2.496 ns and SIMD (tools/ubench_valu, "
                           "profiles/r01_ubench_valu_fpmul.log) x 1024 SIMDs x 64 lanes / 3738 multiply-adds per mixed add = "
                           "7.0 G/s; the same adds in a loop without loads reach 6.96 G/s (tools/icbench, "
                           "profiles/r04_icbench.txt), the kernel alone 5.8 G/s (tools/accbench, profiles/r04_accbench.txt)",
            "step_view": {
                "tree_multiply_adds_per_step": tree_macs,
                "multiply_adds_per_step": macs + tree_macs,
                "frac_of_multiplier_over_the_step": (macs + tree_macs) / (dt / args.steps) / MAC_PEAK,
                "note": "multiply-adds of the accumulation + the tree of one step over the whole step period: the fraction of "
                        "the period the chip's multipliers spend on this algorithm; the rest is the tree's carry adds, the "
                        "in-situ efficiency of the two kernels and the chain of small kernels that paces the step "
                        "(DESIGN.md 8)",
            },
        },
        "roofline_votes": {
            "kernel": "k_votes", "bound": "hbm", "achieved": votes_bytes / (votes_ms * 1e-3) / 1e9 if votes_ms else 0.0,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (votes_bytes / (votes_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if votes_ms else 0.0,
            "avg_launch_ms": votes_ms, "launches": votes["launches"],
        },
        "kernel_avg_ms": kernel_ms,
    }
    if emulate:
        g0 = kept[0]["gx"]
        att_epoch = int(np.asarray(g0["count"]).sum())
        out["emulated_ranks"] = emulate
        out["emulated_job_attestations_per_s"] = att_epoch / (dt / args.steps)
        out["emulated_detail"] = (f"rank 0 of an {emulate}-rank committee-sharded job on ONE GPU: `value` counts the attestations "
                                  f"of rank 0's own {C // emulate} committees; emulated_job_attestations_per_s = the epoch's "
                                  f"{att_epoch} attestations per rank-step time, i.e. the job's rate if every rank ran this step "
                                  "concurrently and the all-gather cost what a device-to-device copy costs")
    if dist is not None:
        out["config"]["torch_distributed_backend"] = torch_backend + " (ids, barriers and the oracle check only)"
    if dist is not None or emulate:
        out["config"]["exchange"] = (("per step: ONE all-gather of the ranks' aggregate attestations (144 B data + flags, count, "
                                      f"256 B of OR-ed bits per aggregate; {(C + world - 1) // world + 8} slots per rank); no G1 "
                                      "collective, no weight all-reduce; " if args.by_committee else
                                      "per step: one all-reduce(sum) of (blocks + 512) u64 = "
                                      f"{(args.blocks + 512) * 8} B, one all-gather of {C} x 192 B XYZZ partials per rank; ")
                                     + exchange_how)
        if dist_fallback:
            out["dist_fallback"] = dist_fallback
        if R["sharded_chk"] is not None:
            out["checked_against_oracle"] = bool(all(R["sharded_chk"].values()))
            out["oracle_check"] = R["sharded_chk"]
            out["oracle_check_detail"] = ("the run's first step (fresh store) on every rank: shard-local outputs and state "
                                          "vs the C oracle, head + all weights vs cport.get_head over the gathered vote "
                                          "tables, every aggregate pubkey vs the registry's closed form; AND over ranks")
            assert out["checked_against_oracle"], f"sharded step differs from the oracle: {R['sharded_chk']}"
    if R["dt_var"] is not None:
        out["ms_per_step_with_shuffle"] = R["dt_var"] / n_var * 1e3
        out["with_shuffle_detail"] = (f"{n_var} further steps, each also enqueuing the NEXT epoch's committee shuffle "
                                      "(pe_compute_committees_async: k_shuffle_tables + k_shuffle_indices, 90 rounds over the "
                                      "registry, + the inverse committee map) on the state-transition stream; ramp and drain "
                                      "of the pipeline included")
    if verify:
        # every step's outputs (kept in the deep ring) against a synchronous host-row replay, after the clock stopped
        same = replay_and_verify(pea, w, local_rank, kept, total_all)
        out["steps_verified"] = int(sum(same[args.warmup:]))
        out["steps_verified_detail"] = ("sha256 of each timed step's outputs (head, statuses, counts, numerators, aggregate "
                                        "rows, OR-ed bits, aggregate pubkeys, grouping) == the same step replayed with "
                                        "synchronous host-row calls on a fresh engine; warm-up steps replayed too: "
                                        f"{int(sum(same[:args.warmup]))}/{args.warmup} identical")
        out["steps_verified"] = int(sum(same[args.warmup:total]))
        if n_var:
            out["steps_verified_with_shuffle"] = int(sum(same[total:]))
            assert out["steps_verified_with_shuffle"] == n_var, "with-shuffle steps differ from their synchronous replay"
        assert out["steps_verified"] == args.steps, f"timed steps differ from their synchronous replay: {same}"
    if world == 1 and not emulate and not args.no_slot_cadence and not args.no_pipeline and not args.host_rows \
            and not args.host_arena and len(w["steps"]) >= 4 and C % w["spe"] == 0:
        try:
            out["slot_cadence"] = slot_cadence(pea, w, local_rank, 3, args.lag)
        except Exception as err:   # as above
            print(f"[bench] slot_cadence failed: {err!r}", file=sys.stderr)
            out["slot_cadence"] = {"error": repr(err)}
    if world == 1 and not emulate and not args.no_signed_steps and not args.no_pipeline and not args.host_rows \
            and not args.host_arena and not args.no_lag:
        n_signed = min(20, args.steps)
        try:
            out["with_signatures"] = signed_steps(pea, w, local_rank, min(3, len(w["steps"]) - n_signed), n_signed, args.lag)
            out["ms_per_step_with_signatures"] = out["with_signatures"]["ms_per_step_with_signatures"]
        except Exception as err:   # an extra leg: reported (here and on stderr), never at the cost of the headline line
            print(f"[bench] with_signatures failed: {err!r}", file=sys.stderr)
            out["with_signatures"] = {"error": repr(err)}
    if not args.no_cpu_baseline and world == 1:
        base, chk = cpu_baseline(w, w["steps"][0])
        out["cpu_baseline"] = base
        out["cpu_baseline"]["pyspec_c1"] = pyspec_c1_baseline()
        # step 0 through a fresh engine, every output and the device state against the oracle's answers (not timed)
        chk_out = whole_step_check(pea, w, w["steps"][0], chk, local_rank)
        out["checked_against_oracle"] = bool(all(chk_out.values()))
        out["oracle_check"] = chk_out
        assert out["checked_against_oracle"], f"GPU step differs from the oracle: {chk_out}"
    if _BREAKDOWN is not None:
        out["host_breakdown_ms_per_step"] = {k: v / total * 1e3 for k, v in _BREAKDOWN.items()}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
