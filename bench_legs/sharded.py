"""bench.py --gpus N: the sharded steps (validator ranges, committees, the emulated ranks) and their oracle checks."""
import functools
import os
import time

import numpy as np

from .cpu import cpu_step, cpu_step_inputs
from .workload import load_registry


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
        load_registry(e2, w)
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

