"""What bench.py checks after the clock has stopped: step 0 against the oracle, every step against its synchronous replay."""
import functools
import os
import time

import numpy as np

from .workload import load_registry, run_step_single


def whole_step_check(pea, w, st, chk, device):
    """Step 0 through a fresh engine (pipelined + resident, as timed) against the oracle's answers for the same step:
    union bits, counts, every aggregate pubkey, the LMD table, the head, all per-block weights, the reward numerators
    and both participation arrays."""
    e2 = pea.Engine(device=device)
    tree = w["tree"]
    e2.store_init(0, 0, tree.roots[0].tobytes())
    for i in range(1, tree.roots.shape[0]):
        e2.add_block(tree.roots[i].tobytes(), tree.roots[int(tree.parent[i])].tobytes(), int(tree.slot[i]))
    load_registry(e2, w)
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
    load_registry(e2, w)
    same = []
    for s, st in enumerate(w["steps"][:total]):
        e2.set_committees(st["epoch"], st["comm"].offsets, st["comm"].members)
        host_st = {k: v for k, v in st.items() if k not in ("rows_in", "arena_in")}
        r = run_step_single(e2, w, host_st, pipelined=False)
        same.append(step_digest(r) == step_digest(results[s]))
    e2.close()
    return same

