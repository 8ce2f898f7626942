"""bench.py `with_signatures`: the step with the signature leg of the aggregation."""
import functools
import os
import time

import numpy as np

from .verify import step_digest
from .workload import load_registry, run_step_single


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
        load_registry(e, w)
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
    # the legs of POSEVO_SIG_BATCH (8) steps share one decompression launch of ~0.95 ms: a lag depth of batch + the ~4-5 steps that
    # launch and the sums behind it last keeps the host from waiting for it (include/posevo.h: pe_pipeline_set_lag; 15 is the deepest)
    sig_lag = max(lag, 15)
    e.set_pipeline_lag(sig_lag)
    e.reuse_outputs(max(len(steps), sig_lag) + 2)
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
        "steps": n_timed, "warmup": n_warm, "lag": sig_lag, "steps_per_decompression": int(os.environ.get("POSEVO_SIG_BATCH", "8")),
        "detail": ("pe_aggregate_signed in pe_aggregate's place: one compressed BLSSignature (96 B, resident in HBM) per partial "
                   "aggregate -> k_g2_decompress (an Fp2 square root each; ONE launch for the legs of `steps_per_decompression` "
                   "consecutive steps) -> per-group G2 sums -> compressed aggregate signatures, on the signature legs' stream beside "
                   "the aggregate pubkeys and the fork choice; the rest of the step as the headline's; streaming pipelines, drain "
                   "included"),
        "steps_verified": int(sum(same[n_warm:])),
        "aggregate_signatures_checked_against_oracle": len(sample),
    }



def unaggregated_signatures(pea, w, device, n_warm=3, n_timed=4, check_groups=32):
    """The epoch's UNAGGREGATED signatures (pe:717: every attester signs; pe:474 / pe:659 / pe:1536: aggregated per committee):
    one compressed 96-byte BLSSignature per validator resident in HBM (V x 96 bytes), the attesters of every committee as an
    index list resident in HBM -> pe_aggregate_signatures: k_g2_decompress over the whole chip (an Fp2 square root per
    signature), k_g2_accumulate / k_g2_finish per committee, the 2048 aggregates handed back compressed.  Synchronous calls,
    one epoch per call; every call's aggregates are sampled against the oracle's closed form.
    -> the `with_unaggregated_signatures` object."""
    import torch
    from oracle import cport, g2
    from pos_evolution_amd import DeviceArena
    import pos_evolution_amd.synth as synth
    from .cpu import cpu_step_inputs

    # n_warm = 3: the leg starts after seconds of host-side checking with the device idle; the first calls run at 2-3 x the
    # steady time (allocations of the engine's scratch and pinned blocks, clocks coming back up): `ms_per_call` lists the timed
    # calls one by one, `warmup_ms_per_call` the warm-up ones
    steps = [w["steps"][k % len(w["steps"])] for k in range(n_warm + n_timed)]
    V = w["bal"].size
    e = pea.Engine(device=device)
    a, b, period = 0xABCDEF12345, 0x1357, 16384
    base = synth.signature_points(e, min(V, period), a, b)              # validator v signs with (a + (v mod 16384) b) G2
    sigs = np.ascontiguousarray(np.tile(base, (-(-V // base.shape[0]), 1))[:V])
    sig_t = torch.from_numpy(sigs.reshape(-1)).cuda()
    sig_dev = DeviceArena(sig_t.data_ptr(), sig_t.numel(), keep=sig_t)
    prepared = []
    for st in steps:   # who attested: the union of the committee's partial aggregates (the oracle's), as an index list per committee
        inp = cpu_step_inputs(w, st)
        union, count = cport.bits_union(inp["group_start"], inp["order"], st["atts"]["bits_offset"], st["arena"], inp["sizes"],
                                        inp["out_off"][:-1], int(inp["out_off"][-1]), mt=True)
        comm, off = st["comm"], inp["out_off"]
        lists = []
        for c in range(inp["n_comm"]):
            bits = np.unpackbits(union[off[c]:off[c + 1]], bitorder="little")[:inp["sizes"][c]].astype(bool)
            lists.append(comm.members[comm.offsets[c]:comm.offsets[c + 1]][bits])
        offsets = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint32)
        index = np.concatenate(lists).astype(np.uint32)
        idx_t = torch.from_numpy(index).cuda()
        prepared.append(dict(lists=lists, offsets=offsets, index=DeviceArena(idx_t.data_ptr(), idx_t.numel() * 4, keep=idx_t),
                             n=int(index.size)))
    torch.cuda.synchronize()
    got, warm_calls = [], []
    for p in prepared[:n_warm]:
        t1 = time.perf_counter()
        got.append(e.aggregate_signatures(sig_dev, p["offsets"], index=p["index"]))
        warm_calls.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_call = []
    for p in prepared[n_warm:]:
        t1 = time.perf_counter()
        got.append(e.aggregate_signatures(sig_dev, p["offsets"], index=p["index"]))
        per_call.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_sig = int(sum(p["n"] for p in prepared[n_warm:]))
    e.close()
    # a sample of every call's committees against the closed form (|S| a + b sum(v mod 16384)) G2
    verified = 0
    for p, (agg, status, bad) in zip(prepared, got):
        ok = not status.any() and not bad.any()
        ng = len(p["lists"])
        for c in sorted(set(int(x) for x in np.linspace(0, ng - 1, check_groups))):
            m = p["lists"][c].astype(np.int64) % period
            want = g2.compress(g2.mul((len(m) * a + b * int(m.sum())) % g2.R_ORDER, g2.G2))
            ok = ok and bytes(agg[c]) == want
        verified += int(ok)
    assert verified == len(prepared), "aggregates of the unaggregated signatures differ from the oracle's closed form"
    products = 1010.0   # two windowed Fp exponentiations of <= 484 products + ~40 around them (g2_kernels.hip, fp_sqrt.h)
    ceiling = 68.6e9    # dependent S29 products per second, chip-wide (tools/fpbench29, profiles/r04_fpbench29.txt)
    return {
        "ms_per_epoch": dt / n_timed * 1e3,
        "ms_per_call": [round(x, 2) for x in per_call],
        "warmup_ms_per_call": [round(x, 2) for x in warm_calls],
        "signatures_per_s": n_sig / dt,
        "signatures_per_epoch": n_sig // n_timed,
        "committees": len(prepared[0]["lists"]),
        "epochs": n_timed, "warmup": n_warm,
        "epochs_verified": verified - n_warm,
        "checked_per_epoch": check_groups,
        "roofline_valu": {"bound": "integer VALU (the S29 Fp product of the square roots)", "products_per_signature": products,
                          "achieved_G_products_per_s": n_sig * products / dt / 1e9, "peak_G_products_per_s": ceiling / 1e9,
                          "frac": n_sig * products / dt / ceiling},
        "detail": ("pe_aggregate_signatures: V compressed BLSSignatures + the attesters' index lists resident in HBM -> "
                   "k_g2_decompress (whole chip) -> k_g2_accumulate -> k_g2_finish -> 2048 compressed aggregates; synchronous, "
                   "one epoch per call; each call's aggregates sampled against oracle/g2.py's closed form"),
    }
