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

from bench_legs.cpu import cpu_baseline, cpu_step, cpu_step_inputs, pyspec_c1_baseline  # noqa: E402,F401
from bench_legs.sharded import (ReplayCollectives, _SoloDist, committee_step_check, run_step_committee,  # noqa: E402,F401
                                run_step_sharded, run_step_sharded_pipelined, sharded_step_check)
from bench_legs.signed import signed_steps, unaggregated_signatures  # noqa: E402,F401
from bench_legs.slots import slot_cadence  # noqa: E402,F401
from bench_legs.verify import replay_and_verify, step_digest, whole_step_check  # noqa: E402,F401
from bench_legs.workload import _BREAKDOWN, build_workload, load_registry, run_step_single, state_ctx  # noqa: E402,F401


# BASELINE.json configs[1..4] as written there (configs[0] is the CPU-only plumbing case: pyspec_c1 below)
# (SURVEY.md 8(d)'s table: c2 = a 2048-block chain with geometric side branches; c5 = mixed balances, 1 % equivocating,
# 0.5 % inactive -- the inactive fraction is the same in every shape)
SHAPES = {
    "configs1": dict(validators=1 << 16, committees=2048, blocks=2048, mixed_balances=False, tree_kind="branchy",
                     equivocating_frac=0.0),
    "configs2": dict(validators=1 << 18, committees=2048, blocks=4096, mixed_balances=False, tree_kind="bushy",
                     equivocating_frac=0.0),
    "configs3": dict(validators=1 << 20, committees=2048, blocks=4096, mixed_balances=False, tree_kind="bushy",
                     equivocating_frac=0.0),
    "configs4": dict(validators=1 << 22, committees=2048, blocks=8192, mixed_balances=True, tree_kind="bushy",
                     equivocating_frac=0.01),
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
    ap.add_argument("--tree-kind", choices=["bushy", "branchy", "chain"], default=None,
                    help="block tree of the workload (synth.random_tree); default: the shape's")
    ap.add_argument("--equivocating-frac", type=float, default=None,
                    help="fraction of the registry in store.equivocating_indices (pe:897); default: the shape's "
                         "(configs4: 0.01, SURVEY 8(d) c5)")
    ap.add_argument("--boost", action="store_true",
                    help="set proposer_boost_root on a recent leaf after every step's on_tick (SURVEY 8(d) \"boost set on a leaf\")")
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

    # fd 1 carries exactly ONE line, the JSON line of rank 0: libraries print to it too (gloo's "[Gloo] Rank 0 is connected
    # to ..." at the rendezvous, RCCL's version banner -- through C stdio, i.e. flushed at exit, BEHIND the JSON line).  From
    # here on everything written to stdout by anyone goes to stderr; the line itself is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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

        if world == 1:  # POSEVO_FORCE_DIST without a launcher: a one-rank rendezvous of its own
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29571")):
                os.environ.setdefault(k, v)

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
        # Inside the timed region only the roofline's kernel is bracketed (k_g1_accumulate, one launch in four): every bracket
        # is two event packets on a stream of latency-sized kernels plus host time, and twelve more of them per step were
        # charged to the engine (round 6: profiles/NOTES_r06.md).  The other kernels' averages are those of the warm-up steps.
        prof_warm = e.profile() if args.warmup >= 3 else None
        if prof_warm is not None:
            e.profile_enable(3)
        e.profile_reset()   # the warm-up launches are not part of the timed region's averages
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
        acc_mhz = e.profile_accumulate_mhz()   # the clock every timed step's dominant kernel ran at
        prof = e.profile()
        e.profile_enable(False)
        if prof_warm is not None:
            prof = {k: (v if k == "g1_accumulate" or v["launches"] else prof_warm[k]) for k, v in prof.items()}
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
                    sharded_chk=sharded_chk, dt_var=dt_var, acc_mhz=acc_mhz)

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
    # k_votes / k_tree alone (in the timed steps they run as halves of paired launches, engine_pair.cpp): their own durations
    # from a few more synchronous heads under the engine's event brackets, after the latency calls
    prof_head = None
    if ex is None or args.by_committee:
        e.profile_enable(True)
        e.profile_reset()
        for _ in range(50):
            e.get_head()
        prof_head = e.profile()
        e.profile_enable(False)
    # The signed step once more BESIDE this handle, before it goes (how rounds 4 and 5 measured that leg until now): the
    # difference to the same leg on a handle of its own further down is the runtime's stream -> hardware-queue mapping and
    # nothing else.
    signed_beside = None
    signed_ok = (world == 1 and not emulate and not args.no_signed_steps and not args.no_pipeline and not args.host_rows
                 and not args.host_arena and not args.no_lag)
    if signed_ok and dist is None:
        try:
            n_signed = min(20, args.steps)
            signed_beside = signed_steps(pea, w, local_rank, min(3, len(w["steps"]) - n_signed), n_signed, args.lag)
        except Exception as err:
            print(f"[bench] with_signatures (beside the main handle) failed: {err!r}", file=sys.stderr)
    # The main handle is done: release it before the legs below create theirs.  The runtime maps a process's streams onto its
    # four hardware queues as they are created; with this handle's six still alive a second handle's streams share queues with
    # their own siblings -- its row kernels queued behind its own finish kernel: the per-slot run read 308 us per slot-step
    # with every stream strictly behind the other for that reason (profiles/r05_slot_timeline.txt).
    if dist is None and not emulate:
        e.close()

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
    macs = mixed_adds * MACS_ADD   # (until round 6 + 4 products per lane: the hand-over's conversion, now the tree's, per group)
    # the multiplier's issue rate (tools/ubench_valu, profiles/r01_ubench_valu_fpmul.log: v_mad_u64_u32 every 2.496 ns per SIMD
    # at two waves, 2.454 at four; v_mad_i64_i32 is the same unit): 1024 SIMDs x 64 lanes
    MAC_PEAK = 1024 * 64 / 2.496e-9
    valu_peak = MAC_PEAK / MACS_ADD              # mixed adds per second if the SIMDs issued nothing but multiply-adds
    valu_ach = mixed_adds / (acc_ms * 1e-3) if acc_ms else 0.0
    # the tree over the lanes' partials (S29 form since round 6: 12 products + 2 squarings per full add, one add per lane but one
    # per committee, + 4 products per committee for the hand-over to k_g1_finish's words); beside the NEXT accumulation
    tree_macs = (12.0 * MACS_MUL + 2.0 * MACS_SQR) * max(lane_runs - C, 0) + 4.0 * MACS_MUL * C
    # the clock the timed steps' accumulations ran at (pe_profile_accumulate_mhz); the multiplier's rate above is one multiply-add
    # per SIX shader cycles and SIMD, which is 2.496 ns at 2.404 GHz -- the sustained clock; a run of a few milliseconds from an
    # idle device sees ~2.05 GHz (tools/clockramp.hip)
    acc_mhz = [float(x) for x in R["acc_mhz"] if x > 0]
    mhz_mean = sum(acc_mhz) / len(acc_mhz) if acc_mhz else None
    votes = prof_head["votes"] if prof_head is not None and prof_head["votes"]["launches"] else prof["votes"]
    votes_ms = votes["total_ms"] / max(votes["launches"], 1)
    votes_bytes = 13.0 * VL + 32.0 * args.blocks
    kernel_ms = {k: (v["total_ms"] / v["launches"] if v["launches"] else None) for k, v in prof.items()}
    per_step = np.diff(np.array(stamps)) * 1e3
    V_total = VL if args.by_committee else VL * world   # committee shards: one registry, on every rank
    named = {(sh["validators"], sh["committees"], sh["blocks"], sh["mixed_balances"], sh["tree_kind"], sh["equivocating_frac"]):
             int(name[-1]) for name, sh in SHAPES.items()}.get(
                 (V_total, C, args.blocks, bool(args.mixed_balances), args.tree_kind, float(args.equivocating_frac)))
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
            "streaming pipelines (pe_pipeline_begin_streaming / _end_lagged: a step's G1 sums overlap the next step"
            + (")" if os.environ.get("POSEVO_PAIR") == "0" else
               "; its fork-choice kernels are launched pairwise with the next step's row kernels, engine_pair.cpp)"))

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
        "dtype_detail": ("381-bit Fp, exact integer arithmetic: accumulation (dominant kernel) and tree in 14 signed 29-bit "
                         "limbs held in int32 with int64 column sums (fp381_s29.h); finish in 12 x u32 Montgomery limbs; "
                         "u64 Gwei weights"),
        "data": "synthetic",
        "config": {
            "workload": shape + f": {V_total} validators on {world} GPU(s) ({VL} per GPU), "
                        f"{C} committees x {V_total // C}, {args.parts} partial aggregates/committee, "
                        f"99% participation, {args.blocks}-block tree ({args.tree_kind})"
                        + (f", {args.equivocating_frac:.1%} of the registry equivocating" if args.equivocating_frac else "")
                        + (", proposer boost set on a leaf in every step" if args.boost else "")
                        + f", one epoch per step: pe_aggregate (union + "
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
        "ms_per_step_first_20": float((stamps[min(20, len(stamps) - 1)] - stamps[0]) / min(20, len(stamps) - 1) * 1e3),
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
            "launches_detail": "launches bracketed with HIP events on the kernel's own stream inside the timed region: one in "
                               "four (every event record on that stream is step time since the accumulation paces the step)",
            "note": "integer-VALU bound (3738 multiply-adds per 100 B gathered), not HBM bound: "
                    "see roofline_valu and DESIGN.md; the votes kernel below is the HBM-streaming one",
        },
        "shader_mhz": ({"k_g1_accumulate_first_timed_launch": round(acc_mhz[0], 1), "last_timed_launch": round(acc_mhz[-1], 1),
                        "mean": round(mhz_mean, 1), "min": round(min(acc_mhz), 1), "max": round(max(acc_mhz), 1),
                        "launches": len(acc_mhz),
                        "how": "workgroup 0 of every timed k_g1_accumulate launch counts shader cycles against the fixed 100 MHz "
                               "counter (pe_profile_accumulate_mhz)",
                        "why": "the step is bound by the integer multiplier, so it follows the power management's clock one to "
                               "one: ~2.05 GHz for the first milliseconds of heavy load after an idle moment, ~2.4 GHz after ~35 ms "
                               "of it (tools/clockramp.hip, profiles/r06_clockramp.txt) -- most of the distance between "
                               "--steps 20 and --steps 200"} if acc_mhz else None),
        "roofline_valu": {
            "kernel": "k_g1_accumulate", "bound": "integer VALU: v_mad_i64_i32 / v_mad_u64_u32 issue (S29 field form)",
            "achieved": valu_ach / 1e9, "peak": valu_peak / 1e9, "unit": "G mixed adds/s", "frac": valu_ach / valu_peak,
            "mixed_adds_per_launch": mixed_adds, "multiply_adds_per_mixed_add": MACS_ADD,
            "multiply_adds_per_launch": macs,
            "peak_source": "instruction ceiling: one 32 x 32 -> 64 multiply-add per 2.496 ns and SIMD (tools/ubench_valu, "
                           "profiles/r01_ubench_valu_fpmul.log) x 1024 SIMDs x 64 lanes / 3738 multiply-adds per mixed add = "
                           "7.0 G/s; the same adds in a loop without loads reach 6.96 G/s (tools/icbench, "
                           "profiles/r04_icbench.txt), the kernel alone 5.8 G/s (tools/accbench, profiles/r04_accbench.txt)",
            "at_measured_clock": ({"mean_shader_mhz": round(mhz_mean, 1), "peak": 1024 * 64 * mhz_mean * 1e6 / 6.0 / MACS_ADD / 1e9,
                                   "frac": valu_ach / (1024 * 64 * mhz_mean * 1e6 / 6.0 / MACS_ADD),
                                   "note": "the same ceiling at the clock the timed launches ran at (six cycles per multiply-add "
                                           "and SIMD): what the kernel makes of the cycles it was given"}
                                  if mhz_mean else None),
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
            "measured": ("stand-alone launches of 50 synchronous pe_get_head calls after the timed region (in the streaming steps "
                         "k_votes runs as one half of a paired launch: kernel_avg_ms.pair_members_votes)"
                         if votes is not prof["votes"] else "the timed steps' own launches"),
        },
        "kernel_avg_ms": kernel_ms,
        "kernel_avg_ms_detail": ("g1_accumulate: HIP events inside the timed region (one launch in four); every other kernel: the "
                                 "warm-up steps' launches (nothing but the roofline's kernel is bracketed while the clock runs)"
                                 if args.warmup >= 3 else "HIP events inside the timed region"),
        "kernel_launches": {k: v["launches"] for k, v in prof.items()},
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
    if signed_ok:
        n_signed = min(20, args.steps)
        try:
            out["with_signatures"] = signed_steps(pea, w, local_rank, min(3, len(w["steps"]) - n_signed), n_signed, args.lag)
            out["ms_per_step_with_signatures"] = out["with_signatures"]["ms_per_step_with_signatures"]
            if signed_beside is not None:
                out["with_signatures"]["ms_per_step_beside_another_handle"] = signed_beside["ms_per_step_with_signatures"]
                out["with_signatures"]["beside_another_handle_detail"] = (
                    "the same leg run while the headline's handle was still alive (rounds 4-5 measured it so): a second "
                    "handle's streams land on other hardware queues than a first one's; steps verified "
                    f"{signed_beside.get('steps_verified')}")
        except Exception as err:   # an extra leg: reported (here and on stderr), never at the cost of the headline line
            print(f"[bench] with_signatures failed: {err!r}", file=sys.stderr)
            out["with_signatures"] = {"error": repr(err)}
    if world == 1 and not emulate and not args.no_signed_steps:
        try:
            out["with_unaggregated_signatures"] = unaggregated_signatures(pea, w, local_rank)
        except Exception as err:   # as above
            print(f"[bench] with_unaggregated_signatures failed: {err!r}", file=sys.stderr)
            out["with_unaggregated_signatures"] = {"error": repr(err)}
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
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

