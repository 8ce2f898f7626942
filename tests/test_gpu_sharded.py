"""-m gpu: the multi-GPU exchange path on ONE GPU: (a) two engines each owning half of the validators, their
partials summed / concatenated exactly as the all-reduce / all-gather would; (b) ShardedForkChoice over RCCL with
world_size 1.  Both must reproduce the unsharded engine bit for bit."""
import os

import numpy as np
import pytest

import pos_evolution_amd.synth as synth
from pos_evolution_amd import _abi
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _workload(V=40000, B=300, C=64, seed=21):
    tree = synth.random_tree(B, seed, "bushy")
    bal = synth.balances(V, seed, mixed=True)
    flags = synth.validator_flags(V, seed, inactive_frac=0.01)
    comm = synth.random_committees(V, C, seed)
    return tree, bal, flags, comm


def _load(e, tree, bal, flags, pts, comm, epoch, boost=True):
    H.load_tree(e, tree)
    e.set_validators(bal, flags, pts)
    e.set_committees(epoch, comm.offsets, comm.members)
    e.on_tick((epoch + 1) * 32 * 12)
    if boost:
        e.set_proposer_boost(tree.roots[tree.roots.shape[0] - 1].tobytes())


def _local_committees(comm, lo, hi):
    members, offs = [], [0]
    for c in range(comm.offsets.size - 1):
        m = comm.members[comm.offsets[c]:comm.offsets[c + 1]]
        m = m[(m >= lo) & (m < hi)] - lo
        members.append(m)
        offs.append(offs[-1] + m.size)
    return synth.Committees(np.array(offs, dtype=np.uint32), np.concatenate(members).astype(np.uint32))


def _local_attestations(atts, bit_rows, comm, lo, hi):
    """Restrict each attestation's bits to the members inside [lo, hi) (order preserved)."""
    spe, cps = 32, (comm.offsets.size - 1) // 32
    rows = []
    for a, bits in zip(atts, bit_rows):
        c = int((a["slot"] % spe) * cps + a["index"])
        m = comm.members[comm.offsets[c]:comm.offsets[c + 1]]
        rows.append(np.asarray(bits)[(m >= lo) & (m < hi)])
    arena, offs, nb = synth.pack_bit_rows(rows)
    out = atts.copy()
    out["bits_offset"], out["n_bits"] = offs, nb
    return out, arena


def test_two_shards_on_one_gpu(engine_factory):
    import torch
    V, C = 40000, 64
    tree, bal, flags, comm = _workload(V=V, C=C)
    whole = engine_factory()
    pts = synth.registry_points(whole, V)
    epoch = int(tree.slot.max()) // 32 + 1
    _load(whole, tree, bal, flags, pts, comm, epoch)
    atts, arena, bit_rows = synth.epoch_attestations(comm, tree, epoch, 32, seed=5, density=0.8, parts=2)
    ref = whole.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    rows = ref["atts"]
    st, _, _ = whole.on_attestation_batch(packed=(rows, ref["out_arena"]))
    assert (st == 0).all()
    ref_head, ref_w = whole.get_head(), whole.get_weights()

    n_shards = 2
    dev = torch.device("cuda", 0)
    B = tree.roots.shape[0]
    wsum = torch.zeros(B + _abi.PE_EXCHANGE_EXTRA, dtype=torch.int64, device=dev)
    PW = _abi.PE_G1_PARTIAL_BYTES // 4
    gathered = torch.zeros(n_shards * C * PW, dtype=torch.int32, device=dev)
    shards = []
    for r in range(n_shards):
        lo, hi = r * V // n_shards, (r + 1) * V // n_shards
        e = engine_factory()
        lc = _local_committees(comm, lo, hi)
        _load(e, tree, bal[lo:hi], flags[lo:hi], pts[lo:hi], lc, epoch)
        la, larena = _local_attestations(atts, bit_rows, comm, lo, hi)
        part = torch.zeros(C * PW, dtype=torch.int32, device=dev)
        with pytest.raises(AssertionError):   # capacity is checked before anything is written (PE_ERR_CAPACITY)
            e.aggregate_partial(part.data_ptr(), packed=(la, larena), capacity_groups=C - 1)
        res = e.aggregate_partial(part.data_ptr(), packed=(la, larena), capacity_groups=C)
        torch.cuda.synchronize()
        assert res["n_groups"] == C
        gathered[r * C * PW:(r + 1) * C * PW] = part
        lrows = res["atts"]
        st, _, _ = e.on_attestation_batch(packed=(lrows, res["out_arena"]))
        assert (st == 0).all()
        buf = torch.zeros_like(wsum)
        e.votes_partial(buf.data_ptr())
        torch.cuda.synchronize()
        wsum += buf                     # what the all-reduce does
        shards.append(e)
    torch.cuda.synchronize()
    # group order must agree between shards and the whole: compare through the committee id of each group
    got_pk = shards[0].g1_finish(gathered.data_ptr(), n_shards, C)
    assert np.array_equal(got_pk, ref["aggpk96"])
    for e in shards:
        assert e.head_from_weights(wsum.data_ptr()) == ref_head
    # per-block weights computed from the REDUCED buffer = the unsharded engine's weights, block for block (incl. the
    # proposer boost, which needs the global active balance carried in the buffer's extra entries)
    for e in shards:
        e.head_from_weights(wsum.data_ptr())
        assert np.array_equal(e.last_weights(), ref_w)
    # a shard on its own weighs only its validators: the two local weight vectors (without boost) add up to the
    # reduced one minus the boost it carries once
    local = [e.get_weights() for e in shards]
    boost_w = whole.get_weights() - ref_w   # zero: same store
    assert not boost_w.any()
    assert (local[0] <= ref_w).all() and (local[1] <= ref_w).all()


def test_sharded_forkchoice_world_size_one(engine_factory):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        from pos_evolution_amd.sharded import ShardedForkChoice
        V, C = 20000, 64
        tree, bal, flags, comm = _workload(V=V, C=C, seed=22)
        e = engine_factory()
        pts = synth.registry_points(e, V)
        epoch = int(tree.slot.max()) // 32 + 1
        _load(e, tree, bal, flags, pts, comm, epoch)
        atts, arena, _ = synth.epoch_attestations(comm, tree, epoch, 32, seed=6, density=0.9, parts=3)
        ref = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
        sh = ShardedForkChoice(e, n_groups_max=C)
        got = sh.aggregate(packed=(atts, arena))
        assert np.array_equal(got["aggpk96"], ref["aggpk96"])
        rows = ref["atts"]
        e.on_attestation_batch(packed=(rows, ref["out_arena"]))
        assert sh.get_head() == e.get_head()
        e.set_stream(0)
    finally:
        if created:
            dist.destroy_process_group()


def test_engine_owned_rccl_world_size_one(engine_factory):
    """The C ABI's own exchange (pe_dist_init / pe_get_head_sharded / pe_aggregate_sharded): the engine loads librccl,
    owns the communicator and issues ncclAllReduce / ncclAllGather on its stream.  One rank here (one GPU per box):
    results must equal the unsharded calls; the N > 1 arithmetic is what test_two_shards_on_one_gpu checks."""
    V, C = 20000, 64
    tree, bal, flags, comm = _workload(V=V, C=C, seed=23)
    e = engine_factory()
    pts = synth.registry_points(e, V)
    epoch = int(tree.slot.max()) // 32 + 1
    _load(e, tree, bal, flags, pts, comm, epoch)
    atts, arena, _ = synth.epoch_attestations(comm, tree, epoch, 32, seed=7, density=0.9, parts=3)
    ref = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    with pytest.raises(AssertionError):
        e.get_head_sharded()                      # PE_ERR_STATE before pe_dist_init
    e.dist_init(e.dist_unique_id(), 0, 1)
    got = e.aggregate_sharded(packed=(atts, arena))
    assert got["n_groups"] == ref["n_groups"]
    assert np.array_equal(got["aggpk96"], ref["aggpk96"])
    assert np.array_equal(got["out_arena"], ref["out_arena"]) and np.array_equal(got["count"], ref["count"])
    st, _, _ = e.on_attestation_batch(packed=(ref["atts"], ref["out_arena"]))
    assert (st == 0).all()
    assert e.get_head_sharded() == e.get_head()
    assert np.array_equal(e.last_weights(), e.get_weights())
    # the exchange buffer is self-cleaning and laid out for one block count / registry size: change both
    for k in range(3):
        tip = tree.roots[tree.roots.shape[0] - 1].tobytes() if k == 0 else new_root
        new_root = bytes([0xA0 + k]) * 32
        e.add_block(new_root, tip, int(tree.slot.max()) + 1 + k, (0, tree.roots[0].tobytes()), (0, tree.roots[0].tobytes()))
        assert e.get_head_sharded() == e.get_head()
        assert np.array_equal(e.last_weights(), e.get_weights())
    e.set_validators(bal[: V // 2], flags[: V // 2], pts[: V // 2])
    assert e.get_head_sharded() == e.get_head()
    assert np.array_equal(e.last_weights(), e.get_weights())
    e.set_validators(bal, flags, pts)
    # the thin Python face of the same path
    from pos_evolution_amd.sharded import ShardedForkChoice
    e.dist_destroy()
    sh = ShardedForkChoice(e, n_groups_max=C, use_engine_rccl=True)
    assert sh.get_head() == e.get_head()
    assert np.array_equal(sh.aggregate(packed=(atts, arena))["aggpk96"], ref["aggpk96"])
    e.dist_destroy()


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_two_ranks_dry_run_on_one_gpu(scaling):
    """bench.py's N > 1 path end to end with two processes sharing this GPU (gloo, host-staged collectives): both ranks
    must reach the same head (bench.py asserts it) and rank 0 must print the contract's JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POSEVO_DIST_BACKEND="gloo", POSEVO_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--validators", "65536", "--blocks", "512", "--committees", "256", "--head-calls", "5", "--no-cpu-baseline",
           "--scaling", scaling]   # a small custom shape, both ways of dividing it
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["parallelism"] == f"validator-range shards x2 ({scaling} scaling)"
    per_gpu = 65536 // 2 if scaling == "strong" else 65536
    assert d["config"]["validators_per_gpu"] == per_gpu and d["config"]["validators_total"] == 2 * per_gpu
    # the engine-owned streaming step over the caller's collectives, and the run's first step held against the oracle on
    # every rank (shard-local state, head + weights over the gathered vote tables, all aggregate pubkeys)
    assert "pe_dist_init_custom" in d["config"]["call_mode"]
    assert d["checked_against_oracle"] is True and d["oracle_check"]["aggregate_pubkeys"] and d["oracle_check"]["weights"]


@pytest.mark.parametrize("shape,index,validators,blocks", [("configs3", 3, 1 << 20, 4096), ("configs4", 4, 1 << 22, 8192)])
def test_bench_two_ranks_run_the_named_configs_by_default(shape, index, validators, blocks):
    """What the round driver launches for N > 1 -- `bench.py --gpus N` with no shape flags -- is BASELINE configs[3] AS
    WRITTEN (one 1 048 576-validator registry divided over the N GPUs, strong scaling), and `--shape configs4` is
    configs[4] (4 194 304 validators, mixed balances, 8192 blocks): two ranks sharing this GPU (gloo, host-staged
    collectives), the first step held against the oracle on every rank, `roofline` in the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POSEVO_DIST_BACKEND="gloo", POSEVO_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29560 + index), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--head-calls", "5", "--no-cpu-baseline"] + ([] if shape == "configs3" else ["--shape", shape])
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["workload"].startswith(f"BASELINE configs[{index}] over 2 GPUs"), d["config"]["workload"]
    assert d["config"]["validators_total"] == validators and d["config"]["validators_per_gpu"] == validators // 2
    assert d["config"]["blocks"] == blocks
    assert d["checked_against_oracle"] is True and d["oracle_check"]["aggregate_pubkeys"] and d["oracle_check"]["weights"]
    assert d["roofline"]["kernel"] == "k_g1_accumulate" and d["roofline"]["achieved"] > 0


def test_bench_two_ranks_committee_sharded_dry_run_on_one_gpu():
    """bench.py --sharded-mode committee with two processes sharing this GPU: committee shards (SURVEY.md 8e Option B), the
    run's first step checked against the oracle on every rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POSEVO_DIST_BACKEND="gloo", POSEVO_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29551", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--validators", "65536", "--blocks", "512", "--committees", "256", "--head-calls", "5", "--no-cpu-baseline",
           "--sharded-mode", "committee"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["parallelism"].startswith("committee shards x2")
    assert d["config"]["validators_per_gpu"] == 65536 and "no G1 collective" in d["config"]["exchange"]
    assert d["checked_against_oracle"] is True, d.get("oracle_check")
    assert d["oracle_check"]["own_aggregate_pubkeys"] and d["oracle_check"]["head"] and d["oracle_check"]["latest_messages"]


@pytest.mark.parametrize("lagged", [False, True])
def test_engine_owned_rccl_pipelined_step_world_size_one(engine_factory, lagged):
    """The sharded step as bench.py runs it for N > 1: pe_aggregate_sharded, the handlers on the resident unions and
    pe_get_head_sharded inside pipelined calls, nothing waiting between them.  One rank: every output must equal the
    unsharded synchronous calls on a twin engine."""
    import pos_evolution_amd as pea
    from pos_evolution_amd._abi import pe_state_ctx

    V, C = 20000, 64
    tree, bal, flags, comm = _workload(V=V, C=C, seed=24)
    epoch = int(tree.slot.max()) // 32 + 1
    atts, arena, _ = synth.epoch_attestations(comm, tree, epoch, 32, seed=8, density=0.9, parts=3,
                                              source=(0, tree.roots[0].tobytes()))
    ctx = pe_state_ctx()
    ctx.slot = (epoch + 1) * 32
    ctx.chain_tip_root[:] = tree.roots[tree.roots.shape[0] - 1].tobytes()
    ctx.current_justified_root[:] = tree.roots[0].tobytes()
    ctx.previous_justified_root[:] = tree.roots[0].tobytes()
    ctx.base_reward_per_increment = 357
    engines = []
    for _ in range(2):
        e = engine_factory()
        pts = synth.registry_points(e, V)
        _load(e, tree, bal, flags, pts, comm, epoch)
        engines.append(e)
    ref_e, e = engines
    ref = ref_e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    ref_st, _, ref_cnt = ref_e.on_attestation_batch(packed=(ref["atts"], ref["out_arena"]))
    ref_head = ref_e.get_head()
    ref_pst, ref_num = ref_e.process_attestation_batch(ctx, packed=(ref["atts"], ref["out_arena"]))

    e.dist_init(e.dist_unique_id(), 0, 1)
    for rep in range(3):   # several steps in flight when lagged; the repeats change nothing (same votes, flags already set)
        with e.pipeline(lagged=lagged):
            got = e.aggregate_sharded(packed=(atts, arena))
            st, _, cnt = e.on_attestation_batch(packed=(got["atts"], pea.RESIDENT))
            head = e.get_head_sharded()
            pst, num = e.process_attestation_batch(ctx, packed=(got["atts"], pea.RESIDENT))
        if rep == 0:
            first = (got, st, cnt, head, pst, num)
    e.drain()
    got, st, cnt, head, pst, num = first
    assert got["n_groups"] == ref["n_groups"]
    assert np.array_equal(got["aggpk96"], ref["aggpk96"])
    assert np.array_equal(got["out_arena"], ref["out_arena"]) and np.array_equal(got["count"], ref["count"])
    assert np.array_equal(st, ref_st) and np.array_equal(cnt, ref_cnt)
    assert head == ref_head
    assert np.array_equal(pst, ref_pst) and np.array_equal(num, ref_num)
    assert (ref_st == 0).all() and (ref_pst == 0).all() and int(ref_num.sum()) > 0
    assert np.array_equal(e.get_weights(), ref_e.get_weights())
    e.dist_destroy()


def test_bench_engine_rccl_path_one_rank():
    """bench.py's default N > 1 step (collectives issued by the engine inside pipelined calls) driven with ONE rank
    over RCCL: the same code path the 2/4/8-GPU runs take, minus the peers."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POSEVO_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29549", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "3",
           "--validators", "65536", "--blocks", "512", "--committees", "256", "--head-calls", "5", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert "collectives issued by the engine" in d["config"]["call_mode"]
    assert d["checked_against_oracle"] is True, d.get("oracle_check")
