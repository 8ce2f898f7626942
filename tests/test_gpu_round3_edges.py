"""-m gpu: corners of round 3's additions -- lag depths, the asynchronous shuffle, the signature leg over overlapping members,
errors of the committee-sharded exchange."""
import hashlib

import numpy as np
import pytest

import pos_evolution_amd as pea
import pos_evolution_amd.synth as synth
from pos_evolution_amd import RESIDENT, ROWS_RESIDENT
from pos_evolution_amd._abi import pe_state_ctx
from tests import helpers as H
from tests.test_gpu_resident_rows import _assert_same, _assert_same_state, _dev_arena, _dev_rows

pytestmark = pytest.mark.gpu


def _epochs(engine_factory, n_val, n_comm, steps, lag=None):
    ea, eb = engine_factory(max_committee_tables=steps + 4), engine_factory(max_committee_tables=steps + 4)
    tree = synth.random_tree(100, 6, "bushy")
    pts, _ = H.oracle_points(n_val)
    bal = synth.balances(n_val, 6, mixed=True)
    flags = synth.validator_flags(n_val, 6, inactive_frac=0.01)
    for e in (ea, eb):
        H.load_tree(e, tree)
        e.set_validators(bal, flags, pts)
    return ea, eb, tree


@pytest.mark.parametrize("lag", [1, 3, 5])
def test_streaming_epochs_at_other_lag_depths(engine_factory, lag):
    """pe_pipeline_set_lag: the same streaming steps at lag depths 1, 3 and 5 give what synchronous host-row calls give,
    and a step's outputs are complete once `lag` further blocks have exited."""
    n_val, n_comm, spe, steps = 30000, 64, 32, 8
    ea, eb, tree = _epochs(engine_factory, n_val, n_comm, steps)
    eb.set_pipeline_lag(lag)
    eb.reuse_outputs(steps + 2)
    ep0 = int(tree.slot.max()) // spe + 1
    ref, got, work = [], [], []
    for s in range(steps):
        ep = ep0 + s
        seed = hashlib.sha256(b"lag" + ep.to_bytes(8, "little")).digest()
        for e in (ea, eb):
            off, mem = e.compute_committees(ep, seed, n_val, n_comm, 8)
        comm = synth.Committees(off, mem)
        atts, arena, _ = synth.epoch_attestations(comm, tree, ep, spe, seed=2, density=0.9, parts=3,
                                                  source=(0, tree.roots[0].tobytes()), vote_recent=16)
        ctx = pe_state_ctx()
        ctx.slot = (ep + 1) * spe
        ctx.chain_tip_root[:] = tree.roots[tree.roots.shape[0] - 1].tobytes()
        ctx.current_justified_root[:] = tree.roots[0].tobytes()
        ctx.previous_justified_root[:] = tree.roots[0].tobytes()
        ctx.base_reward_per_increment = 321
        work.append((ep, atts, arena, ctx, _dev_rows(atts), _dev_arena(arena)))
    for ep, atts, arena, ctx, _, _ in work:
        ea.on_tick((ep + 1) * spe * 12)
        ea.participation_rotate()
        agg = ea.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
        st, _, cnt = ea.on_attestation_batch(packed=(agg["atts"], agg["out_arena"]))
        head = ea.get_head()
        pst, num = ea.process_attestation_batch(ctx, packed=(agg["atts"], agg["out_arena"]))
        ref.append((agg, st, cnt, pst, num, head))
    for k, (ep, atts, arena, ctx, rows, bits) in enumerate(work):
        eb.on_tick((ep + 1) * spe * 12)
        eb.participation_rotate()
        with eb.pipeline(lagged=True):
            agg = eb.aggregate(packed=(rows, bits), want_aggregate_pubkeys=True)
            st, _, cnt = eb.on_attestation_batch(packed=(ROWS_RESIDENT, RESIDENT), cap=n_comm)
            head = eb.get_head_async()
            pst, num = eb.process_attestation_batch(ctx, packed=(ROWS_RESIDENT, RESIDENT), cap=n_comm)
        got.append((agg, st, cnt, pst, num, head))
        if k >= lag:   # complete by contract: `lag` further blocks have exited
            r, g = ref[k - lag], got[k - lag]
            _assert_same(r, (g[0], g[1], g[2], g[3], g[4], bytes(g[5])))
    eb.drain()
    for r, g in zip(ref, got):
        _assert_same(r, (g[0], g[1], g[2], g[3], g[4], bytes(g[5])))
    _assert_same_state(ea, eb)
    with pytest.raises(pea.EngineError):
        with eb.pipeline():
            eb.set_pipeline_lag(2)          # not inside a pipeline


def test_compute_committees_async_equals_synchronous(engine_factory):
    """pe_compute_committees_async: same table as the synchronous call, usable by the very next aggregate (the engine's
    stream waits for the shuffle), rewritten in place when the epoch already has a table."""
    n_val, n_comm, spe = 40000, 64, 32
    ea, eb, tree = _epochs(engine_factory, n_val, n_comm, 4)
    ep = int(tree.slot.max()) // spe + 1
    seed = hashlib.sha256(b"async").digest()
    off, mem = ea.compute_committees(ep, seed, n_val, n_comm, 90)
    eb.compute_committees_async(ep, seed, n_val, n_comm, 90)
    off_b, mem_b = eb.committees(ep)
    assert np.array_equal(off, off_b) and np.array_equal(mem, mem_b)
    comm = synth.Committees(off, mem)
    atts, arena, _ = synth.epoch_attestations(comm, tree, ep, spe, seed=3, density=0.8, parts=2,
                                              source=(0, tree.roots[0].tobytes()))
    for e in (ea, eb):
        e.on_tick((ep + 1) * spe * 12)
    ref = ea.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    # a second asynchronous shuffle of the same epoch (same seed: same table, rewritten in place), then straight into a step
    eb.compute_committees_async(ep, seed, n_val, n_comm, 90)
    with eb.pipeline():
        got = eb.aggregate(packed=(_dev_rows(atts), _dev_arena(arena)), want_aggregate_pubkeys=True)
    assert got["n_groups"] == ref["n_groups"] and np.array_equal(got["aggpk96"], ref["aggpk96"])
    assert np.array_equal(got["out_arena"], ref["out_arena"])


def test_aggregate_signed_overlapping_members_keep_their_plain_sum(engine_factory):
    """Members whose bits overlap: the row carries PE_ATT_FLAG_OVERLAPPING_BITS and loses PE_ATT_FLAG_SIGNATURE_VALID as with
    pe_aggregate (A.8), and its signature is the plain sum of the members' signatures (ADVICE r1 #1 semantics)."""
    from oracle import g2
    from pos_evolution_amd import _abi
    from tests.test_gpu_pipeline import _world

    w = _world(engine_factory, 5000, 32, seed=8, density=0.7, parts=3)
    e, atts, arena = w["e"], w["atts"].copy(), w["arena"].copy()
    n = len(atts)
    # make row 1 of some committee repeat a bit of row 0 of the same committee
    gof = np.asarray(e.aggregate(packed=(atts, arena))["group_of"])
    g0 = int(gof[0])
    mates = np.nonzero(gof == g0)[0]
    a, b = int(mates[0]), int(mates[1])
    first_bit = int(np.nonzero(np.unpackbits(arena[atts["bits_offset"][a]:atts["bits_offset"][a] + 8], bitorder="little"))[0][0])
    arena[atts["bits_offset"][b] + first_bit // 8] |= np.uint8(1 << (first_bit % 8))
    A, B = 0x77, 0x1003
    pts = g2.synthetic_points(n, A, B)
    sigs = np.frombuffer(b"".join(g2.compress(p) for p in pts), dtype=np.uint8).reshape(n, 96)
    for packed in ((atts, arena), (_dev_rows(atts), _dev_arena(arena))):
        res = e.aggregate_signed(sigs, packed=packed)
        k = int(np.asarray(res["group_of"])[a])
        fl = int(res["atts"][k]["flags"])
        assert fl & _abi.PE_ATT_FLAG_OVERLAPPING_BITS and not fl & _abi.PE_ATT_FLAG_SIGNATURE_VALID
        R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
        exp = g2.mul((A * len(mates) + B * int(mates.sum())) % R, g2.G2)
        assert res["sig96c"][k].tobytes() == g2.compress(exp) and (res["sig_status"] == 0).all()
        others = [j for j in range(res["n_groups"]) if j != k]
        assert all(int(res["atts"][j]["flags"]) & _abi.PE_ATT_FLAG_SIGNATURE_VALID for j in others)


def test_aggregate_exchange_needs_its_preconditions(engine_factory):
    from tests.test_gpu_pipeline import _world

    w = _world(engine_factory, 4000, 32, seed=9, parts=2)
    e, atts, arena = w["e"], w["atts"], w["arena"]
    with pytest.raises(pea.EngineError) as err:      # no communicator
        e.aggregate_exchange(cap_groups=64)
    assert err.value.status == pea._abi.PE_ERR_STATE
    calls = []
    e.dist_init_custom(0, 1, lambda b, c, s: 1, lambda s_, r_, nb, st_: calls.append(nb) or 1)   # a failing all-gather
    with pytest.raises(pea.EngineError) as err:      # no aggregate over rows in device memory before it
        e.aggregate_exchange(cap_groups=64)
    assert err.value.status == pea._abi.PE_ERR_STATE
    e.aggregate(packed=(_dev_rows(atts), _dev_arena(arena)))
    with pytest.raises(pea.EngineError) as err:      # no bound on the groups: ranks would size the all-gather differently
        e.aggregate_exchange(cap_groups=len(atts))
    assert err.value.status == pea._abi.PE_ERR_STATE and "pe_dist_set_max_groups" in str(err.value)
    e.dist_set_max_groups(len(atts))
    with pytest.raises(pea.EngineError) as err:      # output arrays smaller than world x bound
        e.aggregate_exchange(cap_groups=3)
    assert err.value.status == pea._abi.PE_ERR_CAPACITY
    with pytest.raises(pea.EngineError) as err:      # the caller's collective reports failure
        e.aggregate_exchange(cap_groups=len(atts))
    assert err.value.status == pea._abi.PE_ERR_NO_DEVICE and calls
    e.dist_destroy()


class _SilentPeer:
    """The two exchange steps of rank 0 of 2 whose peer holds no validators (it adds zeros to the weights and points at
    infinity to the partials), carried out on the stream they are ordered on.  `stall_s` > 0: the exchange additionally
    blocks its stream for that long -- a peer that has not arrived yet (a host function on the stream: nothing spins on
    the device, and the wait ends by itself)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        self.HOSTFN = C.CFUNCTYPE(None, C.c_void_p)
        self.hip.hipLaunchHostFunc.argtypes = [C.c_void_p, self.HOSTFN, C.c_void_p]
        self.stall_s = 0.0
        self.stalled = 0
        self._sleep = self.HOSTFN(self._sleep_cb)   # kept alive: the runtime calls it from its own thread

    def _sleep_cb(self, _):
        import time
        time.sleep(self.stall_s)

    def _stall(self, stream):
        if self.stall_s > 0:
            self.stalled += 1
            assert self.hip.hipLaunchHostFunc(self.C.c_void_p(stream), self._sleep, None) == 0

    def all_reduce_u64(self, buf, count, stream):
        self._stall(stream)
        return 0

    def all_gather(self, send, recv, nbytes, stream):
        self._stall(stream)
        s = self.C.c_void_p(stream)
        assert self.hip.hipMemcpyAsync(self.C.c_void_p(recv), self.C.c_void_p(send), nbytes, 3, s) == 0   # device to device
        assert self.hip.hipMemsetAsync(self.C.c_void_p(recv + nbytes), 0, nbytes, s) == 0
        return 0


def test_exchange_timeout_is_reported_and_the_handle_recovers(engine_factory):
    """a rank whose peer never arrives must not wait for ever.  On a handle that exchanges with other
    ranks every wait is bounded (pe_dist_set_timeout_ms): the call reports PE_ERR_TIMEOUT, the handle refuses further
    exchanges until pe_dist_destroy, and after initialising again it computes what it computed before -- in a synchronous
    call and in the middle of a streaming pipeline (the place bench.py's fallback catches it)."""
    import time
    from tests.test_gpu_pipeline import _world

    w = _world(engine_factory, 6000, 32, seed=12, parts=2)
    e, atts, arena, ctx = w["e"], w["atts"], w["arena"], w["ctx"]
    TIMEOUT = pea._abi.PE_ERR_TIMEOUT
    peer = _SilentPeer()
    e.dist_init_custom(0, 2, peer.all_reduce_u64, peer.all_gather)
    ref = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    got = e.aggregate_sharded(packed=(atts, arena))
    assert np.array_equal(got["aggpk96"], ref["aggpk96"])            # the silent peer changes nothing
    st, _, _ = e.on_attestation_batch(packed=(ref["atts"], ref["out_arena"]))
    assert (st == 0).all()
    head = e.get_head()
    assert e.get_head_sharded() == head

    # ---- a synchronous call
    e.dist_set_timeout_ms(100)
    peer.stall_s = 0.6
    t0 = time.perf_counter()
    with pytest.raises(pea.EngineError) as err:
        e.get_head_sharded()
    waited = time.perf_counter() - t0
    assert err.value.status == TIMEOUT and 0.09 < waited < 0.5 and peer.stalled == 1, (err.value.status, waited)
    with pytest.raises(pea.EngineError) as err:                     # marked: nothing is exchanged on this handle any more
        e.get_head_sharded()
    assert err.value.status == TIMEOUT and peer.stalled == 1
    peer.stall_s = 0.0
    e.dist_destroy()                                                # waits for what the stalled stream still holds
    assert e.get_head() == head                                     # the store is what it was
    e.dist_init_custom(0, 2, peer.all_reduce_u64, peer.all_gather)
    assert e.get_head_sharded() == head
    got = e.aggregate_sharded(packed=(atts, arena))
    assert np.array_equal(got["aggpk96"], ref["aggpk96"])

    # ---- inside streaming pipelines: the timeout surfaces where the pipeline waits; destroy + close do not hang
    e.dist_set_timeout_ms(100)
    e.reuse_outputs(4)
    peer.stall_s = 0.6
    t0 = time.perf_counter()
    with pytest.raises(pea.EngineError) as err:
        for _ in range(4):
            with e.pipeline(lagged=True):
                agg = e.aggregate_sharded(packed=(atts, arena))
                e.on_attestation_batch(packed=(agg["atts"], pea.RESIDENT))
                e.get_head_sharded_async()
        e.drain()
    assert err.value.status == TIMEOUT and time.perf_counter() - t0 < 3.0
    peer.stall_s = 0.0
    e.dist_destroy()
    assert e.get_head() == head                                     # votes re-applied by the pipelines are the same votes
    e.close()
