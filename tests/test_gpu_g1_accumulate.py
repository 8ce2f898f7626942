"""-m gpu: the accumulation kernel (k_g1_accumulate over the S29 field form, g1_kernels.hip; the field and point
arithmetic itself is held against Python integers and oracle/g1.py on the CPU by tests/test_host_fp29.py) through the C
ABI: aggregate pubkeys against the closed form of the synthetic registry and against oracle/g1.py -- random keys, the
structured keys (i + 1) G that hit the doubling branch, P / -P pairs, rows that hold no point (every one of them leaves
the loop's general body and takes the kernel's redo path), caller-supplied points (pe_g1_sum: the table of the form is
built per call)."""
import numpy as np
import pytest

import pos_evolution_amd as pea
import pos_evolution_amd.synth as synth
from oracle import g1
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _aggregate(e, tree, bal, flags, pts, comm, atts, arena, epoch):
    H.load_tree(e, tree)
    e.set_validators(bal, flags, pts)
    e.set_committees(epoch, comm.offsets, comm.members)
    e.on_tick((epoch + 1) * 32 * 12)
    sync = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    with e.pipeline():
        piped = e.aggregate(packed=(atts, arena), want_aggregate_pubkeys=True)
    assert np.array_equal(sync["aggpk96"], piped["aggpk96"])
    return sync


@pytest.mark.parametrize("n_val,n_comm,density", [(4096, 32, 0.9), (30000, 64, 0.5), (20000, 32, 1.0), (70000, 32, 0.97)])
def test_aggregate_pubkeys_equal_the_closed_form(n_val, n_comm, density):
    """Committees of 128 ... 2187 members: lanes of 4 to 16 members, groups inside one workgroup and across several."""
    tree = synth.random_tree(80, 7, "bushy")
    pts, (a, b) = H.oracle_points(n_val)
    bal = synth.balances(n_val, 7, mixed=True)
    flags = synth.validator_flags(n_val, 7, inactive_frac=0.01)
    comm = synth.random_committees(n_val, n_comm, 7)
    epoch = int(tree.slot.max()) // 32 + 1
    atts, arena, bit_rows = synth.epoch_attestations(comm, tree, epoch, 32, seed=7, density=density, parts=2)
    got = _aggregate(pea.Engine(), tree, bal, flags, pts, comm, atts, arena, epoch)
    spe, cps = 32, n_comm // 32
    for k in range(got["n_groups"]):
        row = got["atts"][k]
        c = int((row["slot"] % spe) * cps + row["index"])
        members = comm.members[comm.offsets[c]:comm.offsets[c + 1]]
        idx = members[np.asarray(got["bits"][k], dtype=bool)]
        assert got["aggpk96"][k].tobytes() == H.closed_form_sum(idx, a, b)


def test_structured_keys_doubling_negatives_and_empty_rows():
    """(i + 1) G keys in index order (accumulator meets an equal point), P followed by -P (accumulator returns to
    infinity and goes on), validators without a key (all-zero rows): the same-x filter fires in every lane's run and the run
    is redone by the complete add."""
    n = 512
    keys = [g1.mul(i + 1, g1.G) for i in range(n // 4)]
    pts = np.zeros((n, 96), dtype=np.uint8)
    pattern = []
    for i in range(n):
        k = keys[(i // 4) % len(keys)]
        pt = [k, k, g1.neg(k), None][i % 4]       # runs of: P, P, -P, (no key)
        pattern.append(pt)
        if pt is not None:
            pts[i] = np.frombuffer(g1.to_bytes96(pt), dtype=np.uint8)
    for width in (16, 64, 512):                    # lanes of 4 members; 16 lanes; one group across two workgroups' worth
        offsets = np.arange(0, n + 1, width, dtype=np.uint32)
        index = np.arange(n, dtype=np.uint32)
        e = pea.Engine()
        e.store_init(0, 0, b"\x01" * 32)
        e.set_validators(np.full(n, 32 * 10**9, dtype=np.uint64), np.ones(n, dtype=np.uint8), pts)
        out = e.g1_sum(offsets, index=index)      # pe_g1_sum over the registry: launch_g1_planned, the same kernels
        for c in range(offsets.size - 1):
            exp = g1.sum_points([p for p in pattern[offsets[c]:offsets[c + 1]] if p is not None])
            assert out[c].tobytes() == g1.to_bytes96(exp), (width, c)
        e.close()


def test_sum_over_caller_supplied_points():
    """pe_g1_sum over points that are not the registry: converted to the accumulation's form per call; a second call with
    fewer points must not see rows of the first."""
    rng = np.random.default_rng(5)
    e = pea.Engine()
    e.store_init(0, 0, b"\x02" * 32)
    for n in (300, 40):
        ks = [int(x) for x in rng.integers(1, 1 << 62, size=n)]
        P = [g1.mul(k, g1.G) for k in ks]
        pts = np.stack([np.frombuffer(g1.to_bytes96(p), dtype=np.uint8) for p in P])
        offsets = np.array([0, 1, 1, n // 2, n], dtype=np.uint32)
        out = e.g1_sum(offsets, points96=pts)
        for c in range(4):
            assert out[c].tobytes() == g1.to_bytes96(g1.sum_points(P[offsets[c]:offsets[c + 1]])), (n, c)
    e.close()
