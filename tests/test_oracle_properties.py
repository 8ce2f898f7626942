"""CPU, hypothesis-driven (SURVEY.md 4 plan (a), 5 "race detection / sanitizers"):
* the C oracle's get_head against a definition-level restatement (pe:322 subtree weights, pe:1107-1116 descent,
  A.3 viability) on generated trees / votes / balances / flags;
* update_latest_messages' batch-order semantics (pe:1440: strictly later epoch wins, first in order among equals)
  against the sequential loop;
* the same C source under -fsanitize=address,undefined."""
import os
import subprocess
import sys

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import cport

NONE32 = 0xFFFFFFFF
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def definition_get_head(parent, leaf_ok, roots, vote, bal, flags, justified, boost, boost_percent=40, spe=32):
    """Straight from the text: weight(B) = sum of balances whose latest message is B or a descendant (+ boost on every
    ancestor-or-self of the boosted block); viable(B) = some descendant-or-self leaf passes the leaf test; descend from
    the justified root to the heaviest viable child, ties to the lexicographically higher root."""
    n = len(parent)
    children = [[] for _ in range(n)]
    for i in range(1, n):
        children[parent[i]].append(i)
    counted = [(flags[v] & 1) and not (flags[v] & 4) and vote[v] != NONE32 for v in range(len(vote))]
    w = [0] * n
    for v in range(len(vote)):
        if counted[v]:
            b = int(vote[v])
            while b != NONE32:
                w[b] += int(bal[v])
                b = int(parent[b]) if b else NONE32
    if boost != NONE32:
        act = [v for v in range(len(vote)) if flags[v] & 1]
        if act:
            total = max(10**9, sum(int(bal[v]) for v in act))
            score = ((len(act) // spe) * (total // len(act)) * boost_percent) // 100
            b = boost
            while b != NONE32:
                w[b] += score
                b = int(parent[b]) if b else NONE32

    def viable(b):
        return bool(leaf_ok[b]) if not children[b] else any(viable(c) for c in children[b])

    head = justified
    while True:
        cand = [c for c in children[head] if viable(c)]
        if not cand:
            return head, w
        head = max(cand, key=lambda c: (w[c], bytes(roots[c])))


@st.composite
def worlds(draw):
    n = draw(st.integers(1, 40))
    parent = [NONE32] + [draw(st.integers(0, i - 1)) for i in range(1, n)]
    n_val = draw(st.integers(0, 60))
    vote = [draw(st.one_of(st.just(NONE32), st.integers(0, n - 1))) for _ in range(n_val)]
    bal = [draw(st.integers(1, 64)) * 10**9 for _ in range(n_val)]
    flags = [draw(st.sampled_from([0, 1, 1, 1, 3, 5])) for _ in range(n_val)]
    leaf_ok = [draw(st.sampled_from([1, 1, 1, 0])) for _ in range(n)]
    leaf_ok[0] = 1
    roots = [bytes(draw(st.binary(min_size=32, max_size=32))) for _ in range(n)]
    boost = draw(st.one_of(st.just(NONE32), st.integers(0, n - 1)))
    return parent, leaf_ok, roots, vote, bal, flags, boost


@settings(max_examples=150, deadline=None)
@given(worlds())
def test_c_oracle_get_head_equals_the_definition(wd):
    parent, leaf_ok, roots, vote, bal, flags, boost = wd
    if len(set(roots)) != len(roots):
        return
    head_d, w_d = definition_get_head(parent, leaf_ok, roots, vote, bal, flags, 0, boost)
    r = np.frombuffer(b"".join(roots), dtype=np.uint8).reshape(-1, 32)
    head_c, w_c = cport.get_head(np.array(parent, dtype=np.uint32), np.array(leaf_ok, dtype=np.uint8), r,
                                 np.array(vote, dtype=np.uint32), np.array(bal, dtype=np.uint64),
                                 np.array(flags, dtype=np.uint8), 0, boost)
    assert [int(x) for x in w_c] == w_d
    assert head_c == head_d


@settings(max_examples=100, deadline=None)
@given(st.integers(1, 6), st.integers(1, 12), st.data())
def test_c_oracle_lmd_batch_order(n_comm, n_rows, data):
    """A batch applied by the C oracle == the spec's loop over the attestations in batch order."""
    sizes = [data.draw(st.integers(0, 9)) for _ in range(n_comm)]
    n_val = sum(sizes)
    members = np.arange(n_val, dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    val_flags = np.array([data.draw(st.sampled_from([1, 1, 1, 5])) for _ in range(n_val)], dtype=np.uint8)
    rows = []
    for _ in range(n_rows):
        c = data.draw(st.integers(0, n_comm - 1))
        bits = [data.draw(st.booleans()) for _ in range(sizes[c])]
        rows.append((c, bits, data.draw(st.integers(0, 3)), data.draw(st.integers(0, 7))))
    vote_epoch = np.zeros(max(n_val, 1), dtype=np.uint64)
    vote_block = np.full(max(n_val, 1), NONE32, dtype=np.uint32)
    want_e, want_b = {}, {}
    for c, bits, ep, blk in rows:                       # pe:1435-1441, literally
        for i, bit in enumerate(bits):
            v = int(members[offs[c] + i])
            if bit and not (val_flags[v] & 4) and (v not in want_e or ep > want_e[v]):
                want_e[v], want_b[v] = ep, blk
    arena_rows = [np.asarray(bits, dtype=np.uint8) for _, bits, _, _ in rows]
    chunks, boffs, off = [], [], 0
    for b in arena_rows:
        p = np.packbits(b, bitorder="little") if b.size else np.zeros(0, dtype=np.uint8)
        boffs.append(off)
        chunks.append(p)
        off += p.size
    arena = np.concatenate(chunks + [np.zeros(1, dtype=np.uint8)])
    cport.update_latest_messages(np.array([offs[c] for c, *_ in rows], dtype=np.uint32),
                                 np.array([len(b) for _, b, _, _ in rows], dtype=np.uint32),
                                 np.array(boffs, dtype=np.uint32), np.array([e for *_, e, _ in rows], dtype=np.uint64),
                                 np.array([b for *_, b in rows], dtype=np.uint32), arena, members, val_flags,
                                 vote_epoch, vote_block)
    for v in range(n_val):
        if v in want_e:
            assert (int(vote_epoch[v]), int(vote_block[v])) == (want_e[v], want_b[v]), v
        else:
            assert int(vote_block[v]) == NONE32, v


def test_c_oracle_under_address_and_undefined_sanitizers(tmp_path):
    """The same C source built with -fsanitize=address,undefined runs the oracle test-suite's core calls cleanly."""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.exists(asan):
        import pytest
        pytest.skip("libasan not installed")
    lib = tmp_path / "libposevo_oracle_asan.so"
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fPIC",
                           "-std=c11", "-shared", "-o", str(lib), os.path.join(ROOT, "oracle", "posevo_oracle.c")])
    script = (
        "import numpy as np\n"
        "from oracle import cport, g1\n"
        "import pos_evolution_amd.synth as synth\n"
        "t = synth.random_tree(300, 3, 'bushy')\n"
        "v = synth.zipf_votes(5000, 300, 3)\n"
        "b = synth.balances(5000, 3, True)\n"
        "f = synth.validator_flags(5000, 3)\n"
        "h, w = cport.get_head(t.parent, np.ones(300, np.uint8), t.roots, v, b, f, 0, 299)\n"
        "G = np.frombuffer(g1.to_bytes96(g1.G), dtype=np.uint8)\n"
        "pts = cport.g1_arith_progression(G.tobytes(), G.tobytes(), 200)\n"
        "s = cport.g1_sum_groups(pts, np.arange(200, dtype=np.uint32), np.array([0, 50, 50, 200], dtype=np.uint32))\n"
        "assert s[0].tobytes() == g1.to_bytes96(g1.mul(50 * 51 // 2, g1.G))\n"
        "print('sanitized ok', h, int(w[0]))\n")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", POSEVO_ORACLE_LIB=str(lib), PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", script], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "sanitized ok" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])
