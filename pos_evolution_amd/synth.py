"""Seeded synthetic workloads of the BASELINE.json shapes (numpy only; no oracle, no engine).

Used by bench.py, __graft_entry__.smoke() and the tests to build identical inputs for the engine
and for the checker.  Shapes follow SURVEY.md 8(d):

    validators  effective balances (Gwei, multiples of the increment), activity/slashed flags
    block tree  roots (sha256 of a counter), parent index < child index, strictly increasing slots
    committees  a random partition of the validators into C committees (slot-major ids)
    votes       latest-message block per validator, Zipf over the most recent blocks
    batches     pe_attestation rows + LSB-first bit arena as numpy structured arrays
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import Optional

import numpy as np

ATT_DTYPE = np.dtype([
    ("slot", "<u8"), ("index", "<u8"), ("beacon_block_root", "u1", (32,)),
    ("source_epoch", "<u8"), ("source_root", "u1", (32,)),
    ("target_epoch", "<u8"), ("target_root", "u1", (32,)),
    ("bits_offset", "<u4"), ("n_bits", "<u4"), ("flags", "<u4"), ("reserved0", "<u4"),
])
assert ATT_DTYPE.itemsize == 144

GWEI_PER_ETH = 10**9
NONE32 = 0xFFFFFFFF


def make_roots(n: int, salt: bytes = b"posevo") -> np.ndarray:
    """(n, 32) uint8: sha256(salt || counter) -- opaque block ids."""
    out = np.empty((n, 32), dtype=np.uint8)
    for i in range(n):
        out[i] = np.frombuffer(hashlib.sha256(salt + i.to_bytes(8, "little")).digest(), dtype=np.uint8)
    return out


@dataclass
class Tree:
    roots: np.ndarray    # (B, 32) u8
    parent: np.ndarray   # (B,) u32, parent[0] = NONE32
    slot: np.ndarray     # (B,) u64, strictly increasing along parent links


def random_tree(n_blocks: int, seed: int, kind: str = "branchy", anchor_slot: int = 0) -> Tree:
    """kind: 'chain' (depth = B: the LDS scan stress of config 3), 'branchy' (main chain with
    geometric side branches, config 2), 'bushy' (uniform random parent among the last 64 blocks)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parent = np.full(n_blocks, NONE32, dtype=np.uint32)
    slot = np.zeros(n_blocks, dtype=np.uint64)
    slot[0] = anchor_slot
    if kind == "chain":
        parent[1:] = np.arange(n_blocks - 1, dtype=np.uint32)
        slot[1:] = anchor_slot + np.arange(1, n_blocks, dtype=np.uint64)
    else:
        tip = 0
        for i in range(1, n_blocks):
            if kind == "branchy":
                p = tip if rng.random() > 0.1 else int(rng.integers(max(0, i - 16), i))
            else:
                p = int(rng.integers(max(0, i - 64), i))
            parent[i] = p
            slot[i] = slot[p] + 1 + (1 if rng.random() < 0.05 else 0)  # occasional skipped slot
            if kind == "branchy" and p == tip:
                tip = i
    return Tree(make_roots(n_blocks, b"blk%d" % seed), parent, slot)


def balances(n: int, seed: int, mixed: bool = False, increment: int = GWEI_PER_ETH) -> np.ndarray:
    """32 ETH flat, or the config-5 mix: 70 % at 32 ETH, 30 % uniform integer ETH in [32, 2048]."""
    if not mixed:
        return np.full(n, 32 * increment, dtype=np.uint64)
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    eth = np.where(rng.random(n) < 0.7, 32, rng.integers(32, 2049, size=n)).astype(np.uint64)
    return eth * np.uint64(increment)


def validator_flags(n: int, seed: int, inactive_frac: float = 0.0, slashed_frac: float = 0.0) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed + 2000))
    f = np.full(n, 0x01, dtype=np.uint8)
    if inactive_frac:
        f[rng.random(n) < inactive_frac] &= 0xFE
    if slashed_frac:
        f[rng.random(n) < slashed_frac] |= 0x02
    return f


def zipf_votes(n_val: int, n_blocks: int, seed: int, participation: float = 0.99, recent: int = 64) -> np.ndarray:
    """Latest-message block index per validator (NONE32 = none): Zipf(1.2) rank over the `recent`
    most recently inserted blocks."""
    rng = np.random.Generator(np.random.PCG64(seed + 3000))
    r = min(recent, n_blocks)
    rank = np.minimum(rng.zipf(1.2, size=n_val) - 1, r - 1)
    vote = (n_blocks - 1 - rank).astype(np.uint32)
    vote[rng.random(n_val) >= participation] = NONE32
    return vote


@dataclass
class Committees:
    offsets: np.ndarray  # (C+1,) u32
    members: np.ndarray  # (V,) u32: committee c = members[offsets[c]:offsets[c+1]]


def random_committees(n_val: int, n_committees: int, seed: int, lo: int = 0) -> Committees:
    """Random partition of validators [lo, lo + n_val) into n_committees near-equal committees
    (compute_committee's slicing rule, pe:502-503)."""
    rng = np.random.Generator(np.random.PCG64(seed + 4000))
    perm = (rng.permutation(n_val) + lo).astype(np.uint32)
    idx = np.arange(n_committees + 1, dtype=np.uint64)
    offsets = ((np.uint64(n_val) * idx) // np.uint64(n_committees)).astype(np.uint32)
    return Committees(offsets, perm)


def ancestor_at(tree: Tree, idx: int, slot: int) -> int:
    """get_ancestor on the synthetic tree."""
    while tree.slot[idx] > slot and tree.parent[idx] != NONE32:
        idx = int(tree.parent[idx])
    return idx


def pack_bit_rows(bit_rows) -> tuple:
    """list of bool arrays -> (arena u8, byte offsets u32, n_bits u32); LSB-first, byte aligned per row."""
    offs, nb, chunks, off = [], [], [], 0
    for b in bit_rows:
        b = np.asarray(b, dtype=np.uint8)
        p = np.packbits(b, bitorder="little") if b.size else np.zeros(0, dtype=np.uint8)
        offs.append(off)
        nb.append(b.size)
        chunks.append(p)
        off += p.size
    arena = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    if arena.size == 0:
        arena = np.zeros(1, dtype=np.uint8)
    return np.ascontiguousarray(arena), np.asarray(offs, dtype=np.uint32), np.asarray(nb, dtype=np.uint32)


def epoch_attestations(comm: Committees, tree: Tree, epoch: int, slots_per_epoch: int, seed: int,
                       density: float = 0.99, parts: int = 1, source=(0, None), vote_recent: int = 8,
                       from_block: bool = False, vote_seed: Optional[int] = None):
    """One epoch's attestations: committee c attests in slot epoch*SPE + c // cps with index c % cps,
    voting for one of the `vote_recent` most recent blocks whose slot is <= its slot.  Each committee
    contributes `parts` partial aggregates with disjoint random bit subsets (union density `density`).

    Returns (atts ATT_DTYPE[n], arena u8, bit_rows list) with n = C * parts."""
    rng = np.random.Generator(np.random.PCG64(seed + 5000 + epoch))
    # head votes may need to be the same on every shard of a committee: separate stream
    vrng = rng if vote_seed is None else np.random.Generator(np.random.PCG64(vote_seed + 7000 + epoch))
    n_comm = comm.offsets.size - 1
    cps = n_comm // slots_per_epoch
    assert cps * slots_per_epoch == n_comm
    n = n_comm * parts
    atts = np.zeros(n, dtype=ATT_DTYPE)
    src_root = tree.roots[0] if source[1] is None else np.frombuffer(source[1], dtype=np.uint8)
    bit_rows = []
    order = np.argsort(tree.slot, kind="stable")
    sorted_slots = tree.slot[order]
    k = 0
    for c in range(n_comm):
        size = int(comm.offsets[c + 1] - comm.offsets[c])
        slot = epoch * slots_per_epoch + c // cps
        # candidate head votes: blocks with slot <= attestation slot, among the most recent
        hi = int(np.searchsorted(sorted_slots, slot, side="right"))
        cand = order[max(0, hi - vote_recent):hi]
        on = rng.random(size) < density
        part_of = rng.integers(0, parts, size=size)
        blk = int(cand[vrng.integers(0, cand.size)]) if cand.size else 0
        target_idx = ancestor_at(tree, blk, epoch * slots_per_epoch)  # FFG target consistent with the LMD vote
        for p in range(parts):
            bits = on & (part_of == p)
            a = atts[k]
            a["slot"], a["index"] = slot, c % cps
            a["beacon_block_root"] = tree.roots[blk]
            a["source_epoch"], a["source_root"] = source[0], src_root
            a["target_epoch"], a["target_root"] = epoch, tree.roots[target_idx]
            a["flags"] = 3 if from_block else 1
            bit_rows.append(bits)
            k += 1
    arena, offs, nb = pack_bit_rows(bit_rows)
    atts["bits_offset"], atts["n_bits"] = offs, nb
    return atts, arena, bit_rows


# ---------------------------------------------------------------------------------------------
# Synthetic pubkey registry P_v = A + v*B, built WITHOUT the oracle: a few hundred pure-Python
# affine additions seed three small tables, the engine's own G1 kernel does the bulk pair sums.
# (Subset sums of such a registry have the closed form |S|*A + (sum v)*B -- SURVEY.md 8c.)
# ---------------------------------------------------------------------------------------------
_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_G = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
      0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)


def _ec_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if (y1 + y2) % _P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, _P) % _P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, _P) % _P
    x3 = (lam * lam - x1 - x2) % _P
    return (x3, (lam * (x1 - x3) - y1) % _P)


def _ec_mul(k, p):
    acc = None
    while k:
        if k & 1:
            acc = _ec_add(acc, p)
        p = _ec_add(p, p)
        k >>= 1
    return acc


def _enc96(p) -> bytes:
    if p is None:
        return bytes([0x40]) + bytes(95)
    return p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")


def _progression(start, step, n):
    out, cur = [], start
    for _ in range(n):
        out.append(cur)
        cur = _ec_add(cur, step)
    return out


def registry_points(engine, n: int, a: int = 0x1234567, b: int = 0x89ABCDE, lo: int = 0) -> np.ndarray:
    """(n, 96) uint8, row v = A + (lo + v)*B with A = a*G, B = b*G.
    v = v0 + 256*v1 + 65536*v2  ->  P_v = (A + v0*B) + (v1*256*B) + (v2*65536*B)."""
    A, B = _ec_mul(a, _G), _ec_mul(b, _G)
    hi_n = (lo + n + 65535) // 65536
    t0 = _progression(A, B, 256)
    t1 = _progression(None, _ec_mul(256, B), 256)
    t2 = _progression(None, _ec_mul(65536, B), hi_n)
    table = np.frombuffer(b"".join(_enc96(p) for p in t0 + t1 + t2), dtype=np.uint8).reshape(-1, 96)
    out = np.empty((n, 96), dtype=np.uint8)
    chunk = 1 << 20
    for base in range(0, n, chunk):
        m = min(chunk, n - base)
        v = np.arange(lo + base, lo + base + m, dtype=np.uint64)
        idx = np.empty((m, 3), dtype=np.uint32)
        idx[:, 0] = v & 255
        idx[:, 1] = 256 + ((v >> 8) & 255)
        idx[:, 2] = 512 + (v >> 16)
        offsets = np.arange(0, 3 * m + 1, 3, dtype=np.uint32)
        out[base:base + m] = engine.g1_sum(offsets, index=idx.reshape(-1), points96=table)
    return out


def registry_closed_form_cs(count: int, index_sum: int, a: int = 0x1234567, b: int = 0x89ABCDE) -> bytes:
    """registry_closed_form from the number of indices and their sum alone (what ranks of a sharded run exchange)."""
    r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    return _enc96(_ec_mul((int(count) * a + int(index_sum) * b) % r, _G))


def registry_closed_form(indices, a: int = 0x1234567, b: int = 0x89ABCDE) -> bytes:
    """96-byte encoding of sum_{v in indices} (A + v*B) by one double-and-add (pure Python)."""
    r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    idx = [int(i) for i in indices]
    return _enc96(_ec_mul((len(idx) * a + sum(idx) * b) % r, _G))


# ---------------------------------------------------------------------------------------------
# Synthetic BLSSignatures S_i = (a + i*b) * G2 for the signature leg of the aggregation (pe:659, pe:717), built the
# same way: two short pure-Python progressions on the twist E'/Fp2: y^2 = x^3 + 4(1 + u), the engine's G2 sum for the
# pairs.  A group's aggregate signature then has the closed form (|S|*a + b*sum(i)) * G2.
# ---------------------------------------------------------------------------------------------
_G2 = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
       (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))


def _f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % _P, (a[0] * b[1] + a[1] * b[0]) % _P)


def _f2_inv(a):
    t = pow(a[0] * a[0] + a[1] * a[1], -1, _P)
    return (a[0] * t % _P, -a[1] * t % _P)


def _ec2_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if ((y1[0] + y2[0]) % _P, (y1[1] + y2[1]) % _P) == (0, 0):
            return None
        xx = _f2_mul(x1, x1)
        lam = _f2_mul((3 * xx[0] % _P, 3 * xx[1] % _P), _f2_inv((2 * y1[0] % _P, 2 * y1[1] % _P)))
    else:
        lam = _f2_mul(((y2[0] - y1[0]) % _P, (y2[1] - y1[1]) % _P), _f2_inv(((x2[0] - x1[0]) % _P, (x2[1] - x1[1]) % _P)))
    ll = _f2_mul(lam, lam)
    x3 = ((ll[0] - x1[0] - x2[0]) % _P, (ll[1] - x1[1] - x2[1]) % _P)
    t = _f2_mul(lam, ((x1[0] - x3[0]) % _P, (x1[1] - x3[1]) % _P))
    return (x3, ((t[0] - y1[0]) % _P, (t[1] - y1[1]) % _P))


def _ec2_mul(k, p):
    acc = None
    while k:
        if k & 1:
            acc = _ec2_add(acc, p)
        p = _ec2_add(p, p)
        k >>= 1
    return acc


def _enc192(p) -> bytes:
    """ZCash order: x.c1 || x.c0 || y.c1 || y.c0 (include/posevo.h, PE_SIG_G2_UNCOMPRESSED)."""
    if p is None:
        return bytes([0x40]) + bytes(191)
    (x0, x1), (y0, y1) = p
    return x1.to_bytes(48, "big") + x0.to_bytes(48, "big") + y1.to_bytes(48, "big") + y0.to_bytes(48, "big")


def signature_points(engine, n: int, a: int = 0xABCDEF12345, b: int = 0x1357) -> np.ndarray:
    """(n, 96) uint8: row i = the compressed BLSSignature (a + i*b) * G2.  i = i0 + 128*i1, n <= 16384."""
    assert n <= 128 * 128
    A, B = _ec2_mul(a, _G2), _ec2_mul(b, _G2)
    t0, cur = [], A
    for _ in range(128):
        t0.append(cur)
        cur = _ec2_add(cur, B)
    t1, cur, step = [], None, _ec2_mul(128, B)
    for _ in range(128):
        t1.append(cur)
        cur = _ec2_add(cur, step)
    table = np.frombuffer(b"".join(_enc192(p) for p in t0 + t1), dtype=np.uint8).reshape(-1, 192)
    i = np.arange(n, dtype=np.uint32)
    idx = np.stack([i & 127, 128 + (i >> 7)], axis=1).reshape(-1)
    pts = engine.g2_sum(table, np.arange(0, 2 * n + 1, 2, dtype=np.uint32), index=idx)
    return engine.g2_compress(pts)


def signature_closed_form(rows, a: int = 0xABCDEF12345, b: int = 0x1357):
    """The affine point sum_{i in rows} (a + i*b) * G2 (x, y as (c0, c1) pairs; None = infinity), pure Python."""
    r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    rows = [int(i) for i in rows]
    return _ec2_mul((len(rows) * a + sum(rows) * b) % r, _G2)
