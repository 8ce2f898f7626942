"""The resource signatures the streaming G1 chain's placement rules rest on, read from the compiler's
kernel-resource-usage remarks of the in-tree build (pos_evolution_amd/csrc/g1_kernels.resource.log, written by `make`).

A SIMD has 512 registers per lane, a CU 160 KB of LDS.  Registers are allocated in blocks of 8.
  * k_g1_accumulate + k_g1_tree / k_g1_tree_solo must fit one SIMD together (the tree of step N - 1 runs beside the
    accumulation of step N);
  * two k_g1_tree_solo workgroups must NOT fit one SIMD (that kernel has no LDS padding to keep them apart);
  * the accumulation's exclusive LDS request must exceed half a CU and leave room for every guest kernel's own LDS.
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pos_evolution_amd", "csrc")
LOG = os.path.join(CSRC, "g1_kernels.resource.log")


def _kernels(path):
    out = {}
    text = open(path).read()
    for blk in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = blk.split()[0]

        def num(key):
            m = re.search(re.escape(key) + r": (\d+)", blk)
            return int(m.group(1)) if m else None
        out[name] = dict(vgpr=num("VGPRs"), agpr=num("AGPRs"), scratch=num("ScratchSize [bytes/lane]"),
                         lds=num("LDS Size [bytes/block]"))
    return out


def _alloc(k):  # unified register file: architectural registers rounded up to 4, + accumulation registers, in blocks of 8
    total = (k["vgpr"] + 3) // 4 * 4 + (k["agpr"] or 0)
    return (total + 7) // 8 * 8


def _find(ks, needle):
    hits = [v for n, v in ks.items() if needle in n]
    assert len(hits) == 1, (needle, list(ks))
    return hits[0]


@pytest.mark.skipif(not os.path.exists(LOG), reason="the library has not been built here (make writes the log)")
def test_g1_chain_register_signatures():
    ks = _kernels(LOG)
    acc = _find(ks, "15k_g1_accumulate")
    tree = _find(ks, "9k_g1_treeE")
    solo = _find(ks, "14k_g1_tree_solo")
    fin = _find(ks, "11k_g1_finish")
    assert acc["scratch"] == 0 and acc["lds"] == 0, "the accumulation's loop must not spill and uses no LDS of its own"
    assert _alloc(acc) <= 256, "two accumulation waves per SIMD must stay possible (synchronous calls launch 131 072 lanes)"
    assert _alloc(acc) + _alloc(tree) <= 512, "k_g1_tree no longer fits beside an accumulation wave"
    assert _alloc(acc) + _alloc(solo) <= 512, "k_g1_tree_solo no longer fits beside an accumulation wave"
    assert 2 * _alloc(solo) > 512, "two k_g1_tree_solo waves fit one SIMD: nothing keeps its workgroups one per CU"
    assert _alloc(acc) + _alloc(fin) <= 512, "k_g1_finish no longer fits beside an accumulation wave"


def test_exclusive_lds_request_leaves_room_for_the_guests():
    hdr = open(os.path.join(CSRC, "kernels.h")).read()
    m = re.search(r"G1_ACC_EXCLUSIVE_LDS = (\d+) \* 1024", hdr)
    assert m, "kernels.h: G1_ACC_EXCLUSIVE_LDS"
    req = int(m.group(1)) * 1024
    cu = 160 * 1024
    assert 2 * req > cu, "two accumulation workgroups would fit one CU"
    left = cu - req
    tree_lds = (56 + 2) * 256 * 4                      # k_g1_tree(_solo): 56 words per S29 partial + 2 of block info, 256 lanes
    fc_tree_4096 = 8 * (4096 + 18) + 4 * (2 * 4096 + 16)  # fc_kernels.hip tree_lds_bytes(4096): the paired union | tree launch
    for name, need in (("k_g1_tree_solo", tree_lds), ("k_tree at 4096 blocks", fc_tree_4096)):
        assert need <= left, f"{name} ({need} B of LDS) finds no room beside an exclusive accumulation ({left} B left)"


def test_sweep_knobs_name_switches_the_engine_reads():
    """tools/sweep.py's variants set environment switches; every POSEVO_* name it sets must be one the handle reads
    (engine_internal.h, Tune) -- a stale name would make the sweep compare the default with itself."""
    import ast

    tree = ast.parse(open(os.path.join(ROOT, "tools", "sweep.py")).read())
    knobs = None
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", None) == "KNOBS":
            knobs = ast.literal_eval(node.value)
    assert knobs, "tools/sweep.py: KNOBS"
    hdr = open(os.path.join(CSRC, "engine_internal.h")).read()
    read = set(re.findall(r'env\("(POSEVO_[A-Z_]+)"', hdr))
    for name, (env, _args, _group) in knobs.items():
        for var in env:
            if var.startswith("POSEVO_"):
                assert var in read, f"sweep knob {name} sets {var}, which no handle reads"
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for var in read:
        assert var in design, f"{var} is read by the engine but missing from DESIGN.md's table of run-time knobs"
