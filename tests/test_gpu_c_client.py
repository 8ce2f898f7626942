"""-m gpu: a plain C99 program (examples/step_client.c) drives the engine through include/posevo.h alone -- synchronous
calls, pipelined calls, streaming pipelines -- and must produce, bit for bit, what the same steps give when driven from
Python with synchronous calls: the C ABI is the product, the Python package one client of it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def _fnv(hsh: int, data: bytes) -> int:
    for b in data:
        hsh = ((hsh ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return hsh


def _python_hash(e, w, with_sigs=False):
    """The client's fold() over the same steps, driven through the Python wrapper with synchronous calls."""
    hsh = 0xCBF29CE484222325
    last_sig = b""
    for st in w["steps"]:
        e.on_tick(st["tick"])
        e.participation_rotate()
        if with_sigs:
            agg = e.aggregate_signed(st["sigs"], packed=(st["atts"], st["arena"]), want_aggregate_pubkeys=True)
        else:
            agg = e.aggregate(packed=(st["atts"], st["arena"]), want_aggregate_pubkeys=True)
        rows, g = agg["atts"], agg["n_groups"]
        status, _, cnt = e.on_attestation_batch(packed=(rows, agg["out_arena"]))
        head = e.get_head()
        pst, num = e.process_attestation_batch(st["ctx"], packed=(rows, agg["out_arena"]))
        hsh = _fnv(hsh, np.uint32(g).tobytes())
        raw = np.ascontiguousarray(rows).view(np.uint8).reshape(g, 144)
        for k in range(g):
            hsh = _fnv(hsh, raw[k, :128].tobytes())
            hsh = _fnv(hsh, raw[k, 128:140].tobytes())
        hsh = _fnv(hsh, agg["out_arena"].tobytes())
        for arr, dt in ((agg["count"], np.uint32), (agg["aggpk96"], np.uint8), (status, np.int32), (cnt, np.uint32),
                        (pst, np.int32), (num, np.uint64)):
            hsh = _fnv(hsh, np.ascontiguousarray(arr, dtype=dt).tobytes())
        hsh = _fnv(hsh, head)
        if with_sigs:
            hsh = _fnv(hsh, np.ascontiguousarray(agg["sig96c"]).tobytes())
            last_sig = agg["sig96c"][0].tobytes()
    return (hsh, last_sig) if with_sigs else hsh


def test_c_client_equals_python_client(tmp_path):
    import make_workload

    e, w = make_workload.build(validators=6144, committees=64, blocks=200, steps=5, parts=3, density=0.9, rounds=10)
    path = str(tmp_path / "workload.bin")
    make_workload.write(path, w)
    exe = str(tmp_path / "step_client")
    lib_dir = os.path.join(ROOT, "pos_evolution_amd")
    subprocess.run(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "step_client.c"), "-L", lib_dir, "-lposevo", f"-Wl,-rpath,{lib_dir}",
                    "-o", exe], check=True, capture_output=True)
    want = _python_hash(e, w)
    e.close()
    for mode in ("sync", "pipelined", "streaming"):
        out = subprocess.run([exe, path, mode], check=True, capture_output=True, text=True, timeout=300)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        assert d["mode"] == mode and d["steps"] == 5 and d["attestations"] > 0
        assert int(d["hash"], 16) == want, mode


def test_c_client_obtains_the_compressed_aggregate_signatures(tmp_path):
    """pe_aggregate_signed from plain C: the workload carries one compressed BLSSignature per attestation (made with the
    oracle here, in the test); the client's aggregate signatures -- folded into its hash, the last step's first one printed
    -- equal the Python client's and the oracle's sum."""
    import make_workload
    from oracle import g2

    e, w = make_workload.build(validators=4096, committees=32, blocks=100, steps=3, parts=3, density=0.9, rounds=10)
    a, b = 0x5151, 0x77
    for s, st in enumerate(w["steps"]):
        pts = g2.synthetic_points(len(st["atts"]), a + s, b)
        st["sigs"] = np.frombuffer(b"".join(g2.compress(p) for p in pts), dtype=np.uint8).reshape(-1, 96)
    path = str(tmp_path / "workload_sigs.bin")
    make_workload.write(path, w)
    exe = str(tmp_path / "step_client")
    lib_dir = os.path.join(ROOT, "pos_evolution_amd")
    subprocess.run(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "step_client.c"), "-L", lib_dir, "-lposevo", f"-Wl,-rpath,{lib_dir}",
                    "-o", exe], check=True, capture_output=True)
    want, want_sig = _python_hash(e, w, with_sigs=True)
    # the oracle's word on that last signature: group 0 of the last step = the rows with group_of == 0
    st = w["steps"][-1]
    agg = e.aggregate(packed=(st["atts"], st["arena"]))
    rows0 = np.nonzero(np.asarray(agg["group_of"]) == 0)[0]
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    exp = g2.mul(((a + len(w["steps"]) - 1) * len(rows0) + b * int(rows0.sum())) % R, g2.G2)
    assert want_sig == g2.compress(exp)
    e.close()
    for mode in ("sync", "streaming"):
        out = subprocess.run([exe, path, mode, "0"], check=True, capture_output=True, text=True, timeout=300)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        assert int(d["hash"], 16) == want, mode
        assert bytes.fromhex(d["aggregate_signature"]) == want_sig, mode
