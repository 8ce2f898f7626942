"""CPU: the engine's safegcd inversion (pos_evolution_amd/csrc/fp_inv_safegcd.h, plain C++ path shared by host
and device) against Python's pow(x, -1, p)."""
import ctypes as C
import os
import random
import subprocess

import numpy as np

from oracle import g1

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib():
    src = os.path.join(HERE, "native", "host_safegcd.cpp")
    out = os.path.join(HERE, "native", "libhost_safegcd.so")
    hdr = os.path.join(HERE, "..", "pos_evolution_amd", "csrc", "fp_inv_safegcd.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src])
    return C.CDLL(out)


def test_safegcd_matches_python_pow():
    lib = _lib()
    P = g1.P
    random.seed(2)
    vals = [1, 2, 3, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 2**380] + [random.randrange(1, P) for _ in range(20000)]
    vals += [random.randrange(1, 2**k) for k in range(1, 381) for _ in range(2)]
    x = np.array([[(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)] for v in vals], dtype=np.uint32)
    out = np.zeros_like(x)
    worst = lib.host_modinv_many(out.ctypes.data_as(C.POINTER(C.c_uint32)), x.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 len(vals))
    for v, o in zip(vals, out):
        assert sum(int(w) << (32 * i) for i, w in enumerate(o)) == pow(v, -1, P)
    assert worst <= 30   # half-delta variant: well inside the 37 batches Theorem 11.2 allows the original (1101 divsteps)
    lib.host_modinv_total_batches.restype = C.c_long
    rnd = x[8:20008]
    total = lib.host_modinv_total_batches(rnd.ctypes.data_as(C.POINTER(C.c_uint32)), len(rnd))
    assert total / len(rnd) < 26.5, total / len(rnd)   # ~26.0 batches on random inputs (26.9 with delta = 1)
