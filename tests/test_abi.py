"""CPU: the C-ABI library loads and exports every symbol include/posevo.h declares; no compute without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(names=("posevo.h", "posevo_profile.h")):
    """Every entry point declared under include/: the boundary (posevo.h) and the measurement hooks (posevo_profile.h)."""
    out = set()
    for name in names:
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(pe_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def test_header_and_binding_agree():
    from pos_evolution_amd import _abi
    assert header_symbols() == sorted(_abi.SIGNATURES), "include/*.h and _abi.SIGNATURES list different entry points"
    assert not [n for n in header_symbols(("posevo.h",)) if n.startswith("pe_profile")]   # measurement is not boundary


def test_library_exports_every_symbol():
    from pos_evolution_amd import _abi
    lib = _abi.load()
    for name in header_symbols():
        assert hasattr(lib, name), name
    want = int(re.search(r"#define\s+PE_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "posevo.h")).read()).group(1))
    assert lib.pe_abi_version() == want   # the library was built from this header


def test_struct_layouts():
    from pos_evolution_amd import _abi, synth
    assert C.sizeof(_abi.pe_attestation) == 144 == synth.ATT_DTYPE.itemsize
    assert C.sizeof(_abi.pe_state_ctx) == 8 + 32 + 8 + 32 + 8 + 32 + 8
    for name in synth.ATT_DTYPE.names:
        assert synth.ATT_DTYPE.fields[name][1] == getattr(_abi.pe_attestation, name).offset


def test_config_defaults_are_the_mainnet_preset():
    from pos_evolution_amd import _abi
    lib = _abi.load()
    cfg = _abi.pe_config()
    lib.pe_config_default(C.byref(cfg))
    assert (cfg.slots_per_epoch, cfg.seconds_per_slot, cfg.intervals_per_slot) == (32, 12, 3)
    assert cfg.proposer_score_boost == 40 and cfg.effective_balance_increment == 10**9


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import pos_evolution_amd as pea
    with pytest.raises(pea.EngineError) as e:
        pea.Engine()
    assert e.value.status == _abi_no_device()


def _abi_no_device():
    from pos_evolution_amd import _abi
    return _abi.PE_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pos_evolution_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".inc")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "posevo_oracle" not in text, f


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/posevo.h is the drop-in boundary: it must compile as C99 and as C++ with nothing but <stdint.h>,
    and a C client must link against the library (the cgo / JNI / bindgen stubs of INTEGRATION.md rely on it)."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    src = tmp_path / "client.c"
    src.write_text('#include "posevo.h"\n'
                   'int main(void) {\n'
                   '    pe_config c; pe_engine* h = 0; int rc;\n'
                   '    pe_config_default(&c);\n'
                   '    if (sizeof(pe_attestation) != 144 || pe_abi_version() != PE_ABI_VERSION) return 2;\n'
                   '    rc = pe_engine_create(&c, &h);            /* no GPU here: must report PE_ERR_NO_DEVICE */\n'
                   '    if (rc == PE_OK) { pe_engine_destroy(h); return 0; }\n'
                   '    return rc == PE_ERR_NO_DEVICE ? 0 : 3;\n'
                   '}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", inc, "-x", "c++", "-fsyntax-only", str(src)])
    libdir = os.path.join(ROOT, "pos_evolution_amd")
    exe = tmp_path / "client"
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lposevo",
                           "-Wl,-rpath," + libdir])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    assert subprocess.call([str(exe)], env=env) == 0


def test_example_c_client_builds_against_the_header(tmp_path):
    """examples/step_client.c (the whole step through the C ABI) compiles warning-free as C99 against include/posevo.h
    and links against the library; without a GPU it must stop at pe_engine_create, not crash."""
    import subprocess
    libdir = os.path.join(ROOT, "pos_evolution_amd")
    exe = tmp_path / "step_client"
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Wextra", "-Werror", "-I",
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "step_client.c"), "-o", str(exe),
                           "-L", libdir, "-lposevo", "-Wl,-rpath," + libdir])
    wl = tmp_path / "w.bin"
    import struct
    wl.write_bytes(struct.pack("<8Q", 0x30764F5645534F50, 0, 32, 1, 0, 32, 0, 0) + bytes(32 + 4 + 8))
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe), str(wl), "sync"], env=env, capture_output=True, text=True)
    import torch
    if not torch.cuda.is_available():
        assert out.returncode == 1 and "pe_engine_create" in out.stderr
