"""CPU: the host logic of the Python package above the C ABI, exercised against a STUB of libposevo.so generated from
include/posevo.h (every entry point present, nothing computed) -- marshalling, output-set ring, pipeline book-keeping,
error mapping, the pyspec-level guards of forkchoice.py.  Runs in a subprocess: the stub is selected with
POSEVO_LIB_PATH and must never be the library this test process has loaded."""
import os
import subprocess
import sys

from tests.test_abi import ROOT, header_symbols

SPECIAL = r'''
#include <stdint.h>
#include <string.h>
char stub_log[65536];
static int fail_next;
static uintptr_t last_arena;
static int dummy;
static void note(const char* n) { if (strlen(stub_log) + strlen(n) + 2 < sizeof stub_log) { strcat(stub_log, n); strcat(stub_log, " "); } }
void stub_reset(void) { stub_log[0] = 0; }
void stub_fail_next(int rc) { fail_next = rc; }
uintptr_t stub_last_arena(void) { return last_arena; }
static int ret(const char* n) { note(n); if (fail_next) { int r = fail_next; fail_next = 0; return r; } return 0; }
typedef struct { uint8_t data[128]; uint32_t bits_offset, n_bits, flags, reserved0; } att;
uint32_t pe_abi_version(void) { return 4; }
const char* pe_strerror(int s) { (void)s; return "stub"; }
const char* pe_last_error(const void* h) { (void)h; return "stub detail"; }
int pe_engine_create(const void* cfg, void** out) { (void)cfg; note("pe_engine_create"); *out = &dummy; return 0; }
void pe_engine_destroy(void* h) { (void)h; note("pe_engine_destroy"); }
void pe_config_default(void* c) { memset(c, 0, 96); ((uint64_t*)c)[0] = 32; }
uint32_t pe_num_blocks(const void* h) { (void)h; return 1; }
uint64_t pe_num_validators(const void* h) { (void)h; return 0; }
int pe_get_store_scalars(void* h, uint64_t* t, uint64_t* g, uint64_t* je, uint8_t* jr, uint64_t* fe, uint8_t* fr,
                         uint64_t* be, uint8_t* br, uint8_t* boost)
{ (void)h; *t = 0; *g = 0; *je = 7; memset(jr, 7, 32); *fe = 0; memset(fr, 0, 32); *be = 7; memset(br, 7, 32); memset(boost, 0, 32);
  return 0; }
int pe_aggregate(void* h, const att* a, uint32_t n, const uint8_t* arena, uint64_t alen, const uint8_t* sig, att* out,
                 uint32_t* ng, uint32_t* gof, uint8_t* obits, uint64_t ocap, uint8_t* osig, uint8_t* opk, uint32_t* cnt)
{ (void)h; (void)alen; (void)sig; (void)gof; (void)obits; (void)ocap; (void)osig; (void)opk;
  last_arena = (uintptr_t)arena;
  uint32_t g = n / 4;
  for (uint32_t i = 0; i < g; ++i) { out[i] = a[i]; out[i].bits_offset = 8 * i; out[i].n_bits = 64; cnt[i] = 1; }
  *ng = g; return ret("pe_aggregate"); }
'''


def test_host_wrapper_against_a_stub_library(tmp_path):
    have = {"pe_abi_version", "pe_strerror", "pe_last_error", "pe_engine_create", "pe_engine_destroy", "pe_config_default",
            "pe_num_blocks", "pe_num_validators", "pe_get_store_scalars", "pe_aggregate"}
    src = SPECIAL + "".join(f'int {name}() {{ return ret("{name}"); }}\n' for name in header_symbols() if name not in have)
    c = tmp_path / "stub.c"
    c.write_text(src)
    so = tmp_path / "libposevo_stub.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-w", "-o", str(so), str(c)])
    env = dict(os.environ, POSEVO_LIB_PATH=str(so), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "native", "host_wrapper_check.py")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host wrapper ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
