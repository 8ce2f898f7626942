"""CPU: pos_evolution_amd/csrc/fp381_mul.inc -- the generated body of the device's Montgomery product (inline-assembly
columns of v_mad_u64_u32 + v_addc_co_u32, tools/gen_fp_mul.py) -- INTERPRETED instruction by instruction in Python and
held against Python integers.  The GPU parity tests exercise the compiled kernel; this one pins the generated text itself,
so a change to the generator (a dedicated squaring, a lazy [0, 2p) form) can be developed and checked without a GPU.

Instruction semantics modelled (gfx950 ISA):
    v_mad_u64_u32 D(64), vcc, A(32), B(32), C(64):  D = (A * B + C) mod 2^64, vcc = carry out of bit 63
    v_addc_co_u32 D, vcc, 0, S, vcc:                 D = (S + vcc) mod 2^32, vcc = carry out"""
import os
import random
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "pos_evolution_amd", "csrc", "fp381_mul.inc")
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FP_N0 = 0xFFFCFFFD          # -p^-1 mod 2^32 (fp381.h)
M32, M64 = (1 << 32) - 1, (1 << 64) - 1


def _statements(text):
    text = re.sub(r"//[^\n]*", "", text)
    out, depth, cur, in_str = [], 0, "", False
    for ch in text:
        if ch == '"':
            in_str = not in_str
        if not in_str:
            depth += ch == "("
            depth -= ch == ")"
        if ch == ";" and depth == 0 and not in_str:
            out.append(" ".join(cur.split()))
            cur = ""
        else:
            cur += ch
    return [s for s in out if s]


class Machine:
    def __init__(self, a, b):
        self.env = {"lo": 0, "hi": 0, "br": 0}
        self.a, self.b, self.r = a, b, [None] * 12
        self.mads = self.addcs = 0

    def get(self, expr):
        m = re.fullmatch(r"([abr])\.l\[(\d+)\]", expr)
        if m:
            return {"a": self.a, "b": self.b, "r": self.r}[m.group(1)][int(m.group(2))]
        return self.env[expr]

    def asm(self, stmt):
        body = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', stmt.split(":")[0]))
        ins = [s.strip() for s in body.replace("\\t", "").split("\\n") if s.strip()]
        operands = re.findall(r'"[+=]?v"\(([^)]+)\)', stmt)
        assert operands[:2] == ["lo", "hi"] and stmt.rstrip().endswith('"vcc")')
        vcc = 0
        for line in ins:
            op, args = line.split(None, 1)
            args = [x.strip() for x in args.split(",")]
            if op == "v_mad_u64_u32":
                assert args[0] == "%0" and args[1] == "vcc" and args[4] == "%0"
                x, y = (self.get(operands[int(t[1:])]) for t in args[2:4])
                assert 0 <= x <= M32 and 0 <= y <= M32
                full = x * y + self.env["lo"]
                self.env["lo"], vcc = full & M64, full >> 64
                self.mads += 1
            elif op == "v_addc_co_u32":
                assert args == ["%1", "vcc", "0", "%1", "vcc"]
                full = self.env["hi"] + vcc
                self.env["hi"], vcc = full & M32, full >> 32
                assert vcc == 0, "the third accumulator word overflowed"
                self.addcs += 1
            else:
                raise AssertionError(f"instruction not modelled: {line}")

    def run(self, stmts):
        for s in stmts:
            if s.startswith("asm("):
                self.asm(s)
            elif re.fullmatch(r"uint32_t (m|t)\d+(, (m|t)\d+)*", s) or s in ("uint64_t lo = 0", "uint32_t hi = 0", "uint32_t br = 0"):
                continue
            elif s.startswith("uint32_t p0 = "):
                for k in range(12):
                    self.env[f"p{k}"] = (P >> (32 * k)) & M32
                assert all(f"p{k} = fp_p_limb({k})" in s for k in range(12))
            elif re.fullmatch(r"m\d+ = \(uint32_t\)lo \* FP_N0", s):
                self.env[s.split()[0]] = ((self.env["lo"] & M32) * FP_N0) & M32
            elif s == "lo = (lo >> 32) | ((uint64_t)hi << 32)":
                self.env["lo"] = (self.env["lo"] >> 32) | (self.env["hi"] << 32)
            elif s == "hi = 0":
                self.env["hi"] = 0
            elif re.fullmatch(r"t\d+ = \(uint32_t\)lo", s):
                self.env[s.split()[0]] = self.env["lo"] & M32
            elif s == "t12 = (uint32_t)(lo >> 32)":
                self.env["t12"] = (self.env["lo"] >> 32) & M32
            elif re.fullmatch(r"const uint32_t s\d+ = __builtin_subc\(t\d+, p\d+, br, &br\)", s):
                d, t, p = re.findall(r"\b[stp]\d+\b", s)
                v = self.env[t] - self.env[p] - self.env["br"]
                self.env[d], self.env["br"] = v & M32, 1 if v < 0 else 0
            elif s == "const bool ge = (t12 != 0) || (br == 0)":
                self.env["ge"] = self.env["t12"] != 0 or self.env["br"] == 0
            elif re.fullmatch(r"r\.l\[\d+\] = ge \? s\d+ : t\d+", s):
                j = int(re.search(r"\[(\d+)\]", s).group(1))
                self.r[j] = self.env[f"s{j}"] if self.env["ge"] else self.env[f"t{j}"]
            else:
                raise AssertionError(f"statement not modelled: {s[:120]}")
        return sum(v << (32 * j) for j, v in enumerate(self.r))


def _limbs(v):
    return [(v >> (32 * j)) & M32 for j in range(12)]


def test_generated_product_is_the_montgomery_product():
    stmts = _statements(open(INC).read())
    rinv = pow(1 << 384, -1, P)
    rng = random.Random(381)
    cases = [(0, 0), (1, 1), (P - 1, P - 1), (P - 1, 1), ((1 << 380) - 1, P - 2)]
    cases += [(rng.randrange(P), rng.randrange(P)) for _ in range(60)]
    for a, b in cases:
        m = Machine(_limbs(a), _limbs(b))
        assert m.run(stmts) == a * b * rinv % P
    assert (m.mads, m.addcs) == (288, 288)      # 144 a_i b_j + 144 m_i p_j, one carry add each


def test_the_committed_file_is_what_the_generator_prints():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_fp_mul.py")], capture_output=True, text=True,
                         check=True).stdout
    assert out == open(INC).read()
