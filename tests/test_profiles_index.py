"""CPU: profiles/README.md is the index the docs point into -- every round-5 / round-6 file it names exists, every such file
that exists is named, the rounds' call scripts parse, and the bench line kept as the round's record carries the contract's keys."""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _expand(name):
    """`r05_bench_sig_behind{1,0}.json` -> both names; a `…` or `*` in a name is a family, not a file."""
    m = re.search(r"\{([^}]*)\}", name)
    if not m:
        return [name]
    out = []
    for alt in m.group(1).split(","):
        out += _expand(name[:m.start()] + alt.strip() + name[m.end():])
    return out


import pytest


@pytest.mark.parametrize("rnd,least", [("r05", 30), ("r06", 25)])
def test_round_index_and_files_agree(rnd, least):
    text = open(os.path.join(PROF, "README.md")).read()
    start = text.index("## Round " + rnd[-1])
    nxt = text.find("\n## Round ", start + 1)
    section = text[start:nxt if nxt > 0 else len(text)]
    named = set()
    for tok in re.findall(r"`([^`]+)`", section):
        tok = tok.strip()
        if "/" in tok and not tok.startswith("profiles/"):
            continue                                  # a tool or test path
        tok = tok[len("profiles/"):] if tok.startswith("profiles/") else tok
        if re.fullmatch(r"(%s_[\w{},.+-]+|hbm_traffic\.json|NOTES_%s\.md)" % (rnd, rnd), tok) and tok.endswith((".json", ".txt", ".md")):
            named.update(_expand(tok))
    assert len(named) > least, sorted(named)
    missing = sorted(n for n in named if not os.path.exists(os.path.join(PROF, n)))
    assert not missing, f"profiles/README.md names files that are not there: {missing}"
    present = {f for f in os.listdir(PROF) if f.startswith(rnd + "_")}
    unnamed = sorted(present - named)
    assert not unnamed, f"{rnd} files without a line in profiles/README.md: {unnamed}"


@pytest.mark.parametrize("script,letters", [("r05_calls.sh", "efghijkl"), ("r06_calls.sh", "bcdefghijklmpqr")])
def test_round_call_script_parses(script, letters):
    path = os.path.join(ROOT, "tools", script)
    assert subprocess.run(["bash", "-n", path]).returncode == 0
    src = open(path).read()
    for letter in letters:
        assert f"call_{letter}()" in src


@pytest.mark.parametrize("record", ["r05_bench_full.json", "r06_bench_full.json"])
def test_the_rounds_bench_record_has_the_contract_keys(record):
    d = json.load(open(os.path.join(PROF, record)))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] > 0
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1
    assert d["steps_verified"] == d["steps"] and d["checked_against_oracle"] is True
