"""CPU: what bench.py promises the round driver without running anything on a GPU -- its named shapes are BASELINE.json's
configs as written, `--gpus N` alone means configs[3] under strong scaling, and the JSON keys the contract names exist in
the line's template."""
import ast
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_source():
    """bench.py (arguments, the timed region, the JSON line) + bench_legs/ (workload, CPU baselines, checks, the extra legs)."""
    import glob

    paths = [os.path.join(ROOT, "bench.py")] + sorted(glob.glob(os.path.join(ROOT, "bench_legs", "*.py")))
    return "\n".join(open(p).read() for p in paths)


def _shapes():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", None) == "SHAPES":
            return eval(compile(ast.Expression(node.value), "bench.SHAPES", "eval"))   # a literal dict of dict(...) calls
    raise AssertionError("bench.SHAPES not found")


def _number(text, pattern):
    m = re.search(pattern, text)
    assert m, (pattern, text)
    return int(m.group(1).replace(" ", "").replace(" ", ""))


def test_named_shapes_are_the_baseline_configs_as_written():
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    shapes = _shapes()
    for k in (1, 2, 3, 4):
        sh = shapes[f"configs{k}"]
        assert sh["validators"] == _number(cfgs[k], r"^([\d ]+) validators"), cfgs[k]
    assert shapes["configs1"]["committees"] == 64 * 32                      # "64 committees/slot"
    assert shapes["configs2"]["blocks"] == _number(cfgs[2], r"deep ([\d ]+)-block")
    assert shapes["configs4"]["blocks"] == _number(cfgs[4], r", ([\d ]+)-block tree")
    assert shapes["configs4"]["mixed_balances"] is True and "mixed balances" in cfgs[4]
    assert shapes["configs3"]["validators"] // shapes["configs3"]["committees"] == 512
    # SURVEY.md 8(d)'s table: c2's 2048-block chain with side branches, c5's 1 % equivocating validators
    assert shapes["configs1"]["blocks"] == 2048 and shapes["configs1"]["tree_kind"] == "branchy"
    assert shapes["configs4"]["equivocating_frac"] == 0.01
    assert all(shapes[f"configs{k}"]["equivocating_frac"] == 0.0 for k in (1, 2, 3))


def test_gpus_n_alone_is_a_named_config_under_strong_scaling():
    src = _bench_source()
    assert re.search(r'"--shape".*default="configs3"', src)
    assert re.search(r'"--scaling", choices=\["strong", "weak"\], default="strong"', src)
    # the workload string of a named shape starts the way the driver's reader expects
    assert 'f"BASELINE configs[{named}]"' in src


def test_the_contract_keys_are_in_the_line():
    src = _bench_source()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "slot_cadence", "with_signatures",
                "ms_per_step_with_signatures"):
        assert f'"{key}"' in src, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert f'"{key}"' in src, key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert f'"{key}"' in src or f"{key}=" in src, key


def test_stdout_carries_nothing_but_the_line():
    """The driver reads ONE JSON line from stdout.  bench.py claims descriptor 1 before any library is loaded (gloo and RCCL
    print banners to it, the latter through C stdio at exit -- behind the line) and writes its line to the saved descriptor.
    Without a GPU the run ends before the line exists: stdout stays empty, the reason is on stderr, the exit code says so."""
    import subprocess
    import sys

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=240, cwd=ROOT)
    import torch
    if torch.cuda.is_available():   # on a GPU box: exactly the line
        lines = [l for l in p.stdout.splitlines() if l.strip()]
        assert len(lines) == 1 and json.loads(lines[0])["steps"] == 2, p.stdout[:400]
        return
    assert p.returncode != 0 and p.stdout == "", (p.returncode, p.stdout[:200])
    assert "needs an MI355X" in p.stderr
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("os.dup2(2, 1)") < src.index("import torch"), "descriptor 1 is claimed before the first library loads"
    assert "os.write(json_fd" in src and "print(json.dumps(out))" not in src
