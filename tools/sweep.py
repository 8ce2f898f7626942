#!/usr/bin/env python
"""A/B sweep of the streaming step's scheduling knobs inside ONE gpurun call (one box: variants compare like for like).

    python tools/sweep.py OUTDIR [--budget SECONDS]

Phase 1: every knob alone against the default build (bench.py --steps 200, every timed step verified against its
synchronous replay -- a knob that breaks an ordering fails `steps_verified`).  Phase 2: the winners together, then greedy
additions.  Phase 3: the driver's own command shape (--steps 20 --warmup 5) three times for the default and for the best
set.  Writes OUTDIR/sweep.json and prints a table; profiles/ keeps the table of the round."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--no-cpu-baseline", "--no-slot-cadence", "--no-signed-steps", "--no-shuffle-variant"]

# name -> (environment, extra arguments, exclusive group)
KNOBS = {
    "lag6": ({}, ["--lag", "6"], "lag"),
    "lag7": ({}, ["--lag", "7"], "lag"),
    "hwq6": ({"GPU_MAX_HW_QUEUES": "6"}, [], "hwq"),
    "hwq8": ({"GPU_MAX_HW_QUEUES": "8"}, [], "hwq"),
    "state_on_aux": ({"POSEVO_STATE_ON": "0"}, [], "state"),
    "not_exclusive": ({"POSEVO_ACC_EXCLUSIVE": "0"}, [], "excl"),
    "unpaired": ({"POSEVO_PAIR": "0"}, [], None),
}


def run(out_dir, tag, names, steps=200, warmup=6, log=None):
    env = dict(os.environ)
    args = []
    for n in names:
        e, a, _ = KNOBS[n]
        env.update(e)
        args += a
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup)] + QUICK + args
    t0 = time.time()
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        d = json.loads(line[-1]) if line else None
        err = None if d else (p.stderr[-600:] or "no JSON line")
    except subprocess.TimeoutExpired:
        d, err = None, "timeout"
    rec = {"tag": tag, "knobs": list(names), "steps": steps, "wall_s": round(time.time() - t0, 1), "error": err}
    if d:
        k = d.get("kernel_avg_ms", {})
        rec.update(ms_per_step=d["ms_per_step"], p50=d.get("step_ms_p50"), p90=d.get("step_ms_p90"),
                   first20=d.get("ms_per_step_first_20"), verified=d.get("steps_verified"),
                   acc_ms=k.get("g1_accumulate"), fin_ms=k.get("g1_normalise"), tree_ms=k.get("g1_tree"),
                   head_p50_us=d.get("get_head_p50_us"), oracle=d.get("checked_against_oracle"))
        with open(os.path.join(out_dir, f"bench_{tag}.json"), "w") as f:
            f.write(json.dumps(d) + "\n")
    ok = d is not None and rec.get("verified") == steps  # (the oracle check of step 0 belongs to the CPU legs, which QUICK drops)
    rec["ok"] = ok
    msg = (f"[sweep] {tag:28s} " + (f"{rec['ms_per_step']*1e3:7.1f} us/step  p50 {rec['p50']*1e3:6.1f}  p90 {rec['p90']*1e3:6.1f}  "
                                    f"acc {rec['acc_ms']*1e3:6.1f}  fin {rec['fin_ms']*1e3:6.1f}  verified {rec['verified']}/{steps}"
                                    if d else f"FAILED: {err}") + f"  ({rec['wall_s']} s)")
    print(msg, flush=True)
    if log is not None:
        log.append(rec)
    return rec


def driver_shape(out_dir, best_set, left, log):
    drv = {"default": [], "best": []}
    for i in range(3):
        if left() < 25:
            break
        drv["default"].append(run(out_dir, f"driver_default_{i}", [], steps=20, warmup=5, log=log))
        if best_set:
            drv["best"].append(run(out_dir, f"driver_best_{i}", best_set, steps=20, warmup=5, log=log))
    return drv


def main():
    out_dir = sys.argv[1]
    budget = float(sys.argv[sys.argv.index("--budget") + 1]) if "--budget" in sys.argv else 480.0
    sets = sys.argv[sys.argv.index("--sets") + 1].split(",") if "--sets" in sys.argv else None
    os.makedirs(out_dir, exist_ok=True)
    t_start = time.time()
    left = lambda: budget - (time.time() - t_start)
    log = []
    base = [run(out_dir, "base_a", [], log=log)]
    singles = {}
    if sets:  # named sets (a+b,c+d,...) instead of the knobs one by one
        for spec in sets:
            singles[spec] = run(out_dir, "set_" + spec, spec.split("+"), log=log)
    else:
        for name in KNOBS:
            if left() < 150:
                print(f"[sweep] budget: skipping {name}", flush=True)
                continue
            singles[name] = run(out_dir, name, [name], log=log)
    base.append(run(out_dir, "base_b", [], log=log))
    base_ok = [b["ms_per_step"] for b in base if b["ok"]]
    if not base_ok:
        raise SystemExit("[sweep] the default build failed both of its runs")
    ref = sum(base_ok) / len(base_ok)
    print(f"[sweep] reference {ref*1e3:.1f} us/step", flush=True)
    gains = {n: ref / r["ms_per_step"] - 1.0 for n, r in singles.items() if r["ok"]}
    if sets:
        top = max(gains, key=gains.get) if gains else None
        best_set, best = (top.split("+"), singles[top]["ms_per_step"]) if top and gains[top] > 0.01 else ([], ref)
        chosen = [top] if top else []
    else:
        # winners: > 1 % better than the reference, the best of each exclusive group
        chosen, seen_groups = [], set()
        for n, g in sorted(gains.items(), key=lambda kv: -kv[1]):
            grp = KNOBS[n][2]
            if g < 0.01 or (grp and grp in seen_groups):
                continue
            chosen.append(n)
            if grp:
                seen_groups.add(grp)
        print(f"[sweep] winners alone: {[(n, round(gains[n]*100, 1)) for n in chosen]}", flush=True)
        best_set, best = [], ref
        if chosen and left() > 60:
            r = run(out_dir, "combo_all", chosen, log=log)
            if r["ok"] and r["ms_per_step"] < best:
                best_set, best = list(chosen), r["ms_per_step"]
        # greedy from the best single if the whole set did not beat it
        top = chosen[0] if chosen else None
        if top and (not best_set or singles[top]["ms_per_step"] < best * 0.99):
            best_set, best = [top], singles[top]["ms_per_step"]
            for n in chosen[1:]:
                if left() < 90:
                    break
                r = run(out_dir, "greedy_" + "+".join(best_set + [n]), best_set + [n], log=log)
                if r["ok"] and r["ms_per_step"] < best * 0.995:
                    best_set, best = best_set + [n], r["ms_per_step"]
    print(f"[sweep] best set {best_set}: {best*1e3:.1f} us/step ({(ref/best-1)*100:.1f} % over the default)", flush=True)
    drv = driver_shape(out_dir, best_set, left, log)  # the driver's command shape
    summary = {"reference_ms": ref, "gains_alone": gains, "chosen": chosen, "best_set": best_set, "best_ms": best,
               "driver_default_ms": [r.get("ms_per_step") for r in drv["default"]],
               "driver_best_ms": [r.get("ms_per_step") for r in drv["best"]], "runs": log}
    with open(os.path.join(out_dir, "sweep.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("[sweep] driver-shape default", [round(x * 1e3, 1) for x in summary["driver_default_ms"] if x],
          "best", [round(x * 1e3, 1) for x in summary["driver_best_ms"] if x], flush=True)


if __name__ == "__main__":
    main()
