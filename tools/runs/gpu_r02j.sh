#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02j_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02j_tests.log
tail -n 16 gpurun_out/r02j_tests.log
timeout 600 python bench.py > gpurun_out/r02j_bench_full.json 2> gpurun_out/r02j_bench_full.err; echo "bench rc $?"
tail -n 5 gpurun_out/r02j_bench_full.err
python - <<PY
import json
d=json.load(open("gpurun_out/r02j_bench_full.json"))
for k in ("value","ms_per_step","step_ms_p50","step_ms_min","get_head_p50_us","checked_against_oracle","oracle_check"):
    print(k, d.get(k))
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","traffic","avg_launch_ms")})
print("cpu", json.dumps(d["cpu_baseline"], indent=0)[:1500])
PY
