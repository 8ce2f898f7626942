#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03j_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03j_tests.log
tail -n 3 gpurun_out/r03j_tests.log
timeout 1500 bash tools/profile_round.sh r02 2>&1 | tail -12
timeout 300 python bench.py --host-arena > gpurun_out/r02_bench_host_arena.json 2> gpurun_out/r02_bench_host_arena.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02_bench_k20.json 2>/dev/null
timeout 300 python bench.py --no-lag --no-cpu-baseline > gpurun_out/r02_bench_nolag.json 2>/dev/null
timeout 300 python bench.py --no-pipeline --no-cpu-baseline > gpurun_out/r02_bench_sync.json 2>/dev/null
python - <<'PY'
import json
for n in ["bench_full","bench_host_arena","bench_k20","bench_nolag","bench_sync"]:
    d=json.loads(open(f"gpurun_out/r02_{n}.json").read().strip().splitlines()[-1])
    print(n, "ms/step %.3f p50 %.3f value %.3f G frac %.4f head %.1f"%(d["ms_per_step"], d["step_ms_p50"], d["value"]/1e9, d["roofline"]["frac"], d["get_head_p50_us"]))
PY
