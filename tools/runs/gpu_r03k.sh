#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03k_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03k_tests.log
tail -n 25 gpurun_out/r03k_tests.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r03k_bench.json 2> gpurun_out/r03k.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03k_bench.json").read().strip().splitlines()[-1])
print("ms/step %.3f p50 %.3f frac %.4f head %.1f"%(d["ms_per_step"], d["step_ms_p50"], d["roofline"]["frac"], d["get_head_p50_us"]))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
