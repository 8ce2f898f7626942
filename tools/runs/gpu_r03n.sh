#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03n_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03n_tests.log
grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" gpurun_out/r03n_tests.log | tail -n 4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 bash tools/profile_round.sh r02 > /dev/null 2>&1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_bench_full.json").read().strip().splitlines()[-1])
print("bench_full ms/step %.3f p50 %.3f value %.3f G frac %.4f head %.1f acc %.4f"%(d["ms_per_step"], d["step_ms_p50"], d["value"]/1e9, d["roofline"]["frac"], d["get_head_p50_us"], d["kernel_avg_ms"]["g1_accumulate"]))
PY
