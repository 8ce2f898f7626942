#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2 3 4; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02p_bench_drv$i.json 2> gpurun_out/r02p_bench_drv$i.err
python - <<PY
import json
d=json.load(open("gpurun_out/r02p_bench_drv$i.json"))
print("driver-style run $i: ms/step %.3f p50 %.3f min %.3f p90 %.3f"%(d["ms_per_step"], d["step_ms_p50"], d["step_ms_min"], d["step_ms_p90"]), "head p50 %.1f"%d["get_head_p50_us"])
PY
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02p_bench_drv_full.json 2> gpurun_out/r02p_bench_drv_full.err
python - <<PY
import json
d=json.load(open("gpurun_out/r02p_bench_drv_full.json"))
print("full: ms/step %.3f value %.3e checked %s"%(d["ms_per_step"], d["value"], d["checked_against_oracle"]))
print(json.dumps(d["cpu_baseline"])[:900])
PY
python -m pytest tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -n 2
