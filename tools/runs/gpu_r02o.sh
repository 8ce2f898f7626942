#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in 0 1 0 1; do
POSEVO_G1_SERIAL_FINISH=$v timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r02o_bench_sf$v.json 2> gpurun_out/r02o_bench_sf$v.err
python - <<PY
import json
d=json.load(open("gpurun_out/r02o_bench_sf$v.json"))
print("serial_finish=$v ms/step %.3f p50 %.3f min %.3f"%(d["ms_per_step"], d["step_ms_p50"], d["step_ms_min"]), {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v})
PY
done
