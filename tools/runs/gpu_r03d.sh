#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step %.3f p50 %.3f min %.3f head_p50 %s frac %.4f"%(d["ms_per_step"], d["step_ms_p50"], d["step_ms_min"], d.get("get_head_p50_us"), d["roofline"]["frac"]), {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v})
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
for i in 1 2 3; do
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r03d_adaptive_$i.json 2> gpurun_out/r03d_adaptive_$i.err
show gpurun_out/r03d_adaptive_$i.json; grep "switched" gpurun_out/r03d_adaptive_$i.err
POSEVO_G1_STREAM_ONE_WAVE=0 timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r03d_k8_$i.json 2> gpurun_out/r03d_k8_$i.err
show gpurun_out/r03d_k8_$i.json
POSEVO_G1_STREAM_ONE_WAVE=1 timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r03d_k16_$i.json 2> gpurun_out/r03d_k16_$i.err
show gpurun_out/r03d_k16_$i.json
done
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_pipeline_robust.py tests/test_gpu_engine.py tests/test_gpu_forkchoice.py tests/test_gpu_shapes.py tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -3
