#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_sharded.py -m gpu -q -x > gpurun_out/r02y_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02y_tests.log
tail -n 4 gpurun_out/r02y_tests.log
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step %.3f p50 %.3f min %.3f head_p50 %s"%(d["ms_per_step"], d["step_ms_p50"], d["step_ms_min"], d.get("get_head_p50_us")), {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v})
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
for i in 1 2; do
timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r02y_dev_$i.json 2> gpurun_out/r02y_dev_$i.err
show gpurun_out/r02y_dev_$i.json
timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline --host-arena > gpurun_out/r02y_host_$i.json 2> gpurun_out/r02y_host_$i.err
show gpurun_out/r02y_host_$i.json
done
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 100 --warmup 6 --no-cpu-baseline > gpurun_out/r02y_trace.json 2> gpurun_out/r02y_hosttrace.txt
grep "posevo host" gpurun_out/r02y_hosttrace.txt | grep -v "comm\."
