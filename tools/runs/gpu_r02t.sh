#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_pipeline.py -m gpu -q -x > gpurun_out/r02t_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02t_tests.log
tail -n 5 gpurun_out/r02t_tests.log
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
POSEVO_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2955$i bench.py --gpus 1 --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r02t_engine_rccl_$i.json 2> gpurun_out/r02t_engine_rccl_$i.err
show gpurun_out/r02t_engine_rccl_$i.json
timeout 300 python bench.py --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r02t_single_$i.json 2> gpurun_out/r02t_single_$i.err
show gpurun_out/r02t_single_$i.json
done
POSEVO_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29557 bench.py --gpus 1 --steps 200 --warmup 6 --no-cpu-baseline --sharded-mode torch > gpurun_out/r02t_torch_path.json 2> gpurun_out/r02t_torch_path.err
show gpurun_out/r02t_torch_path.json
POSEVO_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29558 bench.py --gpus 1 --steps 200 --warmup 6 --no-cpu-baseline --validators 131072 > gpurun_out/r02t_engine_rccl_128k.json 2> gpurun_out/r02t_engine_rccl_128k.err
show gpurun_out/r02t_engine_rccl_128k.json
