#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["config"]["workload"][:60], "| ms/step %.3f p50 %.3f value %.3f G frac %.4f head p50 %.1f"%(d["ms_per_step"], d["step_ms_p50"], d["value"]/1e9, d["roofline"]["frac"], d["get_head_p50_us"]), {k:(round(v,4) if v else v) for k,v in d["kernel_avg_ms"].items() if v}, d.get("oracle_check"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
timeout 600 python bench.py --validators 4194304 --blocks 8192 --mixed-balances --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/r02_bench_config5_4m_one_gpu.json 2> gpurun_out/r03p.err
show gpurun_out/r02_bench_config5_4m_one_gpu.json
timeout 300 python bench.py --validators 262144 --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r02_bench_config3_256k.json 2> gpurun_out/r03p.err
show gpurun_out/r02_bench_config3_256k.json
timeout 300 python bench.py --validators 131072 --steps 200 --warmup 6 --no-cpu-baseline > gpurun_out/r02_bench_128k.json 2> gpurun_out/r03p.err
show gpurun_out/r02_bench_128k.json
