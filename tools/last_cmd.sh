mkdir -p gpurun_out/r03u
A="--validators 4194304 --blocks 8192 --mixed-balances --steps 30 --warmup 6 --no-cpu-baseline --no-shuffle-variant"
timeout 900 python bench.py $A > gpurun_out/r03u/c5_one_gpu.json 2> gpurun_out/r03u/c5_one_gpu.err; echo rc $?
for n in 8 4; do
timeout 900 python bench.py $A --emulate-ranks $n > gpurun_out/r03u/c5_emul_$n.json 2> gpurun_out/r03u/c5_emul_$n.err; echo rc $?
done
python - <<PY
import json
for f in ("c5_one_gpu","c5_emul_8","c5_emul_4"):
    try:
        d=json.loads(open(f"gpurun_out/r03u/{f}.json").read().strip().splitlines()[-1])
        print(f, {k:(round(d[k],4) if isinstance(d.get(k),float) else d.get(k)) for k in ("ms_per_step","value","emulated_job_attestations_per_s","checked_against_oracle","steps_verified","get_head_p50_us")}, {k:round(v*1e3,1) for k,v in d["kernel_avg_ms"].items() if v})
    except Exception as e: print(f, "fail", e)
PY
tail -2 gpurun_out/r03u/*.err | cut -c1-200
