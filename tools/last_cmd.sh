export STEPS=100
for q in -1 1; do
POSEVO_PREP_PRIO=$q BENCH_ARGS="--no-verify-steps" bash tools/gpu.sh r03n label:prio$q quick > /dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r03n/bench_quick_prio$q.json").read().strip().splitlines()[-1])
print("prep prio $q", {k:round(d.get(k),4) for k in ("ms_per_step","ms_per_step_with_shuffle")})
PY
done
POSEVO_PREP_PRIO=-1 STEPS=60 BENCH_ARGS="--with-shuffle --no-verify-steps" bash tools/gpu.sh r03n label:wsp timeline | head -45
