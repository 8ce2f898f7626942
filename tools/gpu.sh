#!/bin/bash
# One parameterised driver for everything this repo runs on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/gpu.sh <tag> <action> [<action> ...]'
# Outputs go to gpurun_out/<tag>/ (scratch, merged back by gpurun); what is kept is copied to profiles/ by hand.
# Environment: BENCH_ARGS (extra flags for every bench.py run), PYTEST_ARGS, STEPS (default 200), K (pytest -k filter).
# Actions (run in the order given, all inside ONE call = one box, so variants compare like for like):
#   tests            pytest -m gpu over tests/ (-x -q); K=<expr> narrows it
#   bench            bench.py default line (with cpu_baseline)                    -> bench_full.json
#   quick            bench.py --steps $STEPS --no-cpu-baseline                    -> bench_quick.json
#   driver           bench.py --steps 20 --warmup 5 (what the round driver runs)  -> bench_driver.json
#   trace            quick under POSEVO_HOST_TRACE=1                              -> host_trace.txt
#   timeline         rocprofv3 --kernel-trace of 30 steps                         -> timeline.txt, kernel_stats.txt
#   pmc              FETCH_SIZE / WRITE_SIZE passes (kernel-trace only)           -> pmc_fetch.json, pmc_write.json
#   env:<NAME>=<V>   export a variable for the actions that follow (e.g. env:POSEVO_PIPELINE_LAG=3)
#   label:<name>     suffix for the output files of the actions that follow
set -u
TAG=${1:?tag}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/$TAG
mkdir -p "$O"
STEPS=${STEPS:-200}
BENCH_ARGS=${BENCH_ARGS:-}
SFX=""
line() { timeout 20 python tools/benchline.py < "$1" 2>/dev/null || head -c 300 "$1"; }
for act in "$@"; do
  case "$act" in
    env:*) export "${act#env:}"; echo "[gpu.sh] export ${act#env:}";;
    label:*) SFX="_${act#label:}";;
    tests)
      timeout 1500 python -m pytest tests -x -q -m gpu ${K:+-k "$K"} ${PYTEST_ARGS:-} > "$O/pytest$SFX.log" 2>&1
      echo "[gpu.sh] pytest rc $?"; tail -12 "$O/pytest$SFX.log";;
    bench)
      timeout 900 python bench.py $BENCH_ARGS > "$O/bench_full$SFX.json" 2> "$O/bench_full$SFX.err"
      echo "[gpu.sh] bench rc $?"; line "$O/bench_full$SFX.json";;
    quick)
      timeout 600 python bench.py --steps "$STEPS" --warmup 6 --no-cpu-baseline --no-slot-cadence --no-signed-steps $BENCH_ARGS > "$O/bench_quick$SFX.json" 2> "$O/bench_quick$SFX.err"
      echo "[gpu.sh] quick$SFX rc $?"; line "$O/bench_quick$SFX.json";;
    driver)
      timeout 600 python bench.py --steps 20 --warmup 5 $BENCH_ARGS > "$O/bench_driver$SFX.json" 2> "$O/bench_driver$SFX.err"
      echo "[gpu.sh] driver$SFX rc $?"; line "$O/bench_driver$SFX.json";;
    trace)
      POSEVO_HOST_TRACE=1 timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-verify-steps $BENCH_ARGS \
        > "$O/bench_trace$SFX.json" 2> "$O/bench_trace$SFX.err"
      grep "posevo host" "$O/bench_trace$SFX.err" | grep -v "comm\." | cut -c1-125 > "$O/host_trace$SFX.txt"
      cat "$O/host_trace$SFX.txt"; line "$O/bench_trace$SFX.json";;
    timeline)
      rm -rf "$O/prof"; mkdir -p "$O/prof"
      timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof" -o tl -- python bench.py --steps 30 --warmup 6 --no-cpu-baseline \
        --no-verify-steps --no-slot-cadence --no-signed-steps $BENCH_ARGS > "$O/bench_under_rocprof$SFX.json" 2> "$O/prof_err$SFX.log"
      timeout 120 python tools/rocpd_timeline.py "$O/prof/tl_results.db" 20 2 > "$O/timeline$SFX.txt" 2>&1
      timeout 120 python tools/rocpd_stats.py "$O/prof/tl_results.db" "$O/kernel_stats$SFX.txt" > /dev/null 2>&1
      cat "$O/timeline$SFX.txt"; cut -c1-130 "$O/kernel_stats$SFX.txt" | head -24; line "$O/bench_under_rocprof$SFX.json"
      rm -rf "$O/prof";;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf "$O/pmc"; mkdir -p "$O/pmc"
        timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$O/pmc" -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline \
          --no-verify-steps --no-slot-cadence --no-signed-steps $BENCH_ARGS > "$O/pmc_$c$SFX.log" 2>&1
        timeout 120 python tools/rocpd_pmc.py "$O/pmc/p_results.db" $c > "$O/pmc_$c$SFX.json" 2>> "$O/pmc_$c$SFX.log"
        rm -rf "$O/pmc"
      done
      head -40 "$O/pmc_FETCH_SIZE$SFX.json"; head -40 "$O/pmc_WRITE_SIZE$SFX.json";;
    *) echo "[gpu.sh] unknown action $act";;
  esac
done
