#!/bin/bash
# The GPU calls of round 5, one function per `gpurun` call, in the order they were made (each call = one box: what is compared
# is compared inside a call).  usage on the GPU box:   gpurun --timeout 900 -- 'bash tools/r05_calls.sh <letter>'
# Outputs go to gpurun_out/r05<letter>/ (scratch); what is kept was copied to profiles/ (index: profiles/README.md, Round 5).
cd "${GRAFT_REPO_ROOT:-/root/repo}"

# round 5, call b: the paired launches -- their tests first, then the whole -m gpu suite, then A/B of the step period
call_b() {
  O=gpurun_out/r05b; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_pairing.py -x -q > $O/pytest_pairing.log 2>&1; echo "[r05b] pairing tests rc $?"; tail -15 $O/pytest_pairing.log
  export BENCH_ARGS=""
  bash tools/gpu.sh r05b label:pair quick driver env:POSEVO_PAIR=0 label:nopair quick driver
  unset POSEVO_PAIR
  timeout 300 python tools/engine_timeline.py --steps 24 --show 2 > $O/engine_timeline_pair.txt 2>&1; tail -45 $O/engine_timeline_pair.txt
  timeout 300 python tools/engine_timeline.py --cold 20 > $O/engine_timeline_cold20_pair.txt 2>&1; head -30 $O/engine_timeline_cold20_pair.txt
  bash tools/gpu.sh r05b label:all tests
}

# round 5, call c: pairing after the first-use fixes; the new G2 leg; the whole suite; the full bench line
call_c() {
  O=gpurun_out/r05c; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_pairing.py -q > $O/pytest_pairing.log 2>&1; echo "[r05c] pairing tests rc $?"; tail -8 $O/pytest_pairing.log
  timeout 900 python -m pytest tests/test_gpu_g2.py -q -k "aggregate_signatures" > $O/pytest_g2.log 2>&1; echo "[r05c] g2 tests rc $?"; tail -8 $O/pytest_g2.log
  bash tools/gpu.sh r05c label:pair driver quick
  timeout 300 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20.txt 2>&1; head -26 $O/engine_timeline_cold20.txt
  timeout 300 python tools/engine_timeline.py --steps 24 --show 2 > $O/engine_timeline.txt 2>&1; tail -12 $O/engine_timeline.txt
  bash tools/gpu.sh r05c label:full bench
  bash tools/gpu.sh r05c label:all tests
}

# round 5, call d: sharded steps with held launches, the signature leg behind the accumulation, slot cadence diagnosis, per-rank load of range shards
call_d() {
  O=gpurun_out/r05d; mkdir -p $O
  timeout 1200 python -m pytest tests/test_gpu_dist_custom.py tests/test_gpu_sharded.py tests/test_gpu_pairing.py tests/test_gpu_g2.py -x -q > $O/pytest_sharded.log 2>&1; echo "[r05d] sharded/pairing/g2 tests rc $?"; tail -12 $O/pytest_sharded.log
  POSEVO_SLOT_TIMELINE=$O/slot_timeline.txt POSEVO_HOST_TRACE=1 timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > $O/bench_legs.json 2> $O/bench_legs.err; echo "[r05d] bench legs rc $?"
  python - <<'PY'
import json
d=json.load(open("gpurun_out/r05d/bench_legs.json"))
print("ms/step", d["ms_per_step"], "signed", d.get("ms_per_step_with_signatures"), "shuffle", d.get("ms_per_step_with_shuffle"))
print("slot", {k:v for k,v in d.get("slot_cadence",{}).items() if "us" in k})
print("unagg", {k:v for k,v in d.get("with_unaggregated_signatures",{}).items() if k in ("ms_per_epoch","signatures_per_s","error")})
PY
  grep "posevo host" $O/bench_legs.err | cut -c1-120 | tail -40
  sed -n 1,4p $O/slot_timeline.txt; awk 'NR>400 && NR<470' $O/slot_timeline.txt
  for pair in 1 0; do for shape in "configs3 131072" "configs4 524288"; do set -- $shape
    POSEVO_PAIR=$pair POSEVO_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --shape $1 --validators $2 --steps 100 --warmup 6 --no-cpu-baseline --no-signed-steps --no-slot-cadence --no-shuffle-variant > $O/rank_${1}_pair$pair.json 2> $O/rank_${1}_pair$pair.err
    echo "[r05d] per-rank step of an 8-way range shard, $1, pair=$pair: rc $? $(timeout 20 python tools/benchline.py < $O/rank_${1}_pair$pair.json 2>/dev/null | cut -c1-150)"; done; done
  bash tools/gpu.sh r05d label:all tests
}

# round 5, call e: the whole GPU suite at HEAD, then the round's profile set (bench line, rocprof kernel stats, PMC passes,
# timeline) and the driver's command
call_e() {
  O=gpurun_out/r05e; mkdir -p $O
  bash tools/gpu.sh r05e label:all tests
  bash tools/profile_round.sh r05
  bash tools/gpu.sh r05e label:pair driver
  timeout 300 python tools/engine_timeline.py --steps 24 --show 2 > $O/engine_timeline_paired.txt 2>&1; tail -14 $O/engine_timeline_paired.txt
  timeout 300 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20_paired.txt 2>&1; head -30 $O/engine_timeline_cold20_paired.txt
}

# round 5, call f: the scheduling knobs of the streaming G1 chain A/B on one box (tools/sweep.py), the streaming tests with all
# of them switched on, the per-rank load of the sharded divisions
call_f() {
  O=gpurun_out/r05f; mkdir -p $O
  timeout 560 python tools/sweep.py $O --budget 470 2>&1 | tee $O/sweep.log | grep "^\[sweep\]"
  ( export POSEVO_ACC_EXCLUSIVE=1 POSEVO_TREE_ROTATE=1 POSEVO_ACC_DONE_EVENT=1 POSEVO_ROWS_EVENT=1 POSEVO_STATE_ON=1 POSEVO_SIDE_STREAMS=2
    timeout 400 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_pipeline.py tests/test_gpu_pipeline_robust.py tests/test_gpu_g1_accumulate.py tests/test_gpu_shapes.py -x -q > $O/pytest_knobs.log 2>&1
    echo "[r05f] streaming tests with every knob on: rc $?"; tail -5 $O/pytest_knobs.log )
  for shape in configs3 configs4; do
    timeout 300 python bench.py --emulate-ranks 8 --shape $shape --steps 100 --warmup 6 --no-signed-steps --no-slot-cadence > $O/emulate8_$shape.json 2> $O/emulate8_$shape.err
    echo "[r05f] committee shards, per-rank step of 8, $shape: rc $? $(timeout 20 python tools/benchline.py < $O/emulate8_$shape.json 2>/dev/null | cut -c1-160)"
  done
}

# round 5, call g: the winners of call f together (named sets), the driver's command shape for the default and the best
# set, tools/mfmabench
call_g() {
  O=gpurun_out/r05g; mkdir -p $O
  timeout 330 python tools/sweep.py $O --budget 300 --sets "exclusive+state_on_fin,exclusive+state_on_fin+tree_rotate,exclusive+state_on_fin+rows_event+done_event,exclusive+state_on_fin+tree_rotate+rows_event+done_event,exclusive+state_on_fin+lag6,two_side+state_on_fin,two_side+state_on_fin+rows_event+done_event" 2>&1 | tee $O/sweep.log | grep "^\[sweep\]"
  timeout 120 tools/mfmabench > $O/mfmabench.txt 2>&1; echo "[r05g] mfmabench rc $?"; cat $O/mfmabench.txt
}

# round 5, call h: the streaming step's timeline with the new defaults (exclusive accumulation workgroups, flag passes on the
# tree's stream); the signature leg behind / beside the accumulation on one box
call_h() {
  O=gpurun_out/r05h; mkdir -p $O
  timeout 200 python tools/engine_timeline.py --steps 30 --show 4 > $O/engine_timeline.txt 2>&1; tail -16 $O/engine_timeline.txt
  timeout 200 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20.txt 2>&1; head -26 $O/engine_timeline_cold20.txt
  for b in 1 0; do
    POSEVO_SIG_BEHIND=$b timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant > $O/bench_sig_behind$b.json 2> $O/bench_sig_behind$b.err
    python - <<PY
import json
d=json.load(open("$O/bench_sig_behind$b.json"))
print("[r05h] sig_behind=$b: ms/step", round(d["ms_per_step"],4), "signed", round(d["ms_per_step_with_signatures"],4), d["with_signatures"].get("steps_verified"), "unaggregated ms/epoch", round(d["with_unaggregated_signatures"]["ms_per_epoch"],2))
PY
  done
}

# round 5, call i: the whole GPU suite with the new defaults, the round's profile set (bench line, rocprof kernel stats, PMC
# passes), the driver's command, the engine's own timelines, the accumulation alone on the same box (tools/accbench)
call_i() {
  O=gpurun_out/r05i; mkdir -p $O
  bash tools/gpu.sh r05i label:all tests
  bash tools/profile_round.sh r05
  bash tools/gpu.sh r05i label:final driver
  timeout 200 python tools/engine_timeline.py --steps 30 --show 3 > $O/engine_timeline.txt 2>&1; tail -12 $O/engine_timeline.txt
  timeout 200 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20.txt 2>&1; head -4 $O/engine_timeline_cold20.txt
  [ -x tools/accbench ] && (timeout 120 tools/accbench > $O/accbench.txt 2>&1; tail -6 $O/accbench.txt)
}

# round 5, call j: the accumulation alone and in the step on ONE box (tools/accbench, then the bench line and the driver's
# command); the N > 1 bench path with one rank over the engine's own RCCL (POSEVO_FORCE_DIST) at the per-rank sizes of 8 shards
call_j() {
  O=gpurun_out/r05j; mkdir -p $O
  timeout 120 tools/accbench > $O/accbench.txt 2>&1; echo "[r05j] accbench rc $?"; tail -8 $O/accbench.txt
  bash tools/gpu.sh r05j label:box5 quick driver
  for spec in "configs3 131072 engine" "configs4 524288 engine" "configs3 1048576 committee"; do set -- $spec
    POSEVO_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --shape $1 --validators $2 --sharded-mode $3 --steps 100 --warmup 6 --no-cpu-baseline --no-signed-steps --no-slot-cadence --no-shuffle-variant > $O/rank_${1}_$3.json 2> $O/rank_${1}_$3.err
    echo "[r05j] one rank over the engine's RCCL, $1 x $2 validators, $3 shards: rc $? $(timeout 20 python tools/benchline.py < $O/rank_${1}_$3.json 2>/dev/null | cut -c1-170)"; tail -2 $O/rank_${1}_$3.err | cut -c1-200
  done
}

# round 5, call k: the N > 1 bench path with ONE rank over the engine's own RCCL (POSEVO_FORCE_DIST) at the per-rank sizes of
# 8 shards -- what bench.py --gpus 8 runs on every rank, minus the other seven
call_k() {
  O=gpurun_out/r05k; mkdir -p $O
  export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
  for spec in "configs3 131072 engine" "configs4 524288 engine" "configs3 1048576 committee"; do set -- $spec
    POSEVO_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --shape $1 --validators $2 --sharded-mode $3 --steps 100 --warmup 6 --no-cpu-baseline --no-signed-steps --no-slot-cadence --no-shuffle-variant > $O/rank_${1}_$3.json 2> $O/rank_${1}_$3.err
    echo "[r05k] one rank over the engine's RCCL, $1 x $2 validators, $3 shards: rc $? $(timeout 20 python tools/benchline.py < $O/rank_${1}_$3.json 2>/dev/null | cut -c1-170)"; tail -2 $O/rank_${1}_$3.err | cut -c1-200
  done
}

# round 5, call l: bench.py's stdout carries exactly one line -- the driver's command, the launcher's form of it, and the N > 1
# path (one rank over the engine's RCCL, whose banner and gloo's used to land on stdout)
call_l() {
  O=gpurun_out/r05l; mkdir -p $O
  timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.out 2> $O/driver.err; echo "[r05l] driver command rc $? stdout lines $(wc -l < $O/driver.out)"; timeout 20 python tools/benchline.py < $O/driver.out | cut -c1-120
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-slot-cadence --no-signed-steps > $O/launcher.out 2> $O/launcher.err; echo "[r05l] launcher rc $? stdout lines $(wc -l < $O/launcher.out)"; timeout 20 python tools/benchline.py < $O/launcher.out | cut -c1-120
  POSEVO_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --validators 131072 --steps 20 --warmup 5 > $O/dist.out 2> $O/dist.err; echo "[r05l] N > 1 path, one rank: rc $? stdout lines $(wc -l < $O/dist.out)"; timeout 20 python tools/benchline.py < $O/dist.out | cut -c1-120; grep -c "Gloo\|RCCL version" $O/dist.err
}

"call_${1:?letter b..l}"
