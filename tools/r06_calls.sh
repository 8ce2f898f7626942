#!/bin/bash
# The GPU calls of round 6, one function per `gpurun` call (each call = one box: what is compared is compared inside a call).
# usage on the GPU box:   gpurun --timeout 1500 -- 'bash tools/r06_calls.sh <letter>'
# Outputs go to gpurun_out/r06<letter>/ (scratch); what is kept is copied to profiles/ (index: profiles/README.md, Round 6).
cd "${GRAFT_REPO_ROOT:-/root/repo}"

# call b: the multi-workgroup k_att_plan -- its tests, then A/B against round 5's library (pos_evolution_amd/libposevo_base.so,
# built from the previous commit) on this box: the driver's command, 200 steps, the engine's own timeline
call_b() {
  O=gpurun_out/r06b; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_resident_rows.py tests/test_gpu_pairing.py -x -q > $O/pytest_plan.log 2>&1; echo "[r06b] plan tests rc $?"; tail -15 $O/pytest_plan.log
  bash tools/gpu.sh r06b label:new driver quick
  POSEVO_LIB_PATH=$PWD/pos_evolution_amd/libposevo_base.so bash tools/gpu.sh r06b label:base driver quick
  timeout 300 python tools/engine_timeline.py --steps 24 --show 2 > $O/engine_timeline_new.txt 2>&1; tail -42 $O/engine_timeline_new.txt
  bash tools/gpu.sh r06b label:all tests
}

# call c: (1) what inflates the accumulation in situ -- a diagnostic build (libposevo_dbg.so: POSEVO_DBG_SKIP=1 launches no
# k_g1_finish, =2 neither tree nor finish; results are garbage, nothing is verified) against the same build with both;
# (2) the unaggregated-signature leg under the profiler: kernel stats, two SQ counter passes, FETCH / WRITE
call_c() {
  O=gpurun_out/r06c; mkdir -p $O
  timeout 600 python -m pytest tests/test_gpu_resident_rows.py -x -q -k "many_workgroups" > $O/pytest_plan.log 2>&1; echo "[r06c] plan tests rc $?"; tail -4 $O/pytest_plan.log
  export POSEVO_LIB_PATH=$PWD/pos_evolution_amd/libposevo_dbg.so BENCH_ARGS="--no-verify-steps --no-oracle-check --no-shuffle-variant"
  for k in 0 1 2 0; do POSEVO_DBG_SKIP=$k bash tools/gpu.sh r06c label:skip$k quick; done
  unset POSEVO_LIB_PATH BENCH_ARGS
  cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
  timeout 300 python tools/sig_epoch.py --calls 3 > $O/sig_epoch_plain.txt 2>&1; cat $O/sig_epoch_plain.txt
  rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o sig -- python tools/sig_epoch.py --calls 3 > $O/sig_epoch_rocprof.txt 2>&1
  timeout 120 python tools/rocpd_stats.py $O/prof/sig_results.db $O/sig_kernel_stats.txt > /dev/null 2>&1; cut -c1-150 $O/sig_kernel_stats.txt | head -14; rm -rf $O/prof
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf $O/pmc; timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/pmc -o p -- python tools/sig_epoch.py --calls 2 > $O/sig_pmc_$i.log 2>&1
    for c in $set; do echo "== $c"; timeout 60 python tools/rocpd_pmc.py $O/pmc/p_results.db $c 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'g2' in k: print('  ',k[:40],v)
"; done > $O/sig_pmc_$i.txt 2>&1
    cat $O/sig_pmc_$i.txt; rm -rf $O/pmc
  done
}

# call d: collected signature legs -- their tests, the signed step at 1 / 4 / 8 steps per decompression on one box; the bench's
# unaggregated-signature leg under the profiler (why 45 ms there when the kernel takes 15 alone); the skip diagnostic of call c
# again, this time under the kernel trace (real durations, not the engine's events)
call_d() {
  O=gpurun_out/r06d; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_g2.py -x -q > $O/pytest_sig.log 2>&1; echo "[r06d] pairing + g2 tests rc $?"; tail -6 $O/pytest_sig.log
  for b in 4 1 8; do
    POSEVO_SIG_BATCH=$b timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant > $O/bench_sig$b.json 2> $O/bench_sig$b.err
    echo "[r06d] sig batch $b: rc $? $(python - <<PY
import json
d=json.loads(open("$O/bench_sig$b.json").read().strip().splitlines()[-1])
s=d.get("with_signatures",{}); u=d.get("with_unaggregated_signatures",{})
print("ms/step", round(d["ms_per_step"],4), "signed", d.get("ms_per_step_with_signatures"), "verified", s.get("steps_verified"), "| unagg ms/epoch", u.get("ms_per_epoch"), u.get("error"))
PY
)"
  done
  cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
  rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant --no-verify-steps --no-oracle-check > $O/bench_rocprof.json 2> $O/bench_rocprof.err
  timeout 120 python tools/rocpd_stats.py $O/prof/b_results.db $O/bench_kernel_stats.txt > /dev/null 2>&1; cut -c1-150 $O/bench_kernel_stats.txt | head -30; rm -rf $O/prof
  export POSEVO_LIB_PATH=$PWD/pos_evolution_amd/libposevo_dbg.so
  for k in 0 1 2; do
    rm -rf $O/prof; POSEVO_DBG_SKIP=$k timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o s -- python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant --no-signed-steps --no-verify-steps --no-oracle-check > $O/bench_skip$k.json 2> $O/bench_skip$k.err
    timeout 120 python tools/rocpd_stats.py $O/prof/s_results.db $O/skip${k}_kernel_stats.txt > /dev/null 2>&1; echo "== skip $k"; cut -c1-150 $O/skip${k}_kernel_stats.txt | head -12; rm -rf $O/prof
  done
  unset POSEVO_LIB_PATH
}

# call e: the legs' stream no longer joined into the tree's; fq_pow_pm3d4 with its table in registers (no scratch); the timed
# path on the 4096-deep chain and with the boost at configs[3]; the skip diagnostic with no-op kernels in place of tree / finish
# (every event and dependency as in the real run)
call_e() {
  O=gpurun_out/r06e; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_g2.py tests/test_gpu_keyvalidate.py -x -q > $O/pytest_sig.log 2>&1; echo "[r06e] pairing + g2 tests rc $?"; tail -6 $O/pytest_sig.log
  timeout 1200 python -m pytest tests/test_gpu_shapes.py -x -q -k "timed_path or signed_steps" > $O/pytest_shapes.log 2>&1; echo "[r06e] shapes tests rc $?"; tail -6 $O/pytest_shapes.log
  timeout 300 python tools/sig_epoch.py --calls 3 > $O/sig_epoch_plain.txt 2>&1; cat $O/sig_epoch_plain.txt
  for b in 4 1 8; do
    POSEVO_SIG_BATCH=$b timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant > $O/bench_sig$b.json 2> $O/bench_sig$b.err
    echo "[r06e] sig batch $b: rc $? $(python - <<PY
import json
d=json.loads(open("$O/bench_sig$b.json").read().strip().splitlines()[-1])
s=d.get("with_signatures",{}); u=d.get("with_unaggregated_signatures",{})
print("ms/step", round(d["ms_per_step"],4), "signed", d.get("ms_per_step_with_signatures"), "verified", s.get("steps_verified"), "| unagg ms/epoch", u.get("ms_per_epoch"), u.get("roofline_valu",{}).get("frac"), u.get("error"))
PY
)"
  done
  cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
  export POSEVO_LIB_PATH=$PWD/pos_evolution_amd/libposevo_dbg.so
  for k in 0 1 2 3; do
    rm -rf $O/prof; POSEVO_DBG_SKIP=$k timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o s -- python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant --no-signed-steps --no-verify-steps --no-oracle-check > $O/bench_skip$k.json 2> $O/bench_skip$k.err
    timeout 120 python tools/rocpd_stats.py $O/prof/s_results.db $O/skip${k}_kernel_stats.txt > /dev/null 2>&1; echo "== skip $k: $(timeout 20 python tools/benchline.py < $O/bench_skip$k.json | cut -c1-60)"; grep -E "k_g1_accumulate|k_g1_finish|k_g1_tree|k_dbg|k_pair" $O/skip${k}_kernel_stats.txt | cut -c1-150; rm -rf $O/prof
  done
  unset POSEVO_LIB_PATH
}

# call f: the queue probe's test; the signed step's own timeline (what holds it at 1 ms); the cold 20-step picture of the
# headline; the skip diagnostic once more, skipping only inside the streaming chain (the registry's points come from k_g1_finish too)
call_f() {
  O=gpurun_out/r06f; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_pipeline_robust.py tests/test_gpu_pairing.py -x -q > $O/pytest_a.log 2>&1; echo "[r06f] robust + pairing tests rc $?"; tail -6 $O/pytest_a.log
  timeout 600 python -m pytest tests/test_gpu_shapes.py -x -q -k "signed_steps" > $O/pytest_b.log 2>&1; echo "[r06f] signed bench test rc $?"; tail -4 $O/pytest_b.log
  timeout 300 python tools/engine_timeline.py --signed --lag 7 --steps 30 --show 8 > $O/engine_timeline_signed.txt 2>&1; tail -130 $O/engine_timeline_signed.txt | cut -c1-60
  timeout 300 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20.txt 2>&1; head -30 $O/engine_timeline_cold20.txt
  bash tools/gpu.sh r06f label:probe driver quick
  POSEVO_QUEUE_PROBE=0 bash tools/gpu.sh r06f label:noprobe driver
  cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
  export POSEVO_LIB_PATH=$PWD/pos_evolution_amd/libposevo_dbg.so
  for k in 0 1 3 2; do
    rm -rf $O/prof; POSEVO_DBG_SKIP=$k timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o s -- python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant --no-signed-steps --no-verify-steps --no-oracle-check > $O/bench_skip$k.json 2> $O/bench_skip$k.err
    timeout 120 python tools/rocpd_stats.py $O/prof/s_results.db $O/skip${k}_kernel_stats.txt > /dev/null 2>&1; echo "== skip $k: $(timeout 20 python tools/benchline.py < $O/bench_skip$k.json | cut -c1-60)"; grep -E "k_g1_accumulate|k_g1_finish|k_g1_tree|k_dbg" $O/skip${k}_kernel_stats.txt | cut -c1-150; rm -rf $O/prof
  done
  unset POSEVO_LIB_PATH
}

# call g: the legs' stream at the least priority (a hardware queue of the low-priority set, shared with none of the hot
# streams), one k_g2_aggregate_rows launch per batch; the square roots' exponentiation inlined (no scratch) against the called
# form (build/variants/libposevo_pownoinline.so) in the unaggregated-signature leg
call_g() {
  O=gpurun_out/r06g; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_g2.py tests/test_gpu_keyvalidate.py -x -q > $O/pytest_sig.log 2>&1; echo "[r06g] pairing + g2 tests rc $?"; tail -6 $O/pytest_sig.log
  timeout 300 python tools/engine_timeline.py --signed --lag 7 --steps 30 --show 8 > $O/engine_timeline_signed.txt 2>&1; tail -24 $O/engine_timeline_signed.txt | cut -c1-70; grep -c g2_decompress $O/engine_timeline_signed.txt
  for b in 4 8; do
    POSEVO_SIG_BATCH=$b timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant > $O/bench_sig$b.json 2> $O/bench_sig$b.err
    echo "[r06g] sig batch $b: rc $? $(python - <<PY
import json
d=json.loads(open("$O/bench_sig$b.json").read().strip().splitlines()[-1])
s=d.get("with_signatures",{}); u=d.get("with_unaggregated_signatures",{})
print("ms/step", round(d["ms_per_step"],4), "signed", d.get("ms_per_step_with_signatures"), "beside", s.get("ms_per_step_beside_another_handle"), "verified", s.get("steps_verified"), "| unagg ms/epoch", u.get("ms_per_epoch"), u.get("roofline_valu",{}).get("frac"), u.get("error"))
PY
)"
  done
  POSEVO_LIB_PATH=$PWD/build/variants/libposevo_pownoinline.so timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant > $O/bench_pownoinline.json 2> $O/bench_pownoinline.err
  python - <<PY
import json
d=json.loads(open("$O/bench_pownoinline.json").read().strip().splitlines()[-1])
u=d.get("with_unaggregated_signatures",{})
print("[r06g] called pow (208 B of scratch): signed", d.get("ms_per_step_with_signatures"), "unagg ms/epoch", u.get("ms_per_epoch"), u.get("roofline_valu",{}).get("frac"))
PY
  timeout 300 python tools/sig_epoch.py --calls 3 2>&1 | grep call
  POSEVO_LIB_PATH=$PWD/build/variants/libposevo_pownoinline.so timeout 300 python tools/sig_epoch.py --calls 3 2>&1 | grep call
  bash tools/gpu.sh r06g label:all tests
}

# call h: where the host of the signed step waits (POSEVO_HOST_TRACE), the engine's RCCL path over the stub library with two
# skewed ranks, the unaggregated leg call by call, the full default bench line (slot cadence included)
call_h() {
  O=gpurun_out/r06h; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_dist_custom.py -x -q -k "stub" > $O/pytest_stub.log 2>&1; echo "[r06h] stub rccl tests rc $?"; tail -25 $O/pytest_stub.log | cut -c1-220
  POSEVO_HOST_TRACE=1 timeout 300 python tools/engine_timeline.py --signed --lag 7 --steps 30 --show 8 > $O/engine_timeline_signed.txt 2> $O/host_trace_signed.txt; grep "posevo host" $O/host_trace_signed.txt | cut -c1-120
  bash tools/gpu.sh r06h label:full bench
  python - <<PY
import json
d=json.loads(open("$O/bench_full_full.json").read().strip().splitlines()[-1])
print("slot", d.get("slot_cadence")); print("unagg", {k:v for k,v in d.get("with_unaggregated_signatures",{}).items() if k!="detail"})
print("signed", d.get("ms_per_step_with_signatures"), "shuffle", d.get("ms_per_step_with_shuffle"))
PY
}

# call i: a batch's decompression on the accumulations' stream (between two of them), arena ring of up to 16 (lag 15), 8 steps
# per decompression; the unaggregated leg's host phases in the bench and in tools/sig_epoch.py; the slot cadence's timeline
call_i() {
  O=gpurun_out/r06i; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_g2.py tests/test_gpu_pipeline.py tests/test_gpu_pipeline_robust.py -x -q > $O/pytest_sig.log 2>&1; echo "[r06i] pairing + g2 + pipeline tests rc $?"; tail -6 $O/pytest_sig.log
  timeout 300 python tools/engine_timeline.py --signed --lag 15 --steps 44 --show 12 > $O/engine_timeline_signed.txt 2>&1; tail -22 $O/engine_timeline_signed.txt | cut -c1-70
  for b in 8 4; do
    POSEVO_SIG_BATCH=$b POSEVO_HOST_TRACE=1 POSEVO_SLOT_TIMELINE=$O/slot_timeline_$b.txt timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-shuffle-variant > $O/bench_sig$b.json 2> $O/bench_sig$b.err
    echo "[r06i] sig batch $b: rc $? $(python - <<PY
import json
d=json.loads(open("$O/bench_sig$b.json").read().strip().splitlines()[-1])
s=d.get("with_signatures",{}); u=d.get("with_unaggregated_signatures",{})
print("ms/step", round(d["ms_per_step"],4), "signed", d.get("ms_per_step_with_signatures"), "beside", s.get("ms_per_step_beside_another_handle"), "verified", s.get("steps_verified"), "| unagg", u.get("ms_per_call"), "| slot p50", d.get("slot_cadence",{}).get("slot_step_us_p50"))
PY
)"
  done
  grep "usig\." $O/bench_sig8.err | cut -c1-120
  POSEVO_HOST_TRACE=1 timeout 300 python tools/sig_epoch.py --calls 4 2>&1 | grep "call\|usig" | cut -c1-120
  sed -n 1,4p $O/slot_timeline_8.txt; awk 'NR>300 && NR<360' $O/slot_timeline_8.txt
}

# call j: arenas grow together (no per-arena first-use drain), statuses / index back through the pinned block
call_j() {
  O=gpurun_out/r06j; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_g2.py tests/test_gpu_pipeline.py tests/test_gpu_pipeline_robust.py -x -q > $O/pytest_sig.log 2>&1; echo "[r06j] pairing + g2 + pipeline tests rc $?"; tail -6 $O/pytest_sig.log
  timeout 300 python tools/engine_timeline.py --signed --lag 15 --steps 44 --show 12 > $O/engine_timeline_signed.txt 2>&1; tail -22 $O/engine_timeline_signed.txt | cut -c1-70
  for b in 8 4 1; do
    POSEVO_SIG_BATCH=$b POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-shuffle-variant --no-slot-cadence > $O/bench_sig$b.json 2> $O/bench_sig$b.err
    echo "[r06j] sig batch $b: rc $? $(python - <<PY
import json
d=json.loads(open("$O/bench_sig$b.json").read().strip().splitlines()[-1])
s=d.get("with_signatures",{}); u=d.get("with_unaggregated_signatures",{})
print("ms/step", round(d["ms_per_step"],4), "signed", d.get("ms_per_step_with_signatures"), "beside", s.get("ms_per_step_beside_another_handle"), "verified", s.get("steps_verified"), "| unagg", u.get("ms_per_call"), u.get("roofline_valu",{}).get("frac"))
PY
)"
  done
  grep "usig\.\|sagg\." $O/bench_sig8.err | cut -c1-120
}

# call k: the whole -m gpu suite at HEAD; the full default bench line; the per-rank load of an 8-way range shard (one rank over
# the engine's RCCL communicators, POSEVO_FORCE_DIST=1) at configs[3] / configs[4]; the driver's command
call_k() {
  O=gpurun_out/r06k; mkdir -p $O
  bash tools/gpu.sh r06k label:all tests
  bash tools/gpu.sh r06k label:full bench
  python - <<PY
import json
d=json.loads(open("$O/bench_full_full.json").read().strip().splitlines()[-1])
sc=d.get("slot_cadence",{}); u=d.get("with_unaggregated_signatures",{})
print("slot p50", sc.get("slot_step_us_p50"), "tick->head", sc.get("tick_to_head_us_p50"), "| signed", d.get("ms_per_step_with_signatures"), "shuffle", d.get("ms_per_step_with_shuffle"))
print("unagg", u.get("ms_per_epoch"), u.get("ms_per_call"), u.get("warmup_ms_per_call"), u.get("roofline_valu",{}).get("frac"))
PY
  for shape in "configs3 131072" "configs4 524288"; do set -- $shape
    POSEVO_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --shape $1 --validators $2 --steps 100 --warmup 6 --no-cpu-baseline --no-signed-steps --no-slot-cadence --no-shuffle-variant > $O/rank_$1.json 2> $O/rank_$1.err
    echo "[r06k] per-rank step of an 8-way range shard, $1: rc $? $(timeout 20 python tools/benchline.py < $O/rank_$1.json 2>/dev/null | cut -c1-200)"; done
  bash tools/gpu.sh r06k label:final driver
}

# call l: the driver's command three times (brackets of the roofline's kernel only inside the timed region), 200 steps, the
# per-rank load of an 8-way range shard, the bench tests
call_l() {
  O=gpurun_out/r06l; mkdir -p $O
  for i in 1 2 3; do bash tools/gpu.sh r06l label:d$i driver; done
  bash tools/gpu.sh r06l label:q quick
  for shape in "configs3 131072" "configs4 524288"; do set -- $shape
    POSEVO_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --shape $1 --validators $2 --steps 100 --warmup 6 --no-cpu-baseline --no-signed-steps --no-slot-cadence --no-shuffle-variant > $O/rank_$1.json 2> $O/rank_$1.err
    echo "[r06l] per-rank step of an 8-way range shard, $1: rc $? $(timeout 20 python tools/benchline.py < $O/rank_$1.json 2>/dev/null | cut -c1-200)"; done
  timeout 900 python -m pytest tests/test_gpu_shapes.py tests/test_gpu_pairing.py -x -q -k "signed or bench or knobs" > $O/pytest.log 2>&1; echo "[r06l] tests rc $?"; tail -4 $O/pytest.log
}

# call m: lag depth of the headline (an arena's first use no longer lands inside the timed region): the driver's command and
# 200 steps at lag 4 / 6 / 8 / 12
call_m() {
  O=gpurun_out/r06m; mkdir -p $O
  for lag in 4 6 8 12 4 8; do
    BENCH_ARGS="--lag $lag --no-signed-steps --no-slot-cadence --no-shuffle-variant --no-cpu-baseline" bash tools/gpu.sh r06m label:lag${lag} driver
  done
  for lag in 4 8; do
    BENCH_ARGS="--lag $lag" bash tools/gpu.sh r06m label:lag${lag} quick
  done
}

# call p: the round's profile set at HEAD -- the whole -m gpu suite, tools/profile_round.sh r06 (bench line; rocprofv3
# --kernel-trace --stats of the same command; FETCH_SIZE / WRITE_SIZE passes; the profiled timeline), the driver's exact command, the
# engine's own timelines (steady, cold 20, signed)
call_p() {
  O=gpurun_out/r06p; mkdir -p $O
  bash tools/gpu.sh r06p label:all tests
  bash tools/profile_round.sh r06
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_full.json 2> $O/bench_driver_full.err; echo "[r06p] driver's exact command rc $? $(timeout 20 python tools/benchline.py < $O/bench_driver_full.json | cut -c1-60)"
  timeout 300 python tools/engine_timeline.py --steps 24 --show 3 > $O/engine_timeline.txt 2>&1; tail -14 $O/engine_timeline.txt
  timeout 300 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20.txt 2>&1; head -4 $O/engine_timeline_cold20.txt | cut -c1-150
}

# call q: the tests added after call p (arena growths, device index), the unaggregated leg with its counts on the device, the
# engine's own timelines again (the collector paused)
call_q() {
  O=gpurun_out/r06q; mkdir -p $O
  timeout 900 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_g2.py -x -q > $O/pytest.log 2>&1; echo "[r06q] pairing + g2 tests rc $?"; tail -5 $O/pytest.log
  POSEVO_HOST_TRACE=1 timeout 300 python tools/sig_epoch.py --calls 4 2>&1 | grep "call\|usig" | cut -c1-120
  timeout 300 python tools/engine_timeline.py --steps 24 --show 3 > $O/engine_timeline.txt 2>&1; tail -12 $O/engine_timeline.txt
  timeout 300 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20.txt 2>&1; head -4 $O/engine_timeline_cold20.txt | cut -c1-150
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-slot-cadence --no-shuffle-variant > $O/bench.json 2> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); u=d["with_unaggregated_signatures"]
print("ms/step", d["ms_per_step"], "signed", d["ms_per_step_with_signatures"], "unagg", u["ms_per_epoch"], u["ms_per_call"], u["roofline_valu"]["frac"])
PY
}

# call r: the tree in S29 form (lane partials as 56 limbs, one conversion per group) -- the G1 parity tests, then A/B against the
# previous commit's library (pos_evolution_amd/libposevo_base.so) on this box, alternating; the engine's timeline for the tree
call_r() {
  O=gpurun_out/r06r; mkdir -p $O
  timeout 1200 python -m pytest tests/test_gpu_g1_accumulate.py tests/test_gpu_edge_cases.py tests/test_gpu_shapes.py tests/test_gpu_pairing.py tests/test_gpu_resident_rows.py -x -q > $O/pytest_g1.log 2>&1; echo "[r06r] g1 tests rc $?"; tail -8 $O/pytest_g1.log
  export BENCH_ARGS="--no-shuffle-variant"
  for i in 1 2; do
    bash tools/gpu.sh r06r label:new$i quick
    POSEVO_LIB_PATH=$PWD/pos_evolution_amd/libposevo_base.so bash tools/gpu.sh r06r label:base$i quick
  done
  BENCH_ARGS="--no-cpu-baseline --no-slot-cadence --no-signed-steps --no-shuffle-variant" bash tools/gpu.sh r06r label:new driver
  BENCH_ARGS="--no-cpu-baseline --no-slot-cadence --no-signed-steps --no-shuffle-variant" POSEVO_LIB_PATH=$PWD/pos_evolution_amd/libposevo_base.so bash tools/gpu.sh r06r label:base driver
  timeout 300 python tools/engine_timeline.py --steps 24 --show 2 > $O/engine_timeline_new.txt 2>&1; tail -30 $O/engine_timeline_new.txt
  POSEVO_LIB_PATH=$PWD/pos_evolution_amd/libposevo_base.so timeout 300 python tools/engine_timeline.py --steps 24 --show 2 > $O/engine_timeline_base.txt 2>&1; tail -30 $O/engine_timeline_base.txt
  unset BENCH_ARGS
  bash tools/gpu.sh r06r label:all tests
}

"call_$1"
