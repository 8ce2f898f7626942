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

"call_$1"
