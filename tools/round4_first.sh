#!/bin/bash
# The experiments round 3 prepared and could not measure (its GPU budget ended first), in ONE gpurun call so that they share
# a box:   gpurun --timeout 600 -- 'bash tools/round4_first.sh'      (≈ 4 minutes of box time)
# Outputs: gpurun_out/r04first/*.  Reading guide: DESIGN.md 8 (items 1-2) and 3.1 (the S29 form).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=gpurun_out/r04first
mkdir -p "$O"
# 0. what the step waits for: the engine's own in-situ timeline (no profiler in the process)
timeout 300 python tools/engine_timeline.py --steps 24 --show 3 > "$O/engine_timeline.txt" 2> "$O/engine_timeline.err"
echo "[r04first] engine timeline rc $?"; tail -8 "$O/engine_timeline.txt"
# 1. the S29 field form on hardware: device-vs-host check, products/s and mixed adds/s (tools/fpbench prints the 32-bit
#    form's: 57 G products/s, 4.6-4.9 G mixed adds/s)
( cd tools && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pos_evolution_amd/csrc -o fpbench29 fpbench29.hip \
    && timeout 120 ./fpbench29 ) > "$O/fpbench29.log" 2>&1
tail -12 "$O/fpbench29.log"
( cd tools && [ -x fpbench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pos_evolution_amd/csrc -o fpbench fpbench.hip; \
    timeout 120 ./fpbench ) > "$O/fpbench32.log" 2>&1
grep -E "mul_chain|madd_chain" "$O/fpbench32.log" | tail -6
# 2. the S29 accumulation kernel through the C ABI (parity with the default kernel and the closed form)
POSEVO_TEST_S29=1 timeout 300 python -m pytest tests/test_gpu_g1_s29.py -m gpu -x -q > "$O/pytest_s29.log" 2>&1
echo "[r04first] S29 parity rc $?"; tail -5 "$O/pytest_s29.log"
# 3. step-period A/Bs on this box: default | accumulation in front of k_tree | in front of k_votes | S29 | rows stream + a
#    fifth hardware queue
STEPS=120 BENCH_ARGS=--no-shuffle-variant bash tools/gpu.sh r04first \
    label:base quick \
    env:POSEVO_G1_DEFER=2 label:defer2 quick \
    env:POSEVO_G1_DEFER=3 label:defer3 quick \
    env:POSEVO_G1_DEFER=1 env:POSEVO_G1_S29=1 label:s29 quick \
    env:POSEVO_G1_S29=0 env:POSEVO_ROWS_STREAM=1 env:GPU_MAX_HW_QUEUES=5 label:rows_q5 quick \
    env:GPU_MAX_HW_QUEUES=6 label:rows_q6 quick
