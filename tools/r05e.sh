#!/bin/bash
# round 5, call e: the whole GPU suite at HEAD, then the round's profile set (bench line, rocprof kernel stats, PMC passes,
# timeline) and the driver's command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05e; mkdir -p $O
bash tools/gpu.sh r05e label:all tests
bash tools/profile_round.sh r05
bash tools/gpu.sh r05e label:pair driver
timeout 300 python tools/engine_timeline.py --steps 24 --show 2 > $O/engine_timeline_paired.txt 2>&1; tail -14 $O/engine_timeline_paired.txt
timeout 300 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20_paired.txt 2>&1; head -30 $O/engine_timeline_cold20_paired.txt
