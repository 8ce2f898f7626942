#!/bin/bash
# round 5, call i: the whole GPU suite with the new defaults, the round's profile set (bench line, rocprof kernel stats, PMC
# passes), the driver's command, the engine's own timelines, the accumulation alone on the same box (tools/accbench)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05i; mkdir -p $O
bash tools/gpu.sh r05i label:all tests
bash tools/profile_round.sh r05
bash tools/gpu.sh r05i label:final driver
timeout 200 python tools/engine_timeline.py --steps 30 --show 3 > $O/engine_timeline.txt 2>&1; tail -12 $O/engine_timeline.txt
timeout 200 python tools/engine_timeline.py --cold 20 --steps 26 > $O/engine_timeline_cold20.txt 2>&1; head -4 $O/engine_timeline_cold20.txt
[ -x tools/accbench ] && (timeout 120 tools/accbench > $O/accbench.txt 2>&1; tail -6 $O/accbench.txt)
