#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02q_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02q_tests.log
tail -n 6 gpurun_out/r02q_tests.log
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/r02q_bench_trace.json 2> gpurun_out/r02q_hosttrace.txt
grep "posevo host" gpurun_out/r02q_hosttrace.txt | grep "comm\."
python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, ".")
import pos_evolution_amd as pea, pos_evolution_amd.synth as synth
e = pea.Engine()
n = 1 << 20
pts = synth.registry_points(e, n)
e.set_validators(synth.balances(n, 1), np.ones(n, dtype=np.uint8), pts)
for rep in range(3):
    t = time.perf_counter(); st = e.g1_key_validate(); dt = time.perf_counter() - t
    print("key_validate 1M keys: %.1f ms, all valid: %s" % (dt * 1e3, bool((st == 0).all())))
import hashlib
for want in (True, False):
    ts = []
    for rep in range(5):
        t = time.perf_counter(); e.compute_committees(10 + rep, hashlib.sha256(bytes([rep])).digest(), n, 2048, 90, want_result=want); ts.append(time.perf_counter() - t)
    print("compute_committees 1M x 90 rounds, want_result=%s: %s ms" % (want, ["%.2f" % (x * 1e3) for x in ts]))
PY
