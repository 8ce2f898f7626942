#!/bin/bash
# Produces the artifacts kept under profiles/ for one round, on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh r01'
# 1. bench.py full line (with cpu_baseline)         -> gpurun_out/<tag>_bench_full.json
# 2. rocprofv3 --kernel-trace --stats of bench.py   -> gpurun_out/<tag>_kernel_stats.txt (+ the bench line under rocprof)
#    (--no-slot-cadence --no-signed-steps: the per-slot run launches the same kernels at 1/32 of the size; mixed in, the per-kernel averages
#    and the PMC medians would describe neither workload)
# 3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, kernel-trace only) -> gpurun_out/<tag>_pmc_{fetch,write}.json
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out
mkdir -p $OUT/prof_$TAG $OUT/pmc_$TAG
[ -n "${SKIP_FULL:-}" ] || python bench.py > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_bench_full.err
# (--no-verify-steps --no-shuffle-variant --no-oracle-check: the replays run the same kernels ALONE (synchronous calls), the
# variant steps beside a shuffle; with them in the trace rocprof's per-kernel average is over three different workloads and
# cannot be held against the bench line's own average over the timed steps)
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o $TAG -- python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-slot-cadence --no-signed-steps \
    --no-verify-steps --no-shuffle-variant --no-oracle-check > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof_$TAG/err.log
python tools/rocpd_stats.py $OUT/prof_$TAG/${TAG}_results.db $OUT/${TAG}_kernel_stats.txt > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_$TAG -o fetch -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-slot-cadence --no-signed-steps \
    > $OUT/pmc_$TAG/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_$TAG -o write -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-slot-cadence --no-signed-steps \
    > $OUT/pmc_$TAG/write.log 2>&1
python tools/rocpd_pmc.py $OUT/pmc_$TAG/fetch_results.db FETCH_SIZE > $OUT/${TAG}_pmc_fetch.json
python tools/rocpd_pmc.py $OUT/pmc_$TAG/write_results.db WRITE_SIZE > $OUT/${TAG}_pmc_write.json
python tools/rocpd_timeline.py $OUT/prof_$TAG/${TAG}_results.db 20 3 > $OUT/${TAG}_timeline.txt 2>&1
# shape-keyed HBM traffic entry for profiles/hbm_traffic.json (bench.py quotes it only for this shape)
python - <<PY > $OUT/${TAG}_hbm_traffic_entry.json
import json
f=json.load(open("$OUT/${TAG}_pmc_fetch.json")); w=json.load(open("$OUT/${TAG}_pmc_write.json"))
def kib(d,k): return d[k]["median"]*1024.0 if k in d else None
fa, wa = kib(f,"k_g1_accumulate"), kib(w,"k_g1_accumulate")
fv, wv = kib(f,"k_votes<2>") or kib(f,"k_votes<1>") or kib(f,"k_votes"), kib(w,"k_votes<2>") or kib(w,"k_votes<1>") or kib(w,"k_votes")
print(json.dumps({"1048576,2048,4096": {
  "k_g1_accumulate_fetch_raw": fa, "k_g1_accumulate_write_raw": wa,
  "k_g1_accumulate_bytes_per_launch": (2*fa + wa) if fa is not None and wa is not None else None,
  "k_votes_bytes_per_launch": (2*fv + wv) if fv is not None and wv is not None else None,
  "source": "profiles/${TAG}_pmc_fetch.json + ${TAG}_pmc_write.json (FETCH_SIZE x 2 per MI355X_MICROARCH.md, WRITE_SIZE as reported; per-kernel median over launches)"}}, indent=1))
PY
cat $OUT/${TAG}_hbm_traffic_entry.json; cat $OUT/${TAG}_timeline.txt | tail -30
rm -rf $OUT/prof_$TAG/*.db $OUT/pmc_$TAG/*.db     # the databases are tens of MB; the summaries are what is kept
head -c 600 $OUT/${TAG}_bench_full.json; echo; cut -c1-130 $OUT/${TAG}_kernel_stats.txt | head -12
cat $OUT/${TAG}_pmc_fetch.json | head -30
