#!/bin/bash
# Produces the artifacts kept under profiles/ for one round, on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh r01'
# 1. bench.py full line (with cpu_baseline)         -> gpurun_out/<tag>_bench_full.json
# 2. rocprofv3 --kernel-trace --stats of bench.py   -> gpurun_out/<tag>_kernel_stats.txt (+ the bench line under rocprof)
# 3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, kernel-trace only) -> gpurun_out/<tag>_pmc_{fetch,write}.json
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out
mkdir -p $OUT/prof_$TAG $OUT/pmc_$TAG
python bench.py > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_bench_full.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o $TAG -- python bench.py --steps 60 --warmup 6 --no-cpu-baseline \
    > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof_$TAG/err.log
python tools/rocpd_stats.py $OUT/prof_$TAG/${TAG}_results.db $OUT/${TAG}_kernel_stats.txt > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_$TAG -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline \
    > $OUT/pmc_$TAG/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_$TAG -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline \
    > $OUT/pmc_$TAG/write.log 2>&1
python tools/rocpd_pmc.py $OUT/pmc_$TAG/fetch_results.db FETCH_SIZE > $OUT/${TAG}_pmc_fetch.json
python tools/rocpd_pmc.py $OUT/pmc_$TAG/write_results.db WRITE_SIZE > $OUT/${TAG}_pmc_write.json
rm -rf $OUT/prof_$TAG/*.db $OUT/pmc_$TAG/*.db     # the databases are tens of MB; the summaries are what is kept
head -c 600 $OUT/${TAG}_bench_full.json; echo; cut -c1-130 $OUT/${TAG}_kernel_stats.txt | head -12
cat $OUT/${TAG}_pmc_fetch.json | head -30
