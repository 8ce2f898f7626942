set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident_rows.py -x -q -m gpu 2>&1 | tail -30 > $O/pytest_resident.log; tail -5 $O/pytest_resident.log
POSEVO_HOST_TRACE=1 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-verify-steps > $O/bench_resident_trace.json 2> $O/bench_resident_trace.err
grep "posevo host" $O/bench_resident_trace.err | cut -c1-120
timeout 10 python tools/benchline.py < $O/bench_resident_trace.json
rocprofv3 --kernel-trace -d $O/prof -o tl -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-verify-steps > $O/bench_under_rocprof.json 2> $O/prof_err.log
python tools/rocpd_timeline.py $O/prof/tl_results.db 20 3 > $O/timeline_resident.txt 2>&1
cat $O/timeline_resident.txt
rm -rf $O/prof/*.db
