set -x
mkdir -p gpurun_out/r03a
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_resident_rows.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r03a/pytest_resident.log
cat gpurun_out/r03a/pytest_resident.log | tail -15
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r03a/bench_resident_k40.json 2> gpurun_out/r03a/bench_resident_k40.err
tail -c 3000 gpurun_out/r03a/bench_resident_k40.json; tail -5 gpurun_out/r03a/bench_resident_k40.err
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --host-rows > gpurun_out/r03a/bench_hostrows_k40.json 2> gpurun_out/r03a/bench_hostrows_k40.err
tail -c 1500 gpurun_out/r03a/bench_hostrows_k40.json
