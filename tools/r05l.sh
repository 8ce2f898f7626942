#!/bin/bash
# round 5, call l: bench.py's stdout carries exactly one line -- the driver's command, the launcher's form of it, and the N > 1
# path (one rank over the engine's RCCL, whose banner and gloo's used to land on stdout)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05l; mkdir -p $O
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.out 2> $O/driver.err; echo "[r05l] driver command rc $? stdout lines $(wc -l < $O/driver.out)"; timeout 20 python tools/benchline.py < $O/driver.out | cut -c1-120
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-slot-cadence --no-signed-steps > $O/launcher.out 2> $O/launcher.err; echo "[r05l] launcher rc $? stdout lines $(wc -l < $O/launcher.out)"; timeout 20 python tools/benchline.py < $O/launcher.out | cut -c1-120
POSEVO_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --validators 131072 --steps 20 --warmup 5 > $O/dist.out 2> $O/dist.err; echo "[r05l] N > 1 path, one rank: rc $? stdout lines $(wc -l < $O/dist.out)"; timeout 20 python tools/benchline.py < $O/dist.out | cut -c1-120; grep -c "Gloo\|RCCL version" $O/dist.err
