O=gpurun_out/r03C; mkdir -p $O
shard() { POSEVO_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --validators 1048576 --steps 80 --warmup 6 --no-cpu-baseline > $O/sh_$1.json 2> $O/sh_$1.err; echo rc $?; tail -2 $O/sh_$1.err | cut -c1-200; }
POSEVO_TORCH_BACKEND=gloo shard gloo
shard nccl
python - <<PY
import json
for v in ("gloo","nccl"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03C/sh_{v}.json") if l.startswith("{")][-1])
        print(v, round(d["ms_per_step"],4), d.get("checked_against_oracle"), d["config"]["call_mode"][:80])
    except Exception as e: print(v,"fail",e)
PY
