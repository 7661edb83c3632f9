export TMPDIR=/tmp
O=gpurun_out/r04l; mkdir -p $O
run() { name=$1; shift; "$@" > $O/$name.log 2>&1; tail -1 $O/$name.log > $O/$name.json; }
LSIM_DEBUG_FORCE_COLLECTIVES=1 run force_default timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --no-cpu-baseline
GPU_MAX_HW_QUEUES=8 LSIM_DEBUG_FORCE_COLLECTIVES=1 run force_q8 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --no-cpu-baseline
GPU_MAX_HW_QUEUES=2 LSIM_DEBUG_FORCE_COLLECTIVES=1 run force_q2 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 1 --no-cpu-baseline
LSIM_UPDATE_STREAMS=0 LSIM_DEBUG_FORCE_COLLECTIVES=1 run force_onestream timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --no-cpu-baseline
LSIM_UPDATE_STREAMS=0 run plain_onestream timeout 600 python bench.py --no-cpu-baseline
GPU_MAX_HW_QUEUES=8 run plain_q8 timeout 600 python bench.py --no-cpu-baseline
python - <<PY
import json
for f in ("force_default","force_q8","force_q2","force_onestream","plain_onestream","plain_q8"):
    try:
        j=json.load(open("$O/"+f+".json"))
        print(f, {k:j.get(k) for k in ("value","collection_s_per_iteration","learn_s_per_update","iteration_wall_s_min_median_max")})
    except Exception as e: print(f, "failed", e)
PY
