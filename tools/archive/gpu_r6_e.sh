#!/bin/bash
# round 6, lease e: the reference's 1000-iteration schedules of aliengo and aliengo_stairs with the final tree; checkpoints + closed-loop evaluation
# under the trained policies (tools/train_probe.py ... checkpoint.pt); then the 2-rank line with its per-rank diagnostics.   usage: bash tools/gpu_r6_e.sh TAG
TAG=${1:-r6e}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/train_probe.py 1000 $O/train_curve_aliengo_1000it.json aliengo 1 $O/policy_aliengo_1000it.pt < /dev/null > $O/train_aliengo.log 2>&1; tail -2 $O/train_aliengo.log
timeout 600 python tools/train_probe.py 1000 $O/train_curve_aliengo_stairs_1000it.json aliengo_stairs 1 $O/policy_aliengo_stairs_1000it.pt < /dev/null > $O/train_stairs.log 2>&1; tail -2 $O/train_stairs.log
LSIM_DEBUG_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --no-cpu-baseline < /dev/null > $O/bench_2ranks_debug.log 2>&1; tail -1 $O/bench_2ranks_debug.log > $O/bench_2ranks_debug.json
timeout 20 python -c "import json; d=json.load(open('$O/bench_2ranks_debug.json')); print('2 ranks', round(d['value']), d['per_rank'], d['iteration_skew_s_max_mean'])" < /dev/null
LSIM_DEBUG_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline < /dev/null > $O/bench_rccl_1rank.log 2>&1; tail -1 $O/bench_rccl_1rank.log > $O/bench_rccl_1rank.json
timeout 20 python -c "import json; d=json.load(open('$O/bench_rccl_1rank.json')); print('rccl 1 rank', round(d['value']), d['per_rank'])" < /dev/null
timeout 900 python -m pytest tests/test_bench_cli.py tests/test_nonfinite_counter.py -m gpu -q -x < /dev/null > $O/tests.log 2>&1; tail -3 $O/tests.log
