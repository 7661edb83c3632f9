#!/bin/bash
# round 5: learner tests + the train line (collection time = policy kernel with the rollout's stores + simulator)
TAG=${1:-r5o}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_learner.py tests/test_gpu_learner_golden.py -m gpu -q > $O/learner_tests.log 2>&1; tail -4 $O/learner_tests.log
for i in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_train_$i.json 2> $O/bench_train_$i.err
python - <<PY
import json
j=json.loads(open("$O/bench_train_$i.json").read().strip().splitlines()[-1])
print({k:j.get(k) for k in ("value","collection_s_per_iteration","learn_s_per_update","kernel_a_ms","iteration_spread_frac")})
PY
done
