#!/bin/bash
# round 6, lease c: the discriminator update's fused kernels -- tests, then the AMP bench line on / off, then kernel statistics.   usage: bash tools/gpu_r6_c.sh TAG
TAG=${1:-r6c}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_amp_update.py tests/test_gpu_amp_step.py tests/test_gpu_learner_golden.py tests/test_gpu_learner.py -m gpu -q -x > $O/tests.log 2>&1; tail -25 $O/tests.log
for v in 1 0 1 0; do
  LSIM_AMP_FUSED_UPDATE=$v timeout 600 python bench.py --task aliengo_amp --steps 100 --warmup 100 --no-cpu-baseline > $O/bench_amp_upd$v.log 2>&1; tail -1 $O/bench_amp_upd$v.log > $O/bench_amp_upd$v.json
  python -c "import json; d=json.load(open('$O/bench_amp_upd$v.json')); print('fused update=$v amp value', round(d['value']), 'coll', round(d['collection_s_per_iteration'],4), 'learn', round(d['learn_s_per_update'],4))"
done
bash tools/gpu_amp.sh $TAG/amp 2>&1 | tail -32
