#!/bin/bash
# round 6, lease a: the fused AMP step -- its tests, the rollout test that contains it, its time alone, the policy kernel's time (its layer
# function was refactored), the AMP bench line on / off.   usage: bash tools/gpu_r6_a.sh TAG
TAG=${1:-r6a}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_amp_step.py tests/test_gpu_learner_golden.py tests/test_gpu_learner.py -m gpu -q -x > $O/tests.log 2>&1; tail -15 $O/tests.log
for n in 4096 8192 32768; do timeout 120 python tools/amp_step_time.py $n; done 2>&1 | tee $O/amp_step_time.txt
LSIM_AMP_NSPLIT=1 timeout 120 python tools/amp_step_time.py 4096 2>&1 | tee -a $O/amp_step_time.txt
for i in 1 2; do timeout 120 python tools/policy_time.py; done 2>&1 | tee $O/policy_time.txt
for v in 1 0 1 0; do
  LSIM_AMP_FUSED_STEP=$v timeout 600 python bench.py --task aliengo_amp --steps 100 --warmup 100 --no-cpu-baseline > $O/bench_amp_fused$v.log 2>&1; tail -1 $O/bench_amp_fused$v.log > $O/bench_amp_fused$v.json
  python -c "import json; d=json.load(open('$O/bench_amp_fused$v.json')); print('fused=$v amp value', round(d['value']), 'coll', round(d['collection_s_per_iteration'],4), 'learn', round(d['learn_s_per_update'],4))"
done
