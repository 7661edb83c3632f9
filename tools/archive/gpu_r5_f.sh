#!/bin/bash
# round 5: fused policy kernel -- correctness test, then its time alone: product build against variants (isaacgymloco_amd/csrc/variants/liblsim_NAME.so)
TAG=${1:-r5h}; shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_learner.py -m gpu -q -k "policy" > $O/policy_tests.log 2>&1; tail -3 $O/policy_tests.log
for i in 1 2 3; do
  unset LSIM_LIB; timeout 120 python tools/policy_time.py 2>/dev/null | sed 's/^/product   /'
  for v in "$@"; do LSIM_LIB=$PWD/isaacgymloco_amd/csrc/variants/liblsim_$v.so timeout 120 python tools/policy_time.py 2>/dev/null | sed "s/^/$v  /"; done
done | tee $O/policy_ab.txt
