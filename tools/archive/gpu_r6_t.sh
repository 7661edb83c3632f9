#!/bin/bash
# round 6, lease t: the policy kernel's 16-row form against the 32-row product, alone.   usage: bash tools/archive/gpu_r6_t.sh TAG
TAG=${1:-r6t}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
  timeout 120 python tools/policy_time.py < /dev/null 2>/dev/null
  LSIM_POLICY_ROWS16=1 timeout 120 python tools/policy_time.py < /dev/null 2>/dev/null
done | tee $O/policy_rows.txt
