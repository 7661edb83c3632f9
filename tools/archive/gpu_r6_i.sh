#!/bin/bash
# round 6, lease i: free-running parity with the fork classification.   usage: bash tools/archive/gpu_r6_i.sh TAG
TAG=${1:-r6i}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
rm -f $O/free_running_parity.jsonl
LSIM_PARITY_REPORT=$O/free_running_parity.jsonl timeout 1200 python -m pytest tests/test_gpu_free_running.py -m gpu -q -s -k "free_running" < /dev/null > $O/tests.log 2>&1; grep "forks\|passed\|failed" $O/tests.log | cut -c1-400
