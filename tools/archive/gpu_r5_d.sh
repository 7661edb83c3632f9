#!/bin/bash
# round 5: the free-running parity module with its measured tables
TAG=${1:-r5e}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
rm -f $O/free_running_parity.jsonl
LSIM_PARITY_REPORT=$O/free_running_parity.jsonl timeout 1800 python -m pytest tests/test_gpu_free_running.py -m gpu -q -s > $O/free_running.log 2>&1; tail -30 $O/free_running.log | cut -c1-400
