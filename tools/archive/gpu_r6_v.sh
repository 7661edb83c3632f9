#!/bin/bash
# round 6, lease v: index upload without a pipeline drain (learn/amp.py: _IndexUploader): GPU suite, then lease u's lines again.   usage: bash tools/archive/gpu_r6_v.sh TAG
TAG=${1:-r6v}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q < /dev/null > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
bash tools/archive/gpu_r6_u.sh $TAG
