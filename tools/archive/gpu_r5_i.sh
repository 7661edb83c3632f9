#!/bin/bash
# weight-gradient tests + stand-alone layer times (reduce launch included) for the product build and variants, then the train line A/B
TAG=$1; shift; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_learner.py -m gpu -q -k "wgrad or elu_backward or deferred or arena or skinny" 2>&1 | tail -2
python tools/wgrad_time.py | tee $O/wgrad_time.txt
for v in "$@"; do echo "== $v"; LSIM_LIB=$PWD/isaacgymloco_amd/csrc/variants/liblsim_$v.so python tools/wgrad_time.py | tee $O/wgrad_time_$v.txt; done
bash tools/gpu_ab_train.sh $TAG "$@"
