#!/bin/bash
TAG=${1:-gt}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_learner.py -m gpu -q -k "gather" 2>&1 | tail -2
python tools/gather_time.py | tee $O/gather_time.txt
echo "== before"; LSIM_LIB=$PWD/isaacgymloco_amd/csrc/variants/liblsim_gather_before.so python tools/gather_time.py | tee $O/gather_time_before.txt
