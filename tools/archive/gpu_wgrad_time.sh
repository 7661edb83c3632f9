#!/bin/bash
TAG=${1:-wg}; shift; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
python tools/wgrad_time.py | tee $O/wgrad_time.txt
for v in "$@"; do echo "== $v"; LSIM_LIB=$PWD/isaacgymloco_amd/csrc/variants/liblsim_$v.so python tools/wgrad_time.py | tee $O/wgrad_time_$v.txt; done
