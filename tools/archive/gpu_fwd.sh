#!/bin/bash
# lsim_linear_elu_forward microbenchmark: the product build and probe variants (LSIM_LIB).  usage: bash tools/gpu_fwd.sh TAG [variant ...]
TAG=${1:-fwd}; shift; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
{ echo "== product"; timeout 600 python tools/fwd_time.py
  for v in "$@"; do echo "== $v"; LSIM_LIB=$PWD/isaacgymloco_amd/csrc/variants/liblsim_$v.so timeout 600 python tools/fwd_time.py; done; } 2>&1 | grep -v amdgpu.ids | tee $O/fwd_time.txt
