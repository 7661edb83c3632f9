#!/bin/bash
# round 6, lease j: the weight-gradient kernels alone: product, no operand loads in the loop, no MFMAs.   usage: bash tools/archive/gpu_r6_j.sh TAG
TAG=${1:-r6j}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for v in product wg_noload wg_nomfma; do
  if [ $v = product ]; then unset LSIM_LIB; else export LSIM_LIB=$PWD/tests/_build/variants/liblsim_$v.so; fi
  echo "== $v"; timeout 300 python tools/wgrad_time.py < /dev/null 2>&1 | grep -v amdgpu | head -4
done 2>&1 | tee $O/wgrad_knockouts.txt
