#!/bin/bash
# A/B of the fused policy kernel alone: the product build against isaacgymloco_amd/csrc/variants/liblsim_NAME.so (LSIM_LIB), three runs each,
# interleaved.   usage: bash tools/policy_ab.sh NAME
for i in 1 2 3; do
  unset LSIM_LIB; timeout 120 python tools/policy_time.py 2>/dev/null | sed 's/^/product   /'
  LSIM_LIB=$PWD/isaacgymloco_amd/csrc/variants/liblsim_$1.so timeout 120 python tools/policy_time.py 2>/dev/null | sed "s/^/$1  /"
done
