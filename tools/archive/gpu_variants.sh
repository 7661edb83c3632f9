#!/bin/bash
# A/B of build variants of liblsim.so on the env-only bench (kernel A / B durations): bash tools/gpu_variants.sh NAME [NAME ...]
# (isaacgymloco_amd/csrc/variants/liblsim_NAME.so, selected through LSIM_LIB; "base" = the product build)
for v in "$@"; do
  if [ "$v" = base ]; then unset LSIM_LIB; else export LSIM_LIB=$PWD/isaacgymloco_amd/csrc/variants/liblsim_$v.so; fi
  for rep in 1 2; do
    timeout 300 python bench.py --mode env --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), 'A', round(d['kernel_a_ms'],4), 'B', round(d['kernel_b_ms'],4))"
  done
done
