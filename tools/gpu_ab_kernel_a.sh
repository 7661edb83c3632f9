#!/bin/bash
# kernel A alone (env-only line, flat and stairs), the product against build variants of the library (LSIM_LIB), interleaved over three rounds.
# usage: bash tools/gpu_ab_kernel_a.sh TAG name=path/to/liblsim_variant.so ...      (variants: python -c "from isaacgymloco_amd.csrc import build; build.build_variant(out, flags)")
TAG=$1; shift; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for round in 1 2 3; do
  for v in product "$@"; do
    name=${v%%=*}; path=${v#*=}
    for task in aliengo aliengo_stairs; do
      if [ $name = product ]; then unset LSIM_LIB; else export LSIM_LIB=$PWD/$path; fi
      timeout 300 python bench.py --mode env --task $task --steps 500 --warmup 50 --no-cpu-baseline < /dev/null > $O/env_${name}_${task}_$round.log 2>&1
      timeout 20 python -c "import json; d=json.loads(open('$O/env_${name}_${task}_$round.log').read().strip().splitlines()[-1]); print('$name $task round $round kernel_a %.4f ms  value %.2f M  nonfinite %s' % (d['kernel_a_ms'], d['value'] / 1e6, d.get('nonfinite_envs')))" < /dev/null
    done
  done
done
unset LSIM_LIB
