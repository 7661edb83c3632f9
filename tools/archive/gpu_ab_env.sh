#!/bin/bash
# A/B of kernel A alone: the product build against variants (isaacgymloco_amd/csrc/variants/liblsim_NAME.so via LSIM_LIB), env-only line, interleaved
# usage: bash tools/gpu_ab_env.sh TAG [--task T] NAME...
TAG=$1; shift
TASK=aliengo
if [ "$1" = "--task" ]; then TASK=$2; shift; shift; fi
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
one() { local name=$1; shift
  env "$@" timeout 300 python bench.py --mode env --task $TASK --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'kernel_a_ms %.5f  value %.3f M' % (j['kernel_a_ms'], j['value']/1e6))"; }
for i in 1 2 3; do
  one product
  for v in "$@"; do one $v LSIM_LIB=$PWD/isaacgymloco_amd/csrc/variants/liblsim_$v.so; done
done | tee $O/ab_env_$TASK.txt
