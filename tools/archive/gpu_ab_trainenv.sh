#!/bin/bash
# A/B of the train line under environment switches, interleaved.  usage: bash tools/gpu_ab_trainenv.sh TAG NAME=ENVVAR=VALUE ...   (product = no switch)
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
one() { local name=$1; shift
  env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee -a $O/lines_$name.jsonl | python tools/line_summary.py $name; }
for i in 1 2 3; do
  one product
  for v in "$@"; do one ${v%%=*} ${v#*=}; done
done | tee $O/ab_train.txt
