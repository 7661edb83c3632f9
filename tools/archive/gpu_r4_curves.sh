#!/bin/bash
# round 4: 300-iteration training curves of both solvers (LSIM_SOLVER override) on aliengo and aliengo_stairs + a soak of the default (TGS) solver.
# usage: bash tools/gpu_r4_curves.sh TAG [iterations]
TAG=${1:-r4c}
IT=${2:-300}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
for task in aliengo aliengo_stairs; do
  for sv in tgs pgs; do
    LSIM_SOLVER=$sv timeout 900 python tools/train_probe.py $IT $O/train_curve_${task}_${sv}_${IT}it.json $task 1 > $O/train_${task}_$sv.log 2>&1
    tail -1 $O/train_${task}_$sv.log
  done
done
timeout 900 python tools/soak.py 5000 4096 > $O/soak_tgs.log 2>&1; tail -8 $O/soak_tgs.log
