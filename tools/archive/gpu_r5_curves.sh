#!/bin/bash
# round 5: 300-iteration training curves on the final tree (centre-of-mass velocities, TGS with the limit pass: the defaults) for aliengo and
# aliengo_stairs, and aliengo under round 4's conventions for comparison; a 5000-step soak of the default configuration.
# usage: bash tools/gpu_r5_curves.sh TAG [iterations]
TAG=${1:-r5c}
IT=${2:-300}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
for task in aliengo aliengo_stairs; do
  timeout 900 python tools/train_probe.py $IT $O/train_curve_${task}_${IT}it.json $task 1 > $O/train_${task}.log 2>&1; tail -1 $O/train_${task}.log
done
LSIM_LIN_VEL=origin LSIM_TGS_LIMIT_PASSES=0 timeout 900 python tools/train_probe.py $IT $O/train_curve_aliengo_r4_conventions_${IT}it.json aliengo 1 > $O/train_aliengo_r4conv.log 2>&1; tail -1 $O/train_aliengo_r4conv.log
timeout 900 python tools/soak.py 5000 4096 > $O/soak.log 2>&1; tail -8 $O/soak.log
