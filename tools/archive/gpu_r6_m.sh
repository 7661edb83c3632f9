#!/bin/bash
# round 6, lease m: seed spread of the 1000-iteration aliengo_stairs schedule on the final tree (seeds 2, 3) and one 2000-iteration run (seed 1).   usage: bash tools/archive/gpu_r6_m.sh TAG
TAG=${1:-r6m}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for s in 2 3; do
  timeout 600 python tools/train_probe.py 1000 $O/train_curve_aliengo_stairs_1000it_seed$s.json aliengo_stairs $s $O/policy_aliengo_stairs_1000it_seed$s.pt < /dev/null > $O/train_stairs_seed$s.log 2>&1; tail -2 $O/train_stairs_seed$s.log | cut -c1-700
done
timeout 900 python tools/train_probe.py 2000 $O/train_curve_aliengo_stairs_2000it.json aliengo_stairs 1 $O/policy_aliengo_stairs_2000it.pt < /dev/null > $O/train_stairs_2000.log 2>&1; tail -2 $O/train_stairs_2000.log | cut -c1-700
