#!/bin/bash
# round 6, lease l: the 1000-iteration schedules of aliengo and aliengo_stairs on the FINAL tree (six calf collision points), with checkpoints + closed-loop evaluation;
# 300 iterations of aliengo_amp.   usage: bash tools/archive/gpu_r6_l.sh TAG
TAG=${1:-r6l}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/train_probe.py 1000 $O/train_curve_aliengo_1000it.json aliengo 1 $O/policy_aliengo_1000it.pt < /dev/null > $O/train_aliengo.log 2>&1; tail -2 $O/train_aliengo.log | cut -c1-700
timeout 600 python tools/train_probe.py 1000 $O/train_curve_aliengo_stairs_1000it.json aliengo_stairs 1 $O/policy_aliengo_stairs_1000it.pt < /dev/null > $O/train_stairs.log 2>&1; tail -2 $O/train_stairs.log | cut -c1-700
timeout 600 python tools/train_probe_amp.py 300 < /dev/null > $O/train_amp.log 2>&1; tail -2 $O/train_amp.log | cut -c1-600
