#!/bin/bash
# round 6, lease aa: the stairs task for 4000 iterations on the final tree (seed 1), checkpoint + closed-loop evaluation.   usage: bash tools/archive/gpu_r6_aa.sh TAG
TAG=${1:-r6aa}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python tools/train_probe.py 4000 $O/train_curve_aliengo_stairs_4000it.json aliengo_stairs 1 $O/policy_aliengo_stairs_4000it.pt < /dev/null > $O/train_stairs_4000.log 2>&1; tail -2 $O/train_stairs_4000.log | cut -c1-700
