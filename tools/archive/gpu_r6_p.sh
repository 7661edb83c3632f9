#!/bin/bash
# round 6, lease p: soak on the final tree, every task, 20 000 steps x 4096 robots, the kernel's own non-finite counter read at the end.   usage: bash tools/archive/gpu_r6_p.sh TAG
TAG=${1:-r6p}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/soak.py 20000 4096 < /dev/null > $O/soak_20000steps.txt 2>&1; cat $O/soak_20000steps.txt | cut -c1-400
