#!/bin/bash
# round 6, lease y: host profile of one update (cProfile), default and AMP.   usage: bash tools/archive/gpu_r6_y.sh TAG
TAG=${1:-r6y}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for t in aliengo aliengo_amp; do timeout 600 python tools/update_host_profile.py $t 45 < /dev/null > $O/update_host_profile_$t.txt 2>&1; head -3 $O/update_host_profile_$t.txt | cut -c1-200; done
