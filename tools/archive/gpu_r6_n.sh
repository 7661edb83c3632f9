#!/bin/bash
# round 6, lease n: kernel A's per-phase and per-wave diagnostics on the final tree (round 5's published phase profiles were a stale variant's error text).   usage: bash tools/archive/gpu_r6_n.sh TAG
TAG=${1:-r6n}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for t in aliengo aliengo_stairs; do
  timeout 600 python tools/phase_profile.py $t 4096 < /dev/null > $O/phase_profile_$t.txt 2>&1; tail -3 $O/phase_profile_$t.txt | cut -c1-300
  timeout 600 python tools/wave_times.py $t 4096 < /dev/null > $O/wave_times_$t.txt 2>&1; tail -3 $O/wave_times_$t.txt | cut -c1-300
done
timeout 600 python tools/wave_times.py aliengo 4096 --phases < /dev/null > $O/wave_phases_aliengo.txt 2>&1; tail -3 $O/wave_phases_aliengo.txt | cut -c1-300
