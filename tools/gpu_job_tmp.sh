#!/bin/bash
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python tools/phase_profile.py aliengo 256 2>&1 | grep -v amdgpu.ids | cut -c1-60,150-215 > gpurun_out/pp256.txt
timeout 300 python tools/phase_profile.py aliengo 4096 2>&1 | grep -v amdgpu.ids | cut -c1-60,150-215 > gpurun_out/pp4096.txt
paste -d'|' <(cut -c1-60 gpurun_out/pp256.txt) <(cut -c28-60 gpurun_out/pp4096.txt) <(cut -c61-130 gpurun_out/pp4096.txt)
