#!/bin/bash
# round 6, lease b: kernel sequence of one AMP minibatch (trace of tools/amp_update_probe.py).  usage: bash tools/gpu_r6_b.sh TAG
TAG=${1:-r6b}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o ampupd -- python3 $R/tools/amp_update_probe.py 1 > $R/$O/probe.log 2>&1
cd $R; tail -3 $O/probe.log
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_seq.py $f 330 > $O/seq_last_minibatch.txt
tail -300 $O/seq_last_minibatch.txt | awk '{print}' | head -300
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
