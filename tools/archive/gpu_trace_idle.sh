#!/bin/bash
# device idle inside the last PPO update of the default train line (rocprofv3 kernel trace, analysed on the box: the trace is too big to bring back)
# usage: bash tools/gpu_trace_idle.sh TAG
TAG=${1:-idle}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/tr; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); cd $GRAFT_REPO_ROOT
python tools/trace_idle.py $f 24 | tee $O/trace_idle.txt
python tools/trace_seq.py $f > $O/trace_seq.txt 2>&1; head -5 $O/trace_seq.txt
