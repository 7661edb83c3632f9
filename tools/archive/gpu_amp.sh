#!/bin/bash
# AMP configuration (BASELINE config 4): bench line + rocprofv3 kernel statistics of the same command.  usage: bash tools/gpu_amp.sh TAG
TAG=${1:-amp}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --task aliengo_amp --steps 200 --warmup 100 --no-cpu-baseline > $O/bench_amp.log 2>&1; tail -1 $O/bench_amp.log > $O/bench_amp.json
python -c "import json; d=json.load(open('$O/bench_amp.json')); print('amp value', round(d['value']), 'coll', round(d['collection_s_per_iteration'],4), 'learn', round(d['learn_s_per_update'],4), 'A', round(d['kernel_a_ms'],4))"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o amp -- python3 bench.py --task aliengo_amp --steps 100 --warmup 100 --no-cpu-baseline > $O/prof.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/prof/amp_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time ms', tot/1e6)
for r in rows[:28]:
    print('%6.2f%% %8.2f ms %6s calls  %s' % (100*float(r['TotalDurationNs'])/tot, float(r['TotalDurationNs'])/1e6, r['Calls'], r['Name'][:110]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete
