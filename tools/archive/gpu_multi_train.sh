#!/bin/bash
# eight consecutive default-size train lines with the per-iteration update times (is the update stable from run to run on this lease?)   usage: bash tools/gpu_multi_train.sh
for i in 1 2 3 4 5 6 7 8; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run', $i, 'value %.3f M' % (j['value']/1e6), 'update %.2f ms' % (j['learn_s_per_update']*1e3), [round(x[1]*1e3,1) for x in j['collection_learn_s_by_iteration']])"; done
