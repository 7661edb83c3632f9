#!/bin/bash
# round 6, lease g: the driver's command three times (with the CPU baseline), the AMP line twice.   usage: bash tools/archive/gpu_r6_g.sh TAG
TAG=${1:-r6g}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null > $O/driver_$i.log 2>&1; tail -1 $O/driver_$i.log > $O/driver_$i.json
  timeout 20 python -c "import json; d=json.load(open('$O/driver_$i.json')); print('driver cmd $i', round(d['value']), d['collection_s_per_iteration'], d['learn_s_per_update'], [round(c+l,4) for c,l in d['collection_learn_s_by_iteration']], d['warmup_iterations'], round(d['cpu_baseline']['value']))" < /dev/null; done
for i in 1 2; do timeout 600 python bench.py --task aliengo_amp --no-cpu-baseline < /dev/null > $O/amp_$i.log 2>&1; tail -1 $O/amp_$i.log > $O/amp_$i.json
  timeout 20 python -c "import json; d=json.load(open('$O/amp_$i.json')); print('amp $i', round(d['value']), d['collection_s_per_iteration'], d['learn_s_per_update'])" < /dev/null; done
