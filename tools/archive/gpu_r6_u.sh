#!/bin/bash
# round 6, lease u: is the update bounded by the host's launch rate?  update_host_enqueue_s against learn_s_per_update, default and AMP, three runs each.   usage: bash tools/archive/gpu_r6_u.sh TAG
TAG=${1:-r6u}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3; do
  for t in aliengo aliengo_amp; do
    timeout 600 python bench.py --task $t --no-cpu-baseline < /dev/null > $O/bench_${t}_$i.log 2>&1
    timeout 20 python -c "import json; d=json.loads(open('$O/bench_${t}_$i.log').read().strip().splitlines()[-1]); print('$t run $i value %.3f M  collection %.5f  learn %.5f  enqueue %.5f  pci %s' % (d['value'] / 1e6, d['collection_s_per_iteration'], d['learn_s_per_update'], d['update_host_enqueue_s'], d['device']['pci']))" < /dev/null
  done
done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; uptime
