#!/bin/bash
# round 6, lease z: lsim_amp_pair_rows (normalise + concatenate the sampled AMP pairs in one pass): GPU suite, AMP line x 3.   usage: bash tools/archive/gpu_r6_z.sh TAG
TAG=${1:-r6z}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q < /dev/null > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
for i in 1 2 3; do
  for t in aliengo_amp; do
    timeout 600 python bench.py --task $t --no-cpu-baseline < /dev/null > $O/bench_${t}_$i.log 2>&1
    timeout 20 python -c "import json; d=json.loads(open('$O/bench_${t}_$i.log').read().strip().splitlines()[-1]); print('$t run $i value %.3f M  collection %.5f  learn %.5f  enqueue %.5f' % (d['value'] / 1e6, d['collection_s_per_iteration'], d['learn_s_per_update'], d['update_host_enqueue_s']))" < /dev/null
  done
done
