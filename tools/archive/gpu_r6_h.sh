#!/bin/bash
# round 6, lease h: padded shuffle rows -- learner tests, then the train line with LSIM_PAD_SHUFFLED = 1 / 0 interleaved.   usage: bash tools/archive/gpu_r6_h.sh TAG
TAG=${1:-r6h}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_learner.py tests/test_gpu_learner_golden.py tests/test_gpu_amp_update.py -m gpu -q -x < /dev/null > $O/tests.log 2>&1; tail -12 $O/tests.log
for round in 1 2 3; do for v in 1 0; do
  LSIM_PAD_SHUFFLED=$v timeout 600 python bench.py --no-cpu-baseline < /dev/null > $O/train_pad${v}_$round.log 2>&1
  timeout 20 python -c "import json; d=json.loads(open('$O/train_pad${v}_$round.log').read().strip().splitlines()[-1]); print('pad=$v round $round value %.3f M  coll %.5f  update %.5f  median it %.5f  tunable %s' % (d['value']/1e6, d['collection_s_per_iteration'], d['learn_s_per_update'], d['iteration_wall_s_min_median_max'][1], d['update_two_streams']))" < /dev/null
done; done
