#!/bin/bash
# round 6, lease x: does the size of torch's CPU thread pool (256 visible hardware threads, 16-CPU quota) matter to the N = 1 line?  OMP_NUM_THREADS unset against 1, interleaved.   usage: bash tools/archive/gpu_r6_x.sh TAG
TAG=${1:-r6x}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3; do
  for omp in unset 1; do
    for t in aliengo aliengo_amp; do
      if [ $omp = unset ]; then unset OMP_NUM_THREADS; else export OMP_NUM_THREADS=$omp; fi
      timeout 600 python bench.py --task $t --no-cpu-baseline < /dev/null > $O/bench_${t}_omp${omp}_$i.log 2>&1
      timeout 20 python -c "import json; d=json.loads(open('$O/bench_${t}_omp${omp}_$i.log').read().strip().splitlines()[-1]); print('$t omp=$omp run $i value %.3f M  collection %.5f  learn %.5f  enqueue %.5f' % (d['value'] / 1e6, d['collection_s_per_iteration'], d['learn_s_per_update'], d['update_host_enqueue_s']))" < /dev/null
    done
  done
done
unset OMP_NUM_THREADS
timeout 600 python -m pytest tests/test_gpu_amp_update.py -q -k index_upload < /dev/null 2>&1 | tail -2
