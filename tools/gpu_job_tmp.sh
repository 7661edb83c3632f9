#!/bin/bash
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04w; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for t in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', round(d['value']), d.get('collection_s_per_iteration'), d.get('learn_s_per_update'))"; done
timeout 600 python bench.py --task aliengo_stairs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train stairs', round(d['value']), d.get('collection_s_per_iteration'), d.get('learn_s_per_update'))"
