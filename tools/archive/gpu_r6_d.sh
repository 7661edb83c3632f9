#!/bin/bash
# round 6, lease d: the whole GPU suite, smoke, the default line, the env-only line (kernel A with the non-finite ballot), the AMP line.   usage: bash tools/gpu_r6_d.sh TAG
TAG=${1:-r6d}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
b() { name=$1; shift; "$@" > $O/$name.log 2>&1; tail -1 $O/$name.log > $O/$name.json; }
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -8 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
b bench_default timeout 900 python bench.py
b bench_env timeout 600 python bench.py --mode env --steps 500 --warmup 50 --no-cpu-baseline
b bench_env_stairs timeout 600 python bench.py --mode env --task aliengo_stairs --steps 500 --warmup 50 --no-cpu-baseline
b bench_amp timeout 600 python bench.py --task aliengo_amp --steps 200 --warmup 100 --no-cpu-baseline
for f in bench_default bench_env bench_env_stairs bench_amp; do timeout 20 python -c "import json; d=json.load(open('$O/$f.json')); print('$f', round(d['value']), d.get('kernel_a_ms'), d.get('collection_s_per_iteration'), d.get('learn_s_per_update'), d.get('nonfinite_envs'))" < /dev/null; done
