#!/bin/bash
# round 3 iteration loop: GPU tests, smoke, the default train line and the env-only line.   usage: bash tools/gpu_r3_quick.sh TAG [pytest -k expr]
TAG=${1:-r3q}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
if [ -n "$2" ]; then K=(-k "$2"); else K=(); fi
timeout 1800 python -m pytest tests -m gpu -q "${K[@]}" > $O/gpu_tests.log 2>&1; tail -15 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
timeout 600 python bench.py --mode env --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_env.log 2>&1; tail -1 $O/bench_env.log > $O/bench_env.json
python - <<PY
import json
for f in ("bench_default","bench_env"):
    try:
        j=json.load(open("$O/"+f+".json"))
        print(f, {k:j.get(k) for k in ("value","ms_per_step","kernel_a_ms","kernel_b_ms","collection_s_per_iteration","learn_s_per_update","iteration_wall_s_min_median_max")})
    except Exception as e: print(f, "failed", e)
PY
