#!/bin/bash
# round 5: whole GPU suite + the default train line (driver's arguments)
TAG=${1:-r5d}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q ${2:+-k "$2"} > $O/gpu_tests.log 2>&1; tail -8 $O/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; tail -2 $O/bench_driver_args.err | cut -c1-300
python - <<PY
import json
j=json.loads(open("$O/bench_driver_args.json").read().strip().splitlines()[-1])
print({k:j.get(k) for k in ("value","collection_s_per_iteration","learn_s_per_update","iteration_spread_frac","update_two_streams","gpu_max_hw_queues","kernel_a_ms")}, j.get("gemm_probe_after_timed_region"), j.get("sclk_during_timed_region"), j.get("tunableop"), j.get("cpu_baseline"))
PY
