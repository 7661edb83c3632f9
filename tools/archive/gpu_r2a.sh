#!/bin/bash
# round-2 first GPU check: new bench.py (whole iterations, --gpus launcher), the GPU suite.  usage: bash tools/gpu_r2a.sh TAG
TAG=${1:-r02a}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.log 2>&1; tail -1 $O/bench_driver_args.log > $O/bench_driver_args.json; cat $O/bench_driver_args.json
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json; cat $O/bench_default.json
LSIM_DEBUG_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_2ranks_debug.log 2>&1; tail -1 $O/bench_2ranks_debug.log > $O/bench_2ranks_debug.json; cat $O/bench_2ranks_debug.json
