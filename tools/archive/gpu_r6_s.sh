#!/bin/bash
# round 6, lease s: wavefront-scope fences between the phases of kernels A / B as the product: the GPU suite, the default line, the A/B against the __syncthreads() build.   usage: bash tools/archive/gpu_r6_s.sh TAG
TAG=${1:-r6s}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q < /dev/null > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python bench.py --no-cpu-baseline < /dev/null > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-400
bash tools/gpu_ab_kernel_a.sh $TAG syncthreads=isaacgymloco_amd/csrc/variants/liblsim_syncthreads.so
