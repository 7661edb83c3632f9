#!/bin/bash
bash tools/gpu_ab_kernel_a.sh r6f before=tests/_build/variants/liblsim_before.so i6_38=tests/_build/variants/liblsim_i6_38.so
LSIM_LIB=$PWD/tests/_build/variants/liblsim_i6_38.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_free_running.py tests/test_nonfinite_counter.py -m gpu -q -x < /dev/null 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_physics_invariants.py tests/test_nonfinite_counter.py -m gpu -q -x < /dev/null 2>&1 | tail -3
timeout 300 python oracle/cpu_bench.py --seconds 20 < /dev/null 2>&1 | tail -1 | cut -c1-900
