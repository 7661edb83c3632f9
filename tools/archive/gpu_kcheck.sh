#!/bin/bash
# kernel-A iteration loop: physics / parity tests on the GPU, then the env-only bench (kernel A / B durations).  usage: bash tools/gpu_kcheck.sh TAG
TAG=${1:-k}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for t in aliengo aliengo_stairs; do timeout 300 python bench.py --task $t --mode env --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > $O/env_$t.json; python -c "import json; d=json.load(open('$O/env_$t.json')); print('$t', round(d['value']), 'A', round(d['kernel_a_ms'],4), 'B', round(d['kernel_b_ms'],4))"; done
timeout 300 python bench.py --mode env --actions zeros --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > $O/env_zero.json; python -c "import json; d=json.load(open('$O/env_zero.json')); print('zeros', round(d['value']), 'A', round(d['kernel_a_ms'],4), 'B', round(d['kernel_b_ms'],4))"
