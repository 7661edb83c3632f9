#!/bin/bash
export TMPDIR=/tmp
ulimit -c 0
line() { timeout 300 python bench.py --mode env "$@" --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'A', round(d['kernel_a_ms'],4), 'B', round(d['kernel_b_ms'],4))" 2>/dev/null || echo failed; }
for rep in 1 2 3; do
for fl in 0 256 512 768; do
  echo "flags $fl flat:   $(LSIM_STEP_FLAGS=$fl line)"
  echo "flags $fl stairs: $(LSIM_STEP_FLAGS=$fl line --task aliengo_stairs)"
done
done
