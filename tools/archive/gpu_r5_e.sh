#!/bin/bash
# round 5: whole GPU suite with the free-running parity tables, env-only lines (flat / stairs) with the two round-5 switches off for comparison, the train line
TAG=${1:-r5f}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
rm -f $O/free_running_parity.jsonl
LSIM_PARITY_REPORT=$O/free_running_parity.jsonl timeout 2400 python -m pytest tests/test_gpu_free_running.py -m gpu -q -s > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log | cut -c1-300
grep -h "joint speeds beyond\|agreement\|base velocity vs\|fastest joint" $O/gpu_tests.log | cut -c1-400
run() { local name=$1; shift; local args=$1; shift
  env "$@" timeout 900 python bench.py $args --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; tail -1 $O/bench_$name.err | cut -c1-200; }
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], {k:j.get(k) for k in ("value","kernel_a_ms","collection_s_per_iteration","learn_s_per_update")})
    except Exception as e: print(f, "failed", e)
PY
