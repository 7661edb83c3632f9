#!/bin/bash
# quick GPU check: parity tests, env-only bench on two tasks, HBM traffic counters of kernel A.  usage: bash tools/gpu_quick.sh TAG
TAG=${1:-quick}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for t in aliengo aliengo_stairs; do timeout 300 python bench.py --task $t --mode env --steps 300 --warmup 50 --no-cpu-baseline 2>&1 | tail -1 > $O/env_$t.json; python -c "import json; d=json.load(open('$O/env_$t.json')); print('$t', round(d['value']), d['kernel_a_ms'], d['kernel_b_ms'])"; done
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$set -o pmc -- python3 bench.py --mode env --steps 20 --warmup 5 --no-cpu-baseline > $O/pmc_$set.log 2>&1
done
python tools/pmc_summary.py $O/pmc.csv --traffic $O/pmc_traffic.json --kernel lsim_k_step_a --task aliengo --envs 4096 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE | grep -v "^#"
find $O -name "*counter_collection.csv" -size +8M -delete
