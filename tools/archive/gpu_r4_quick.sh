#!/bin/bash
# round 4 iteration loop: GPU tests (optionally a -k subset), smoke, then env-only and train lines for BOTH solvers (LSIM_SOLVER override).
# usage: bash tools/gpu_r4_quick.sh TAG [pytest -k expr]
TAG=${1:-r4q}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
if [ -n "$2" ]; then K=(-k "$2"); else K=(); fi
timeout 1800 python -m pytest tests -m gpu -q "${K[@]}" > $O/gpu_tests.log 2>&1; tail -25 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for sv in tgs pgs; do
  LSIM_SOLVER=$sv timeout 600 python bench.py --mode env --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_env_$sv.log 2>&1; tail -1 $O/bench_env_$sv.log > $O/bench_env_$sv.json
  LSIM_SOLVER=$sv timeout 600 python bench.py --mode env --task aliengo_stairs --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_env_stairs_$sv.log 2>&1; tail -1 $O/bench_env_stairs_$sv.log > $O/bench_env_stairs_$sv.json
  LSIM_SOLVER=$sv timeout 900 python bench.py --no-cpu-baseline > $O/bench_train_$sv.log 2>&1; tail -1 $O/bench_train_$sv.log > $O/bench_train_$sv.json
done
python - <<PY
import json
for f in ("bench_env_tgs","bench_env_pgs","bench_env_stairs_tgs","bench_env_stairs_pgs","bench_train_tgs","bench_train_pgs"):
    try:
        j=json.load(open("$O/"+f+".json"))
        print(f, {k:j.get(k) for k in ("value","ms_per_step","kernel_a_ms","kernel_b_ms","collection_s_per_iteration","learn_s_per_update","iteration_wall_s_min_median_max")})
    except Exception as e: print(f, "failed", e)
PY
