export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_physics_invariants.py tests/test_gpu_physics_anchors.py -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
timeout 600 python tools/phase_profile.py aliengo 4096 > $O/phase_profile_aliengo.txt 2>&1; grep -v "^/opt" $O/phase_profile_aliengo.txt | awk '{ if ($6+0 > 2000 || NR==1) print }'
for sv in tgs pgs; do
  LSIM_SOLVER=$sv timeout 600 python bench.py --mode env --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_env_$sv.log 2>&1; tail -1 $O/bench_env_$sv.log > $O/bench_env_$sv.json
done
LSIM_SOLVER=tgs timeout 600 python bench.py --mode env --task aliengo_stairs --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_env_stairs_tgs.log 2>&1; tail -1 $O/bench_env_stairs_tgs.log > $O/bench_env_stairs_tgs.json
timeout 900 python bench.py --no-cpu-baseline > $O/bench_train_tgs.log 2>&1; tail -1 $O/bench_train_tgs.log > $O/bench_train_tgs.json
python - <<PY
import json
for f in ("bench_env_tgs","bench_env_pgs","bench_env_stairs_tgs","bench_train_tgs"):
    try:
        j=json.load(open("$O/"+f+".json"))
        print(f, {k:j.get(k) for k in ("value","ms_per_step","kernel_a_ms","kernel_b_ms","collection_s_per_iteration","learn_s_per_update")})
    except Exception as e: print(f, "failed", e)
PY
