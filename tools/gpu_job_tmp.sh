export TMPDIR=/tmp
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_physics_invariants.py tests/test_gpu_physics_anchors.py tests/test_contact_cap.py -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
run() { name=$1; shift; "$@" > $O/$name.log 2>&1; tail -1 $O/$name.log > $O/$name.json; }
for t in aliengo aliengo_stairs; do
  for n in 4096 65536; do
    run env_${t}_$n timeout 300 python bench.py --mode env --task $t --envs $n --steps 200 --warmup 50 --no-cpu-baseline
  done
done
run train timeout 600 python bench.py --no-cpu-baseline
run train_stairs timeout 600 python bench.py --task aliengo_stairs --no-cpu-baseline
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.load(open(f))
        print(os.path.basename(f), {k:j.get(k) for k in ("value","kernel_a_ms","ms_per_step","collection_s_per_iteration","learn_s_per_update")})
    except Exception as e: print(f, "failed", e)
PY
