#!/bin/bash
# round 5, first lease: sysfs clock source, the new learner tests, then the train line with the update from HIP graphs / eager / hardware-queue settings
# usage: bash tools/gpu_r5_a.sh TAG
TAG=${1:-r5a}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
{ ls /sys/class/drm/; for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo "== $f"; cat $f; done; nproc; } > $O/sysfs.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_learner.py -m gpu -q -x -k "hip_graphs or replaced_optimiser or applied_twice or device_lr or checkpoint" > $O/gpu_tests_new.log 2>&1; tail -15 $O/gpu_tests_new.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; tail -3 $O/bench_$name.err
}
run graphs
run eager LSIM_UPDATE_GRAPH=0




run graphs_1stream LSIM_UPDATE_STREAMS=0
run graphs_again

python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], {k:j.get(k) for k in ("value","collection_s_per_iteration","learn_s_per_update","iteration_spread_frac","update_two_streams","update_hip_graphs","gpu_max_hw_queues")}, (j.get("gemm_probe_after_timed_region") or {}).get("tflops"), j.get("sclk_during_timed_region"), {k:(j.get("tunableop") or {}).get(k) for k in ("entries_loaded","explicit_solutions_loaded","validators_match")})
    except Exception as e: print(f, "failed", e)
PY
