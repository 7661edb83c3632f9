#!/bin/bash
# round 5: whole GPU suite, then the update eager vs from HIP graphs on an idle and on a fully loaded host
TAG=${1:-r5c}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -8 $O/gpu_tests.log
run() { local name=$1; shift
  env "$@" timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; tail -2 $O/bench_$name.err | cut -c1-300; }
run graphs
run eager LSIM_UPDATE_GRAPH=0
python tools/cpu_hog.py 75 & HOG=$!
sleep 2
run graphs_loaded_host
run eager_loaded_host LSIM_UPDATE_GRAPH=0
run graphs_loaded_host2
run eager_loaded_host2 LSIM_UPDATE_GRAPH=0
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], {k:j.get(k) for k in ("value","collection_s_per_iteration","learn_s_per_update","iteration_spread_frac","update_two_streams","update_hip_graphs")}, (j.get("gemm_probe_after_timed_region") or {}).get("tflops"), (j.get("sclk_during_timed_region") or {}).get("mean_mhz"))
    except Exception as e: print(f, "failed", e)
PY
