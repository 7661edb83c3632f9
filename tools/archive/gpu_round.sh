#!/bin/bash
# One gpurun call: GPU tests, default bench, env-only bench, rocprofv3 kernel traces and PMC passes.  Outputs under gpurun_out/$TAG.
# usage (from the repo root, on the GPU box):  bash tools/gpu_round.sh r01b
TAG=${1:-run}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
timeout 600 python bench.py --mode env --steps 500 --warmup 50 > $O/bench_env.log 2>&1; tail -1 $O/bench_env.log > $O/bench_env.json
timeout 900 python bench.py --mode env --envs 262144 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_env_N262144.log 2>&1; tail -1 $O/bench_env_N262144.log > $O/bench_env_N262144.json
timeout 300 python bench.py --mode env --actions zeros --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_env_zero_actions.log 2>&1; tail -1 $O/bench_env_zero_actions.log > $O/bench_env_zero_actions.json
timeout 300 python bench.py --mode env --envs 64 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_env_N64.log 2>&1; tail -1 $O/bench_env_N64.log > $O/bench_env_N64.json
for t in aliengo_stairs aliengo_amp; do timeout 900 python bench.py --task $t --no-cpu-baseline > $O/bench_$t.log 2>&1; tail -1 $O/bench_$t.log > $O/bench_$t.json; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_env -o env -- python3 bench.py --mode env --steps 100 --warmup 20 --no-cpu-baseline > $O/prof_env.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 bench.py --no-cpu-baseline > $O/prof_train.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$name -o pmc -- python3 bench.py --mode env --steps 20 --warmup 5 --no-cpu-baseline > $O/pmc_$name.log 2>&1
done
python tools/pmc_summary.py $O/pmc_env_N4096.csv --traffic $O/pmc_traffic.json --kernel lsim_k_step_a --task aliengo --envs 4096 \
  --note "rocprofv3 --pmc (4 separate passes), bench.py --mode env --steps 20 --warmup 5, aliengo N=4096; FETCH_SIZE/WRITE_SIZE in KB" \
  $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_WAVES $O/pmc_SQ_WAIT_ANY > /dev/null 2>$O/pmc_summary.err
find $O -name "*kernel_stats.csv" | head; cat $O/bench_default.json; cat $O/bench_env.json
# keep the merge-back small
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
