#!/bin/bash
# One gpurun call: GPU tests, smoke, benches (train default, driver arguments, env-only, sizes, tasks), rocprofv3 kernel traces and PMC passes.
# Outputs under gpurun_out/$TAG; tools/publish_profiles.py copies the judged summaries into profiles/.   usage: bash tools/gpu_round3.sh r03a
TAG=${1:-run}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args.log 2>&1; tail -1 $O/bench_driver_args.log > $O/bench_driver_args.json
timeout 600 python bench.py --mode env --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_env.log 2>&1; tail -1 $O/bench_env.log > $O/bench_env.json
timeout 300 python bench.py --mode env --envs 262144 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_env_N262144.log 2>&1; tail -1 $O/bench_env_N262144.log > $O/bench_env_N262144.json
timeout 300 python bench.py --mode env --actions zeros --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_env_zero_actions.log 2>&1; tail -1 $O/bench_env_zero_actions.log > $O/bench_env_zero_actions.json
timeout 300 python bench.py --mode env --envs 64 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_env_N64.log 2>&1; tail -1 $O/bench_env_N64.log > $O/bench_env_N64.json
for t in aliengo_stairs aliengo_amp go1; do timeout 400 python bench.py --task $t --no-cpu-baseline > $O/bench_$t.log 2>&1; tail -1 $O/bench_$t.log > $O/bench_$t.json; done
LSIM_DEBUG_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_2ranks_debug.log 2>&1; tail -1 $O/bench_2ranks_debug.log > $O/bench_2ranks_debug.json
LSIM_DEBUG_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --mixed-robots --no-cpu-baseline > $O/bench_2ranks_mixed_debug.log 2>&1; tail -1 $O/bench_2ranks_mixed_debug.log > $O/bench_2ranks_mixed_debug.json
# the RCCL calls of the N > 1 path on one GPU: a 1-rank group with every collective issued, started the way the driver starts N ranks
LSIM_DEBUG_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rccl_1rank.log 2>&1; tail -1 $O/bench_rccl_1rank.log > $O/bench_rccl_1rank.json
timeout 120 tools/micro/valu_peak > $O/valu_peak.json 2>/dev/null
timeout 120 tools/micro/delassus_mfma > $O/delassus_mfma.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_env -o env -- python3 bench.py --mode env --steps 100 --warmup 20 --no-cpu-baseline > $O/prof_env.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 bench.py --no-cpu-baseline > $O/prof_train.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_amp -o amp -- python3 bench.py --task aliengo_amp --steps 100 --warmup 100 --no-cpu-baseline > $O/prof_amp.log 2>&1
# PMC passes: the DEFAULT bench command (train mode) so that bench.py's roofline.traffic / valu_issue_frac match the driver's run, and env mode
for mode in train env; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE"; do
    name=$(echo $set | cut -d' ' -f1)
    if [ $mode = train ]; then extra="--steps 100 --warmup 100"; else extra="--mode env --steps 20 --warmup 5"; fi
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${mode}_$name -o pmc -- python3 bench.py $extra --no-cpu-baseline > $O/pmc_${mode}_$name.log 2>&1
  done
done
python tools/pmc_summary.py $O/pmc_N4096.csv --traffic $O/pmc_traffic.json --kernel lsim_k_step_a --task aliengo --envs 4096 \
  --run train:policy:$O/pmc_train_FETCH_SIZE,$O/pmc_train_WRITE_SIZE,$O/pmc_train_SQ_WAVES,$O/pmc_train_SQ_WAIT_ANY,$O/pmc_train_GRBM_GUI_ACTIVE \
  --run env:normal:$O/pmc_env_FETCH_SIZE,$O/pmc_env_WRITE_SIZE,$O/pmc_env_SQ_WAVES,$O/pmc_env_SQ_WAIT_ANY,$O/pmc_env_GRBM_GUI_ACTIVE > /dev/null 2>$O/pmc_summary.err
cat $O/bench_default.json; cat $O/bench_env.json
# keep the merge-back small
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
