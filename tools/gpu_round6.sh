#!/bin/bash
# One gpurun call (round 6): GPU tests, smoke, the bench lines (train default with the CPU baseline, driver arguments, env-only flat / stairs / large N,
# the other tasks, the multi-rank proxies), the fused kernels alone, rocprofv3 kernel statistics (env, stairs, train, AMP) and the PMC passes of the train,
# env and stairs workloads (kernel lsim_k_step_a_tgs).  EVERY command runs under its own `timeout` with stdin closed (round 6 lost an hour of lease to a
# summary script that waited on stdin).  Outputs under gpurun_out/$TAG; tools/publish_profiles.py TAG r06 copies the judged summaries into profiles/.
# usage: bash tools/gpu_round6.sh r06x
TAG=${1:-run}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
b() { name=$1; shift; "$@" < /dev/null > $O/$name.log 2>&1; tail -1 $O/$name.log > $O/$name.json; }
rm -f $O/free_running_parity.jsonl
LSIM_PARITY_REPORT=$O/free_running_parity.jsonl timeout 1500 python -m pytest tests -m gpu -q < /dev/null > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null > $O/smoke.log 2>&1; tail -1 $O/smoke.log
b bench_default timeout 900 python bench.py
b bench_driver_args timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
b bench_env timeout 600 python bench.py --mode env --steps 500 --warmup 50 --no-cpu-baseline
b bench_env_N262144 timeout 300 python bench.py --mode env --envs 262144 --steps 50 --warmup 10 --no-cpu-baseline
b bench_env_zero_actions timeout 300 python bench.py --mode env --actions zeros --steps 500 --warmup 50 --no-cpu-baseline
b bench_env_aliengo_stairs timeout 300 python bench.py --mode env --task aliengo_stairs --steps 500 --warmup 50 --no-cpu-baseline
for t in aliengo_stairs aliengo_amp go1 go2; do b bench_$t timeout 400 python bench.py --task $t --no-cpu-baseline; done
LSIM_AMP_FUSED_STEP=0 LSIM_AMP_FUSED_UPDATE=0 b bench_aliengo_amp_round5_path timeout 400 python bench.py --task aliengo_amp --no-cpu-baseline
LSIM_DEBUG_SINGLE_DEVICE=1 b bench_2ranks_debug timeout 600 python bench.py --gpus 2 --no-cpu-baseline
LSIM_DEBUG_SINGLE_DEVICE=1 b bench_2ranks_mixed_debug timeout 600 python bench.py --gpus 2 --mixed-robots --no-cpu-baseline
LSIM_DEBUG_FORCE_COLLECTIVES=1 b bench_rccl_1rank timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline
b bench_plain_again timeout 600 python bench.py --no-cpu-baseline
for i in 1 2 3; do timeout 120 python tools/policy_time.py < /dev/null 2>/dev/null; done > $O/policy_time.txt
for n in 4096 8192; do timeout 120 python tools/amp_step_time.py $n < /dev/null 2>/dev/null; done > $O/amp_step_time.txt
timeout 120 tools/micro/valu_peak < /dev/null > $O/valu_peak.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_env -o env -- python3 $R/bench.py --mode env --steps 100 --warmup 20 --no-cpu-baseline < /dev/null > $R/$O/prof_env.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_env_stairs -o env_stairs -- python3 $R/bench.py --mode env --task aliengo_stairs --steps 100 --warmup 20 --no-cpu-baseline < /dev/null > $R/$O/prof_env_stairs.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_train -o train -- python3 $R/bench.py --no-cpu-baseline < /dev/null > $R/$O/prof_train.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_amp -o amp -- python3 $R/bench.py --task aliengo_amp --steps 100 --warmup 100 --no-cpu-baseline < /dev/null > $R/$O/prof_amp.log 2>&1
# PMC passes (separate runs per counter set): the DEFAULT bench command (train mode) so that bench.py's roofline.traffic / valu_issue_frac match the
# driver's run, env mode, and env mode on the stairs task
for wl in train env stairs; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE"; do
    name=$(echo $set | cut -d' ' -f1)
    if [ $wl = train ]; then extra="--steps 100 --warmup 100"; elif [ $wl = env ]; then extra="--mode env --steps 20 --warmup 5"; else extra="--mode env --task aliengo_stairs --steps 20 --warmup 5"; fi
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $R/$O/pmc_${wl}_$name -o pmc -- python3 $R/bench.py $extra --no-cpu-baseline < /dev/null > $R/$O/pmc_${wl}_$name.log 2>&1
  done
done
cd $R
timeout 120 python tools/pmc_summary.py $O/pmc_N4096.csv --traffic $O/pmc_traffic.json --kernel lsim_k_step_a --task aliengo --envs 4096 --solver tgs \
  --run train:policy:$O/pmc_train_FETCH_SIZE,$O/pmc_train_WRITE_SIZE,$O/pmc_train_SQ_WAVES,$O/pmc_train_SQ_WAIT_ANY,$O/pmc_train_GRBM_GUI_ACTIVE \
  --run env:normal:$O/pmc_env_FETCH_SIZE,$O/pmc_env_WRITE_SIZE,$O/pmc_env_SQ_WAVES,$O/pmc_env_SQ_WAIT_ANY,$O/pmc_env_GRBM_GUI_ACTIVE < /dev/null > /dev/null 2>$O/pmc_summary.err
timeout 120 python tools/pmc_summary.py $O/pmc_stairs_N4096.csv --traffic $O/pmc_traffic.json --kernel lsim_k_step_a --task aliengo_stairs --envs 4096 --solver tgs --append 1 \
  --run env:normal:$O/pmc_stairs_FETCH_SIZE,$O/pmc_stairs_WRITE_SIZE,$O/pmc_stairs_SQ_WAVES,$O/pmc_stairs_SQ_WAIT_ANY,$O/pmc_stairs_GRBM_GUI_ACTIVE < /dev/null > /dev/null 2>>$O/pmc_summary.err
cat $O/bench_default.json | cut -c1-300; cat $O/bench_env.json | cut -c1-200
# keep the merge-back small
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
