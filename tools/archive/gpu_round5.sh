#!/bin/bash
# One gpurun call (round 5): GPU tests, smoke, benches (train default, driver arguments, env-only for both solvers and both kernel forms, sizes,
# tasks), the multi-rank proxies, rocprofv3 kernel traces and PMC passes for the flat AND the stairs task (kernel lsim_k_step_a_tgs).
# Outputs under gpurun_out/$TAG; tools/publish_profiles.py copies the judged summaries into profiles/.   usage: bash tools/gpu_round5.sh r04x
TAG=${1:-run}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
b() { name=$1; shift; "$@" > $O/$name.log 2>&1; tail -1 $O/$name.log > $O/$name.json; }
rm -f $O/free_running_parity.jsonl
LSIM_PARITY_REPORT=$O/free_running_parity.jsonl timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
b bench_default timeout 900 python bench.py
b bench_driver_args timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
b bench_env timeout 600 python bench.py --mode env --steps 500 --warmup 50 --no-cpu-baseline
LSIM_SOLVER=pgs b bench_env_pgs timeout 600 python bench.py --mode env --steps 500 --warmup 50 --no-cpu-baseline
LSIM_SOLVER=pgs b bench_default_pgs timeout 600 python bench.py --no-cpu-baseline
LSIM_LIN_VEL=origin LSIM_TGS_LIMIT_PASSES=0 b bench_env_r4_conventions timeout 600 python bench.py --mode env --steps 500 --warmup 50 --no-cpu-baseline
b bench_env_N262144 timeout 300 python bench.py --mode env --envs 262144 --steps 50 --warmup 10 --no-cpu-baseline
b bench_env_zero_actions timeout 300 python bench.py --mode env --actions zeros --steps 500 --warmup 50 --no-cpu-baseline
b bench_env_N64 timeout 300 python bench.py --mode env --envs 64 --steps 500 --warmup 50 --no-cpu-baseline
b bench_env_aliengo_stairs timeout 300 python bench.py --mode env --task aliengo_stairs --steps 500 --warmup 50 --no-cpu-baseline
LSIM_SOLVER=pgs b bench_env_aliengo_stairs_pgs timeout 300 python bench.py --mode env --task aliengo_stairs --steps 500 --warmup 50 --no-cpu-baseline
for t in aliengo_stairs aliengo_amp go1 go2; do b bench_$t timeout 400 python bench.py --task $t --no-cpu-baseline; done
LSIM_DEBUG_SINGLE_DEVICE=1 b bench_2ranks_debug timeout 600 python bench.py --gpus 2 --no-cpu-baseline
LSIM_DEBUG_SINGLE_DEVICE=1 b bench_2ranks_mixed_debug timeout 600 python bench.py --gpus 2 --mixed-robots --no-cpu-baseline
# the RCCL calls of the N > 1 path on one GPU: a 1-rank group with every collective issued, started the way the driver starts N ranks
LSIM_DEBUG_FORCE_COLLECTIVES=1 b bench_rccl_1rank timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline
GPU_MAX_HW_QUEUES=4 LSIM_DEBUG_FORCE_COLLECTIVES=1 b bench_rccl_1rank_4queues timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --no-cpu-baseline
b bench_plain_again timeout 600 python bench.py --no-cpu-baseline
b bench_driver_args_again timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
for i in 1 2 3; do timeout 120 python tools/policy_time.py 2>/dev/null; done > $O/policy_time.txt
timeout 120 tools/micro/valu_peak > $O/valu_peak.json 2>/dev/null
timeout 300 python tools/phase_profile.py aliengo 4096 > $O/phase_profile_aliengo.txt 2>&1
timeout 300 python tools/phase_profile.py aliengo_stairs 4096 > $O/phase_profile_aliengo_stairs.txt 2>&1
timeout 300 python tools/wave_times.py aliengo 4096 > $O/wave_times_aliengo.txt 2>&1
timeout 300 python tools/wave_times.py aliengo_stairs 4096 > $O/wave_times_aliengo_stairs.txt 2>&1
timeout 300 python tools/wave_times.py aliengo 256 > $O/wave_times_aliengo_N256.txt 2>&1
timeout 300 python tools/wave_times.py aliengo 4096 --phases > $O/wave_phases_aliengo.txt 2>&1
timeout 300 python tools/wave_times.py aliengo 256 --phases > $O/wave_phases_aliengo_N256.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_env -o env -- python3 $R/bench.py --mode env --steps 100 --warmup 20 --no-cpu-baseline > $R/$O/prof_env.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_env_stairs -o env_stairs -- python3 $R/bench.py --mode env --task aliengo_stairs --steps 100 --warmup 20 --no-cpu-baseline > $R/$O/prof_env_stairs.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_train -o train -- python3 $R/bench.py --no-cpu-baseline > $R/$O/prof_train.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_amp -o amp -- python3 $R/bench.py --task aliengo_amp --steps 100 --warmup 100 --no-cpu-baseline > $R/$O/prof_amp.log 2>&1
LSIM_DEBUG_FORCE_COLLECTIVES=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof_rccl -o rccl -- python3 $R/bench.py --gpus 1 --steps 100 --warmup 100 --no-cpu-baseline > $R/$O/prof_rccl.log 2>&1
# PMC passes (separate runs per counter set): the DEFAULT bench command (train mode) so that bench.py's roofline.traffic / valu_issue_frac match the
# driver's run, env mode, and env mode on the stairs task
for wl in train env stairs; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE"; do
    name=$(echo $set | cut -d' ' -f1)
    if [ $wl = train ]; then extra="--steps 100 --warmup 100"; elif [ $wl = env ]; then extra="--mode env --steps 20 --warmup 5"; else extra="--mode env --task aliengo_stairs --steps 20 --warmup 5"; fi
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $R/$O/pmc_${wl}_$name -o pmc -- python3 $R/bench.py $extra --no-cpu-baseline > $R/$O/pmc_${wl}_$name.log 2>&1
  done
done
cd $R
python tools/pmc_summary.py $O/pmc_N4096.csv --traffic $O/pmc_traffic.json --kernel lsim_k_step_a --task aliengo --envs 4096 --solver tgs \
  --run train:policy:$O/pmc_train_FETCH_SIZE,$O/pmc_train_WRITE_SIZE,$O/pmc_train_SQ_WAVES,$O/pmc_train_SQ_WAIT_ANY,$O/pmc_train_GRBM_GUI_ACTIVE \
  --run env:normal:$O/pmc_env_FETCH_SIZE,$O/pmc_env_WRITE_SIZE,$O/pmc_env_SQ_WAVES,$O/pmc_env_SQ_WAIT_ANY,$O/pmc_env_GRBM_GUI_ACTIVE > /dev/null 2>$O/pmc_summary.err
python tools/pmc_summary.py $O/pmc_stairs_N4096.csv --traffic $O/pmc_traffic.json --kernel lsim_k_step_a --task aliengo_stairs --envs 4096 --solver tgs --append 1 \
  --run env:normal:$O/pmc_stairs_FETCH_SIZE,$O/pmc_stairs_WRITE_SIZE,$O/pmc_stairs_SQ_WAVES,$O/pmc_stairs_SQ_WAIT_ANY,$O/pmc_stairs_GRBM_GUI_ACTIVE > /dev/null 2>>$O/pmc_summary.err
f=$(find $O/prof_rccl -name "*kernel_trace.csv" | head -1); python tools/trace_idle.py $f 12 > $O/trace_idle_rccl.txt 2>&1
cat $O/bench_default.json; cat $O/bench_env.json
# keep the merge-back small
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
