#!/bin/bash
# round-2 kernel-A investigation: VALU micro-benchmark, per-phase ticks, size sweep, PMC passes.  usage: bash tools/gpu_r2b.sh TAG
TAG=${1:-r02b}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 120 tools/micro/valu_peak > $O/valu_peak.json 2>$O/valu_peak.err; cat $O/valu_peak.json
timeout 600 python tools/phase_profile.py aliengo 4096 > $O/phase_profile_N4096.txt 2>&1; cat $O/phase_profile_N4096.txt
timeout 600 python tools/phase_profile.py aliengo 1024 > $O/phase_profile_N1024.txt 2>&1; head -3 $O/phase_profile_N1024.txt
for n in 64 256 1024 2048 4096 8192 16384; do
  timeout 300 python bench.py --mode env --envs $n --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > $O/env_N$n.json
  python -c "import json; d=json.load(open('$O/env_N$n.json')); print('N', $n, 'value', round(d['value']), 'A ms', round(d['kernel_a_ms'],4), 'B ms', round(d['kernel_b_ms'],4))"
done
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_IFETCH" \
           "SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$name -o pmc -- python3 bench.py --mode env --steps 20 --warmup 5 --no-cpu-baseline > $O/pmc_$name.log 2>&1
  tail -2 $O/pmc_$name.log | cut -c1-300
done
python tools/pmc_summary.py $O/pmc_env_N4096.csv $O/pmc_SQ_WAVES $O/pmc_SQ_WAIT_ANY $O/pmc_SQ_IFETCH_LEVEL $O/pmc_GRBM_GUI_ACTIVE
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
