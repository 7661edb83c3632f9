#!/bin/bash
# round 6, lease o: is kernel A (93 KB of code, 64 KB instruction cache per CU pair) waiting on instruction fetch?  PMC passes of the env-only command.   usage: bash tools/archive/gpu_r6_o.sh TAG
TAG=${1:-r6o}; R=$(pwd); O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $R/$O/pmc_env_$name -o pmc -- python3 $R/bench.py --mode env --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $R/$O/pmc_env_$name.log 2>&1
  tail -1 $R/$O/pmc_env_$name.log | cut -c1-200
done
cd $R
python - <<'P' $O
import csv, glob, sys, collections
O = sys.argv[1]
for f in sorted(glob.glob(O + "/pmc_env_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("lsim_k_step_a"):
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in acc.items():
        print(f"{k:32s} {v / max(n, 1):16.1f} per launch ({n} launches)")
P
