#!/bin/bash
# PMC counters of lsim_k_linear_fwd on one layer shape.  usage: bash tools/gpu_fwd_pmc.sh TAG "hidden 2"
TAG=${1:-fwdpmc}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp; export FWD_ONLY="${2:-hidden 2}"
cd /tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  rm -rf /tmp/fp; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/fp -o p -- python3 $GRAFT_REPO_ROOT/tools/fwd_time.py > /dev/null 2>&1
  python3 - "$set" <<'PY'
import csv, sys, glob
f = glob.glob("/tmp/fp/**/p_counter_collection.csv", recursive=True)
if not f: print("no counters for", sys.argv[1]); sys.exit()
acc = {}
n = {}
for r in csv.DictReader(open(f[0])):
    if "linear_fwd" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0) + float(r["Counter_Value"]); n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
for k in acc: print(f"{k:36s} {acc[k] / n[k]:16.0f} per launch ({n[k]} launches)")
PY
done | tee $O/pmc.txt
