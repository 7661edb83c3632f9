#!/bin/bash
# wave-cycle / MFMA-busy counters of a stand-alone micro-benchmark kernel (separate --pmc passes):  bash tools/micro/pmc_micro.sh "<kernel substring>" <binary> [args...]
export TMPDIR=/tmp
PAT=$1; shift
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  rm -rf /tmp/pmcm; (cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmcm -o w -- "$@" > /dev/null 2>&1)
  PAT="$PAT" python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob("/tmp/pmcm/**/*counter_collection.csv", recursive=True)
if not f: print("no counters"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if os.environ["PAT"] not in k: continue
    acc[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"]); n[k[:70]].add(r["Dispatch_Id"])
for k, d in acc.items():
    print(k, "dispatches", len(n[k]), {c: round(v / len(n[k])) for c, v in d.items()})
PY
done
