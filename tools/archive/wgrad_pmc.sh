#!/bin/bash
# wave-cycle accounting of the tiled weight-gradient kernel for one shape (separate --pmc passes); usage: bash tools/wgrad_pmc.sh K_IN N_OUT
export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/wgp; timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/wgp -o w -- python3 $(dirname $0)/wgrad_probe.py $1 $2 tuned $3 > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/wgp/**/*counter_collection.csv", recursive=True)
if not f: print("no counters"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "lsim_k_linear_wgrad" not in k: continue
    acc[k[:44]][r["Counter_Name"]] += float(r["Counter_Value"]); n[k[:44]].add(r["Dispatch_Id"])
for k, d in acc.items():
    print(k, "dispatches", len(n[k]), {c: round(v / len(n[k])) for c, v in d.items()})
PY
done
