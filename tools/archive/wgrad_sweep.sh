#!/bin/bash
# per-shape kernel durations of BLAS vs lsim_linear_wgrad (rocprofv3 kernel trace); usage: bash tools/wgrad_sweep.sh
export TMPDIR=/tmp
for shape in "128 12" "128 1" "64 19" "45 128" "64 16" "16 32" "128 64" "64 512" "256 128" "270 128" "238 512" "512 256"; do
  rm -rf /tmp/wgs; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wgs -o w -- python3 tools/wgrad_probe.py $shape tuned > /dev/null 2>&1
  python3 - "$shape" <<'PY'
import csv, sys
rows = list(csv.DictReader(open("/tmp/wgs/w_kernel_stats.csv")))
blas = sum(float(r["AverageNs"]) for r in rows if (r["Name"].startswith("Cijk") or "reduce_kernel" in r["Name"]) and int(r["Calls"]) >= 12)
mine = sum(float(r["AverageNs"]) * int(r["Calls"]) / 12 for r in rows if "lsim_k" in r["Name"])
print(f"{sys.argv[1]:>8s}  BLAS dW+db {blas / 1e3:7.1f} us   lsim_linear_wgrad (all kernels) {mine / 1e3:7.1f} us")
PY
done
