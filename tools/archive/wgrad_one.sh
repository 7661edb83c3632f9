#!/bin/bash
# kernel durations of BLAS vs lsim_linear_wgrad for one shape; usage: bash tools/wgrad_one.sh K_IN N_OUT
export TMPDIR=/tmp
rm -rf /tmp/wgs; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wgs -o w -- python3 tools/wgrad_probe.py $1 $2 tuned $3 > /dev/null 2>&1
python3 - <<'PY'
import csv
for r in sorted(csv.DictReader(open("/tmp/wgs/w_kernel_stats.csv")), key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print(f'{float(r["AverageNs"]) / 1e3:8.1f} us x {r["Calls"]:>4s}  {r["Name"][:110]}')
PY
