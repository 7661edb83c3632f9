#!/bin/bash
TAG=${1:-est}; shift; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_learner.py tests/test_gpu_learner_golden.py -m gpu -q -k "estimator or sinkhorn or learner_golden or himppo" 2>&1 | tail -2
for i in 1 2 3; do python tools/est_loss_time.py; for v in "$@"; do echo -n "$v: "; LSIM_LIB=$PWD/isaacgymloco_amd/csrc/variants/liblsim_$v.so python tools/est_loss_time.py; done; done | tee $O/est_loss_time.txt
cd /tmp; rm -rf /tmp/estp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/estp -o e -- python3 $GRAFT_REPO_ROOT/tools/est_loss_time.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee $O/est_kernels.txt
import csv
for r in sorted(csv.DictReader(open("/tmp/estp/e_kernel_stats.csv")), key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f'{float(r["AverageNs"]) / 1e3:8.1f} us x {r["Calls"]:>4s}  {r["Name"][:100]}')
PY
