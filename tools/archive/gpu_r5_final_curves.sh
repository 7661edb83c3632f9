#!/bin/bash
# round 5, final tree: the reference's full 1000-iteration Aliengo schedule, 300 iterations on the stairs task, and 300 Aliengo iterations with the learner's
# round-5 kernels switched back to BLAS + ELU (same seed) for comparison.   usage: bash tools/gpu_r5_final_curves.sh TAG
TAG=${1:-r5f}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python tools/train_probe.py 1000 $O/train_curve_aliengo_1000it.json aliengo 1 > $O/train_aliengo_1000.log 2>&1; tail -1 $O/train_aliengo_1000.log
timeout 900 python tools/train_probe.py 300 $O/train_curve_aliengo_stairs_300it.json aliengo_stairs 1 > $O/train_stairs.log 2>&1; tail -1 $O/train_stairs.log
LSIM_ELU_FORWARD=0 timeout 900 python tools/train_probe.py 300 $O/train_curve_aliengo_blas_elu_300it.json aliengo 1 > $O/train_aliengo_blas.log 2>&1; tail -1 $O/train_aliengo_blas.log
python3 - $O <<'PY'
import json, sys
O = sys.argv[1]
for f in ("train_curve_aliengo_1000it", "train_curve_aliengo_stairs_300it", "train_curve_aliengo_blas_elu_300it"):
    c = json.load(open(f"{O}/{f}.json"))["curve"]
    for it in (99, 299, 999):
        if it < len(c):
            r = c[it]
            print(f, "it", it, "ep_len %.0f ep_rew %.2f terrain %.2f est %.4f swap %.4f wall %.0f s" % (r["mean_ep_len"], r["mean_ep_reward"], r["terrain_level"], r["est_loss"], r["swap_loss"], r["wall_s"]))
PY
