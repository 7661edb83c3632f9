#!/bin/bash
# round 6, lease k: the 32-row MFMA Delassus build: physics tests, then kernel A against the build without it (interleaved).   usage: bash tools/archive/gpu_r6_k.sh TAG
TAG=${1:-r6k}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_free_running.py tests/test_gpu_physics_anchors.py tests/test_physics_invariants.py tests/test_contact_cap.py tests/test_gpu_wall_contacts_forms.py tests/test_nonfinite_counter.py -m gpu -q -x < /dev/null > $O/tests.log 2>&1; tail -4 $O/tests.log
bash tools/gpu_ab_kernel_a.sh $TAG nomfma32=tests/_build/variants/liblsim_nomfma32.so
