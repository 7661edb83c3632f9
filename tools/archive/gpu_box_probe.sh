#!/bin/bash
TAG=${1:-box}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/box_probe.py > $O/box_probe.json 2> $O/box_probe.err; python - <<PY
import json
j=json.load(open("$O/box_probe.json"))
print({k:v for k,v in j.items() if k not in ("rocm_smi","kernel_a_by_envs")})
print(j["kernel_a_by_envs"])
print(j["rocm_smi"][-1500:])
PY
