#!/usr/bin/env python3
"""stdin: bench.py's JSON line -> one short line (A/B scripts).  usage: python bench.py ... | python tools/line_summary.py NAME"""
import json
import sys

j = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = j.get("sclk_during_timed_region") or {}
f = lambda v, p=5: "-" if v is None else f"{v:.{p}f}"
print(sys.argv[1] if len(sys.argv) > 1 else "", f"value {j['value'] / 1e6:.3f} M  collection {f(j.get('collection_s_per_iteration'))}  update {f(j.get('learn_s_per_update'))}  "
      f"kernel_a {f(j.get('kernel_a_ms'))}  sclk collection / update {f(s.get('mean_mhz_collection'), 0)} / {f(s.get('mean_mhz_update'), 0)} MHz  "
      f"gemm {f((j.get('gemm_probe_after_timed_region') or {}).get('tflops'), 1)} TF  streams {j.get('update_two_streams')}")
