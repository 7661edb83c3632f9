#!/usr/bin/env python3
"""What kind of box is this?  Device properties, shader clock, an fp32 GEMM probe, and kernel A's time at batch sizes around the one-round
limit (a launch of N robots is N waves of 10 KB of LDS: 16 per CU, 4096 on 256 CUs -- one CU or 640 bytes of LDS less and N = 4096 needs a
second round of waves).  usage: python tools/box_probe.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

p = torch.cuda.get_device_properties(0)
out = {"name": p.name, "multi_processor_count": p.multi_processor_count, "total_memory_gb": round(p.total_memory / 2**30, 1),
       "gcn_arch": getattr(p, "gcnArchName", None), "pci": f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0",
       "clock_rate_khz": getattr(p, "clock_rate", None), "max_threads_per_mp": getattr(p, "max_threads_per_multi_processor", None),
       "shared_memory_per_mp": getattr(p, "shared_memory_per_multiprocessor", None), "regs_per_mp": getattr(p, "regs_per_multiprocessor", None)}
try:
    out["rocm_smi"] = subprocess.run(["rocm-smi", "--showperflevel", "--showpower", "--showclocks", "--showcomputepartition", "--showmemorypartition"],
                                     capture_output=True, text=True, timeout=60).stdout[-3000:]
except Exception as e:
    out["rocm_smi"] = f"{type(e).__name__}: {e}"
sizes = {}
for n in (3584, 3840, 4032, 4096, 4160, 8192):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "env", "--envs", str(n), "--steps", "300", "--warmup", "50", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        sizes[n] = {"kernel_a_ms": j["kernel_a_ms"], "value": j["value"]}
    except Exception as e:
        sizes[n] = f"failed: {e}: {r.stderr[-300:]}"
out["kernel_a_by_envs"] = sizes
print(json.dumps(out, indent=1))
