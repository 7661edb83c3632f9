#!/usr/bin/env python3
"""lsim_gather_rows alone: the ten fields of HIMRolloutStorage's once-per-update shuffle at the BASELINE size (409 600 rows), CUDA events.
usage: python tools/gather_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from isaacgymloco_amd.learn.storage import _gather_rows  # noqa: E402

n = 409600
perm = torch.randperm(n, device="cuda:0")
tot = 0.0
for cols in (270, 238, 238, 12, 1, 1, 1, 1, 12, 12):
    f = torch.randn(n, cols, device="cuda:0")
    out = torch.empty_like(f)
    for _ in range(3):
        _gather_rows(f, perm, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _gather_rows(f, perm, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    print(f"cols {cols:4d}: {us:7.1f} us  ({2 * n * cols * 4 / us / 1e6:5.2f} TB/s read + write)")
print(f"ten fields: {tot:.0f} us")
