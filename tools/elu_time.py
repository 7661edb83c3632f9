#!/usr/bin/env python3
"""ELU forward of the learner's hidden activations (102 400 rows): torch out of place / in place against a plain copy, CUDA events.
usage: python tools/elu_time.py"""
import torch
import torch.nn.functional as F


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (512, 256, 128, 64):
    y = torch.randn(102400, n, device="cuda:0")
    o = torch.empty_like(y)
    mb = y.numel() * 8 / 1e6
    a, b, c = t(lambda: F.elu(y)), t(lambda: F.elu_(y)), t(lambda: o.copy_(y))
    print(f"[102400 x {n}] ({mb:.0f} MB moved): elu {a:.1f} us ({mb / a:.2f} TB/s)  elu_ {b:.1f} us ({mb / b:.2f} TB/s)  copy {c:.1f} us ({mb / c:.2f} TB/s)")
