#!/usr/bin/env python3
"""lsim_estimator_loss alone (the estimator's loss head of HIMEstimator.update: ten launches) at the BASELINE minibatch, CUDA events.
usage: python tools/est_loss_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from isaacgymloco_amd.learn import fused_linear as FL  # noqa: E402

B, D, K = 102400, 16, 32
g = torch.Generator(device="cuda:0").manual_seed(0)
enc = torch.randn(B, 3 + D, device="cuda:0", generator=g)
tgt = torch.randn(B, D, device="cuda:0", generator=g)
proto = torch.nn.functional.normalize(torch.randn(K, D, device="cuda:0", generator=g), dim=-1)
vel = torch.randn(B, 3, device="cuda:0", generator=g)


def call():
    return FL.estimator_loss_hip(enc, tgt, proto, vel, 3.0)


for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    call()
e1.record(); torch.cuda.synchronize()
total, parts = call()
print(f"estimator loss head, B = {B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (losses {float(parts[0]):.5f} {float(parts[1]):.5f})")
