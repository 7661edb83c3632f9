"""lsim_relu_backward_bias against aten's threshold_backward + sum on the AMP discriminator's shapes (MI355X).   python tools/relu_bwd_probe.py"""
import json
import sys

import torch

sys.path.insert(0, ".")
from isaacgymloco_amd.learn.fused_linear import relu_backward_bias_hip  # noqa: E402


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


out = []
for n in (1024, 512):
    act = torch.relu(torch.randn(102400, n, device="cuda:0"))
    g = torch.randn(102400, n, device="cuda:0")
    t_aten = timed(lambda: torch.ops.aten.threshold_backward(g, act, 0.0).sum(0))
    t_fused = timed(lambda: relu_backward_bias_hip(act, g))
    t_sum_only = timed(lambda: relu_backward_bias_hip(act, g, want_grad=False))
    out.append({"rows": 102400, "n": n, "aten_threshold_plus_sum_us": round(t_aten, 1), "lsim_us": round(t_fused, 1), "lsim_sums_only_us": round(t_sum_only, 1),
                "lsim_TBps": round(3 * act.numel() * 4 / t_fused / 1e6, 2)})
    print(out[-1], flush=True)
print(json.dumps(out))
