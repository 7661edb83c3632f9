"""GPU probe: lsim_linear_wgrad vs BLAS (g.t() @ x, g.sum(0)) timing for the learner's narrow layers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymloco_amd.learn.fused_linear import linear_wgrad
B = 102400
def bench(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for k, n in [(128, 12), (128, 1), (64, 19), (45, 128), (64, 16), (128, 64), (16, 32)]:
    x = torch.randn(B, k, device="cuda"); g = torch.randn(B, n, device="cuda")
    t_blas = bench(lambda: (g.t() @ x, g.sum(0)))
    t_hip = bench(lambda: linear_wgrad(x, g))
    print(f"{k:4d}->{n:4d}  BLAS dW+db {t_blas * 1e6:7.1f} us   lsim_linear_wgrad {t_hip * 1e6:7.1f} us")
