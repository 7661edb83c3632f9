"""GPU probe: forward + backward of one (Linear, ELU) pair through the fused backward (lsim_linear_elu_wgrad) at the minibatch size of the
reference configuration, host-timed.  usage: python tools/fz_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgymloco_amd.learn.fused_linear import _LinearEluFn
B = 102400
for k, n in ((64, 512), (256, 128), (512, 256), (238, 512)):
    x = torch.randn(B, k, device="cuda", requires_grad=True); g = torch.randn(B, n, device="cuda")
    w = (torch.randn(n, k, device="cuda") * 0.05).requires_grad_(True); bias = torch.zeros(n, device="cuda", requires_grad=True)
    for _ in range(3):
        zf = _LinearEluFn.apply(x, w, bias); zf.backward(g)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10):
        zf = _LinearEluFn.apply(x, w, bias); zf.backward(g)
    torch.cuda.synchronize(); print(k, n, (time.time() - t) / 10 * 1e6, "us per fwd+bwd", flush=True)
