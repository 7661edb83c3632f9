"""GPU probe: lsim_linear_wgrad vs BLAS (g.t() @ x, g.sum(0)) for one layer shape; run under rocprofv3 --kernel-trace and read the
kernel durations (host timing is launch-bound for the small shapes).  usage: python3 tools/wgrad_probe.py K_IN N_OUT [tuned]"""
import os, sys, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ELU = "elu" in sys.argv[3:]          # time the (Linear, ELU) backward: lsim_linear_elu_wgrad vs elu_backward + BLAS
if "tuned" in sys.argv[3:]:
    tdir = tempfile.mkdtemp(prefix="lsim_tunableop_")
    shutil.copy(os.path.join(ROOT, "isaacgymloco_amd", "learn", "tunableop_gfx950.csv"), os.path.join(tdir, "tuned0.csv"))
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"; os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(tdir, "tuned.csv"); os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
import torch
from isaacgymloco_amd.learn.fused_linear import linear_wgrad
B = 102400
k, n = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(B, k, device="cuda"); g = torch.randn(B, n, device="cuda")
if ELU:
    from isaacgymloco_amd.learn.fused_linear import _LinearEluFn
    w = torch.randn(n, k, device="cuda") * 0.05; bias = torch.zeros(n, device="cuda")
    for t in (x, w, bias):
        t.requires_grad_(True)
    for _ in range(12):
        y = torch.nn.functional.elu(torch.nn.functional.linear(x, w, bias)); y.backward(g)
        zf = _LinearEluFn.apply(x, w, bias); zf.backward(g)
else:
    for _ in range(12):
        a = g.t() @ x; b = g.sum(0)
        c, d = linear_wgrad(x, g)
torch.cuda.synchronize()
