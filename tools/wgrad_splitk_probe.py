"""GPU probe: weight gradient dW = g^T x of the learner's layers three ways -- the library's MFMA kernel (lsim_linear_wgrad), one BLAS GEMM
(g.t() @ x), and split-K through a batched BLAS GEMM (bmm over S slices of the 102 400-row batch, then a sum over the slices).
Timed with CUDA events over 30 repetitions.   usage: python tools/wgrad_splitk_probe.py"""
import os, sys, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdir = tempfile.mkdtemp(prefix="lsim_tunableop_")
shutil.copy(os.path.join(ROOT, "isaacgymloco_amd", "learn", "tunableop_gfx950.csv"), os.path.join(tdir, "tuned0.csv"))
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"; os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(tdir, "tuned.csv"); os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
import torch
from isaacgymloco_amd.learn.fused_linear import linear_wgrad

B = 102400


def timeit(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


for K, N in [(512, 256), (256, 128), (238, 512), (64, 512), (270, 128), (128, 64), (45, 128), (1024, 512), (60, 1024)]:
    x = torch.randn(B, K, device="cuda"); g = torch.randn(B, N, device="cuda")
    ref = g.t() @ x
    row = [f"{K:4d}->{N:4d}  {2 * B * K * N / 1e9:6.1f} GFLOP"]
    try:
        t = timeit(lambda: linear_wgrad(x, g)); row.append(f"lsim {t:7.1f} us ({2 * B * K * N / t / 1e6:5.1f} TF)")
    except Exception as e:
        row.append(f"lsim n/a ({type(e).__name__})")
    t = timeit(lambda: g.t() @ x); row.append(f"blas {t:7.1f} us ({2 * B * K * N / t / 1e6:5.1f} TF)")
    for S in (8, 16, 32, 64):
        def f():
            return torch.bmm(g.view(S, B // S, N).transpose(1, 2), x.view(S, B // S, K)).sum(0)
        out = f()
        err = float((out - ref).abs().max() / ref.abs().max())
        t = timeit(f); row.append(f"bmm{S} {t:7.1f} us ({2 * B * K * N / t / 1e6:5.1f} TF, err {err:.1e})")
    print("  ".join(row))
