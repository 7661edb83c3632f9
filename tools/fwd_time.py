#!/usr/bin/env python3
"""lsim_linear_elu_forward against F.elu(F.linear(x, W, b)) with the TunableOp-selected BLAS kernels of the training loop, on the hidden layers
of the learner at the minibatch of 102 400 rows: maximum error and CUDA-event times.
usage: python tools/fwd_time.py"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tuned = os.path.join(ROOT, "isaacgymloco_amd", "learn", "tunableop_gfx950.csv")
tdir = tempfile.mkdtemp(prefix="lsim_tunableop_")
shutil.copy(tuned, os.path.join(tdir, "tuned0.csv"))
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(tdir, "tuned.csv"))
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from isaacgymloco_amd import lib  # noqa: E402

L = lib.load()
B = int(os.environ.get("ROWS", "102400"))


def ours(x, W, b, out):
    lib.check(L.lsim_linear_elu_forward(x.data_ptr(), x.stride(0), W.data_ptr(), b.data_ptr(), x.shape[0], W.shape[1], W.shape[0], out.data_ptr(), out.stride(0),
                                        torch.cuda.current_stream().cuda_stream), what="lsim_linear_elu_forward")
    return out


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator(device="cuda:0").manual_seed(1)
for name, K, N in (("actor 1", 64, 512), ("critic 1", 238, 512), ("hidden 2", 512, 256), ("hidden 3", 256, 128), ("encoder 1", 270, 128), ("encoder 2", 128, 64),
                   ("target 1", 45, 128)):
    if os.environ.get("FWD_ONLY") and os.environ["FWD_ONLY"] != name:
        continue
    x = torch.randn(B, K, device="cuda:0", generator=g)
    W = torch.randn(N, K, device="cuda:0", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda:0", generator=g) * 0.1
    out = torch.empty(B, N, device="cuda:0")
    ref = F.elu(F.linear(x, W, b))
    ours(x, W, b, out)
    err = (out - ref).abs().max().item()
    exact = F.elu(F.linear(x.double(), W.double(), b.double()))
    e_ours, e_ref = (out.double() - exact).abs().max().item(), (ref.double() - exact).abs().max().item()
    tb, tl, to = t(lambda: F.elu(F.linear(x, W, b))), t(lambda: F.linear(x, W, b)), t(lambda: ours(x, W, b, out))
    fl = 2.0 * B * K * N / 1e6
    print(f"{name:10s} {K:4d} -> {N:4d}: |ours - torch| {err:.2e} (vs fp64: ours {e_ours:.2e}, torch {e_ref:.2e})   torch linear+elu {tb:6.1f} us (linear {tl:6.1f})   "
          f"ours {to:6.1f} us = {fl / to:.0f} TFLOP/s   x{tb / to:.2f}")
