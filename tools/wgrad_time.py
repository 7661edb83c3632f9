#!/usr/bin/env python3
"""The library's weight-gradient kernels alone (CUDA events over 40 calls): plain g^T x and the (Linear, ELU)-backward form (g * elu'(z) formed
on the fly, g_y written or not), per layer shape of the learner at the BASELINE minibatch.  usage: python tools/wgrad_time.py [B]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from isaacgymloco_amd import lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
L = lib.load()
dev = "cuda:0"


def timeit(f, n=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K, N in [(512, 256), (256, 128), (238, 512), (64, 512), (270, 128), (128, 64), (45, 128)]:
    x, g, z = torch.randn(B, K, device=dev), torch.randn(B, N, device=dev), torch.randn(B, N, device=dev)
    need, parts = ctypes.c_size_t(), ctypes.c_int()
    lib.check(L.lsim_linear_wgrad_workspace(B, K, N, ctypes.byref(need), ctypes.byref(parts)))
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    dw, db, gy = torch.empty(N, K, device=dev), torch.empty(N, device=dev), torch.empty(B, N, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    plain = lambda: L.lsim_linear_wgrad(x.data_ptr(), K, g.data_ptr(), N, B, K, N, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), s)
    elu = lambda: L.lsim_linear_elu_wgrad(x.data_ptr(), K, g.data_ptr(), N, z.data_ptr(), N, B, K, N, dw.data_ptr(), db.data_ptr(), gy.data_ptr(), ws.data_ptr(), ws.numel(), s)
    elu_nogy = lambda: L.lsim_linear_elu_wgrad(x.data_ptr(), K, g.data_ptr(), N, z.data_ptr(), N, B, K, N, dw.data_ptr(), db.data_ptr(), None, ws.data_ptr(), ws.numel(), s)
    fl = 2.0 * B * K * N
    t0, t1, t2 = timeit(plain), timeit(elu), timeit(elu_nogy)
    print(f"{K:4d} -> {N:4d}  {fl / 1e9:6.1f} GFLOP  plain {t0:7.1f} us ({fl / t0 / 1e6:6.1f} TF)   elu + g_y {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF)   elu, no g_y {t2:7.1f} us ({fl / t2 / 1e6:6.1f} TF)")
