"""Where lsim_k_linear_wgrad_split's time goes: the kernel with one phase knocked out at a time (-DLS_SP_KNOCKOUT=n diagnostic builds of the
library; results are wrong by construction, only the time is read).  python tools/wgrad_split_probe.py build | run"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "isaacgymloco_amd", "csrc", "variants")
NAMES = {0: "whole kernel", 1: "no MFMA phase", 2: "no global loads in the loop", 3: "no split / LDS writes"}


def build():
    from isaacgymloco_amd.csrc import build as B
    for k in NAMES:
        B.build_variant(os.path.join(VAR, f"liblsim_spko{k}.so"), [f"-DLS_SP_KNOCKOUT={k}"])


def run():
    import torch
    from isaacgymloco_amd import lib
    B_ = 102400
    for k_in, n_out in ((512, 256), (256, 128)):
        x = torch.randn(B_, k_in, device="cuda:0"); g = torch.randn(B_, n_out, device="cuda:0")
        z = torch.nn.functional.elu(torch.randn(B_, n_out, device="cuda:0"))
        for k in NAMES:
            L = lib.load_path(os.path.join(VAR, f"liblsim_spko{k}.so"))
            L.lsim_wgrad_split_bf16(1)
            need, parts = ctypes.c_size_t(), ctypes.c_int()
            lib.check(L.lsim_linear_wgrad_workspace(B_, k_in, n_out, ctypes.byref(need), ctypes.byref(parts)))
            ws = torch.empty(need.value // 4, device="cuda:0")
            dw, db, gy = torch.empty(n_out, k_in, device="cuda:0"), torch.empty(n_out, device="cuda:0"), torch.empty(B_, n_out, device="cuda:0")
            s = torch.cuda.current_stream().cuda_stream

            def call():
                lib.check(L.lsim_linear_elu_wgrad(x.data_ptr(), k_in, g.data_ptr(), n_out, z.data_ptr(), n_out, B_, k_in, n_out, dw.data_ptr(), db.data_ptr(),
                                                  gy.data_ptr(), ws.data_ptr(), need.value, s))
            for _ in range(5):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                call()
            e1.record(); torch.cuda.synchronize()
            print(f"{k_in}->{n_out}  {NAMES[k]:32s} {e0.elapsed_time(e1) * 1000 / 30:8.1f} us (incl. the partial-sum launch)")


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
