"""Time lsim_linear_elu_wgrad (weight gradient + ELU backward, incl. the sum of the partial results) with the bf16-pipe form off and on.
python tools/wgrad_split_time.py [out.json]"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgymloco_amd import lib

L = lib.load()
B = 102400
out = {}
for k_in, n_out in ((512, 256), (256, 128), (128, 128)):
    x = torch.randn(B, k_in, device="cuda:0")
    g = torch.randn(B, n_out, device="cuda:0")
    z = torch.nn.functional.elu(torch.randn(B, n_out, device="cuda:0"))
    row = {}
    for on in (0, 1):
        was = L.lsim_wgrad_split_bf16(on)
        need, parts = ctypes.c_size_t(), ctypes.c_int()
        lib.check(L.lsim_linear_wgrad_workspace(B, k_in, n_out, ctypes.byref(need), ctypes.byref(parts)))
        ws = torch.empty(need.value // 4, device="cuda:0")
        dw, db, gy = torch.empty(n_out, k_in, device="cuda:0"), torch.empty(n_out, device="cuda:0"), torch.empty(B, n_out, device="cuda:0")
        s = torch.cuda.current_stream().cuda_stream

        def call():
            lib.check(L.lsim_linear_elu_wgrad(x.data_ptr(), k_in, g.data_ptr(), n_out, z.data_ptr(), n_out, B, k_in, n_out, dw.data_ptr(), db.data_ptr(),
                                              gy.data_ptr(), ws.data_ptr(), need.value, s))
        for _ in range(5):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        row["split" if on else "fp32_pipe"] = {"us": round(us, 1), "tflops": round(2.0 * B * k_in * n_out / us * 1e-6, 1), "partials": parts.value}
        L.lsim_wgrad_split_bf16(was)
    out[f"{k_in}->{n_out}"] = row
    print(f"{k_in}->{n_out}", row)
if len(sys.argv) > 1:
    json.dump({"what": "lsim_linear_elu_wgrad, batch 102400, bf16-pipe form off / on (tools/wgrad_split_time.py)", "shapes": out}, open(sys.argv[1], "w"), indent=1)
