"""The update's activation pass in isolation (MI355X): torch's elu kernel on the minibatch shapes of the update (out of place / in
place), and the forward of one MLP layer by layer on the whole minibatch against depth-first over row chunks (every chunk runs through all
layers while its activations are still in the L2 / Infinity Cache).  Round 3's run (profiles/r03_elu_probe.json) also timed a prototype
streaming kernel of the library, not adopted: torch's kernel is at the HBM roof already.   python tools/elu_probe.py [rows]"""
import json
import sys

import torch
import torch.nn.functional as F

dev = "cuda:0"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 102400


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


out = {"rows": rows, "elementwise": [], "mlp": []}
for width in (512, 256, 128, 64):
    x = torch.randn(rows, width, device=dev)
    y = torch.empty_like(x)
    gb = 2 * x.numel() * 4 / 1e9
    t_torch = timed(lambda: F.elu(x))
    t_torch_in = timed(lambda: F.elu_(y.copy_(x)))
    out["elementwise"].append({"width": width, "torch_us": round(t_torch, 1), "torch_copy_plus_inplace_us": round(t_torch_in, 1),
                               "torch_TBps": round(gb / t_torch * 1e3, 2)})
    print(out["elementwise"][-1], flush=True)

# one MLP forward (critic: 238 -> 512 -> 256 -> 128 -> 1; actor: 64 + 45 ... -> 512 -> 256 -> 128 -> 12)
for name, dims in (("critic", (238, 512, 256, 128)), ("actor", (109, 512, 256, 128))):
    Ws = [torch.randn(dims[i + 1], dims[i], device=dev) * 0.05 for i in range(3)]
    bs = [torch.randn(dims[i + 1], device=dev) * 0.05 for i in range(3)]
    x = torch.randn(rows, dims[0], device=dev)
    acts = [torch.empty(rows, d, device=dev) for d in dims[1:]]

    def layerwise_torch():
        h = x
        for W, b in zip(Ws, bs):
            h = F.elu(F.linear(h, W, b))
        return h

    def layerwise_inplace():
        h = x
        for W, b, a in zip(Ws, bs, acts):
            torch.addmm(b, h, W.t(), out=a)
            F.elu_(a)
            h = a
        return h

    def chunked(nchunk):
        step = (rows + nchunk - 1) // nchunk
        for r0 in range(0, rows, step):
            h = x[r0:r0 + step]
            for W, b, a in zip(Ws, bs, acts):
                ac = a[r0:r0 + step]
                torch.addmm(b, h, W.t(), out=ac)
                F.elu_(ac)
                h = ac
        return acts[-1]

    ref = layerwise_torch()
    got = chunked(4).clone()
    torch.cuda.synchronize()
    rec = {"mlp": name, "dims": dims, "max_abs_diff_chunked_vs_layerwise": float((ref - got).abs().max()),
           "layerwise_torch_us": round(timed(layerwise_torch, 10), 1), "layerwise_inplace_us": round(timed(layerwise_inplace, 10), 1)}
    for n in (2, 4, 8, 16):
        rec[f"chunks{n}_us"] = round(timed(lambda: chunked(n), 10), 1)
    out["mlp"].append(rec)
    print(rec, flush=True)
print(json.dumps(out))
