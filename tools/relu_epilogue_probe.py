"""Does torch._addmm_activation (GEMM with a bias + ReLU epilogue in hipBLASLt) beat relu(linear(x)) for the AMP discriminator's layers
(DISC:18-25: 60 -> 1024 -> 512, ReLU) at the update's minibatch of 102 400 rows?  forward only and forward + backward."""
import sys
import torch

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
for k, n in ((60, 1024), (1024, 512)):
    x = torch.randn(B, k, device=dev, requires_grad=True)
    lin = torch.nn.Linear(k, n).to(dev)

    def plain():
        return torch.relu(torch.nn.functional.linear(x, lin.weight, lin.bias))

    def fused():
        return torch._addmm_activation(lin.bias, x, lin.weight.t(), use_gelu=False)
    torch.testing.assert_close(plain(), fused(), rtol=1e-5, atol=1e-5)
    for name, fn in (("relu(linear)", plain), ("_addmm_activation", fused)):
        for bwd in ((False, True) if name.startswith("relu") else (False,)):     # (_addmm_activation has no autograd derivative: a custom Function would wrap it)
            for _ in range(3):
                y = fn()
                if bwd:
                    y.sum().backward()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y = fn()
                if bwd:
                    y.sum().backward()
            e1.record(); torch.cuda.synchronize()
            print(f"{k}->{n} {name:20s} {'fwd+bwd' if bwd else 'fwd    '} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
