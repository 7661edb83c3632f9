"""GPU probe: fp32 GEMM throughput of the learner's layer shapes under the BLAS back-ends PyTorch-ROCm offers."""
import os, sys, time, torch
B = 102400
layers = [(64,512),(512,256),(256,128),(128,12),(238,512),(128,1),(270,128),(128,64),(64,19),(45,128),(64,16)]
def bench(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
def run(tag):
    tot=0; totf=0
    for K,N in layers:
        x=torch.randn(B,K,device='cuda'); w=torch.randn(N,K,device='cuda'); g=torch.randn(B,N,device='cuda'); b=torch.randn(N,device='cuda')
        t1=bench(lambda: torch.nn.functional.linear(x,w,b)); t2=bench(lambda: g@w); t3=bench(lambda: g.t()@x)
        fl=2*B*K*N
        tot+=t1+t2+t3; totf+=3*fl
        print(f'{tag} {K:4d}x{N:4d} fwd {fl/t1/1e12:6.1f} dX {fl/t2/1e12:6.1f} dW {fl/t3/1e12:6.1f} TF/s  ({(t1+t2+t3)*1e3:.2f} ms)')
    print(f'{tag} TOTAL {tot*1e3:.2f} ms per minibatch-equivalent, {totf/tot/1e12:.1f} TF/s')
mode=sys.argv[1]
if mode in ('cublas','cublaslt'): torch.backends.cuda.preferred_blas_library(mode)
print('preferred', torch.backends.cuda.preferred_blas_library(), 'tunable', os.environ.get('PYTORCH_TUNABLEOP_ENABLED'))
run(mode)
