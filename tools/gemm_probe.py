import os, sys, time, shutil
ROOT='/root/repo' if os.path.exists('/root/repo') else os.getcwd()
tdir=os.path.join(os.getcwd(),'gpurun_out','tunableop'); os.makedirs(tdir,exist_ok=True)
shutil.copy(os.path.join(os.getcwd(),'isaacgymloco_amd','learn','tunableop_gfx950.csv'), os.path.join(tdir,'tuned0.csv'))
os.environ['PYTORCH_TUNABLEOP_ENABLED']='1'; os.environ['PYTORCH_TUNABLEOP_FILENAME']=os.path.join(tdir,'tuned.csv'); os.environ['PYTORCH_TUNABLEOP_TUNING']='0'
import torch
B=102400
layers=[(64,512),(512,256),(256,128),(128,12),(238,512),(128,1),(270,128),(128,64),(64,19),(45,128),(64,16),(16,32)]
def bench(f,n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
tot=[0,0,0,0]
for K,N in layers:
    x=torch.randn(B,K,device='cuda'); w=torch.randn(N,K,device='cuda'); g=torch.randn(B,N,device='cuda'); b=torch.randn(N,device='cuda')
    t1=bench(lambda: torch.nn.functional.linear(x,w,b)); t2=bench(lambda: g@w); t3=bench(lambda: g.t()@x); t4=bench(lambda: g.sum(0))
    ideal=lambda nbytes: nbytes/5e12
    print(f'{K:4d}->{N:4d} fwd {t1*1e6:7.1f} us  dX {t2*1e6:7.1f} us  dW {t3*1e6:7.1f} us (min-traffic {ideal(4*B*(K+N))*1e6:5.1f} us)  db {t4*1e6:6.1f} us')
    for i,t in enumerate((t1,t2,t3,t4)): tot[i]+=t
print('totals ms: fwd %.2f dX %.2f dW %.2f db %.2f'%tuple(1e3*t for t in tot))
