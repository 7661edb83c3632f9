"""Host time of the pieces of one data-parallel gradient exchange (1-rank RCCL group on one GPU): the torch.distributed call, the wait,
GradArena.Bucket.adopt over ~80 parameters.  usage: python tools/collective_host_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = "cuda:0"
flat = torch.zeros(605540, device=dev)
x = torch.randn(4096, 4096, device=dev)
def timeit(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    h = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return h * 1e6
pg = dist.distributed_c10d._get_default_group()
opts = dist.AllreduceOptions(); opts.reduceOp = dist.ReduceOp.AVG
print("dist.all_reduce(async) + wait: %.1f us host" % timeit(lambda: dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True).wait()))
print("pg.allreduce + wait:           %.1f us host" % timeit(lambda: pg.allreduce([flat], opts).wait()))
print("pg.allreduce only:             %.1f us host" % timeit(lambda: pg.allreduce([flat], opts)))
print("tiny kernel launch (add_):     %.1f us host" % timeit(lambda: flat.add_(1.0)))
from isaacgymloco_amd.learn.fused_linear import GradArena
ps = [torch.nn.Parameter(torch.zeros(128, 64, device=dev)) for _ in range(80)]
ar = GradArena()
b = ar.bucket("all", ps, 5)
for p, v in zip(ps, b.views): p.grad = v
print("arena.bucket lookup + adopt:   %.1f us host" % timeit(lambda: ar.bucket("all", ps, 5).adopt()))
# device-side: how long is the queue stalled by one collective between two kernels?
def seq():
    torch.mm(x, x); pg.allreduce([flat], opts).wait(); flat.add_(1.0)
torch.cuda.synchronize()
for name, f in (("mm ; allreduce ; add", seq), ("mm ; add", lambda: (torch.mm(x, x), flat.add_(1.0)))):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); print("%-24s %.1f us per round (device-bound loop)" % (name, (time.perf_counter() - t) / 50 * 1e6))
dist.destroy_process_group()
