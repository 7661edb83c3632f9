import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymloco_amd.learn.modules import HIMActorCritic
from isaacgymloco_amd.learn.fused_policy import PackedHimPolicy
N = 4096
torch.manual_seed(0)
ac = HIMActorCritic(270, 238, 45, 12).to("cuda:0")
pk = PackedHimPolicy(ac)
obs, priv = torch.randn(N, 270, device="cuda:0"), torch.randn(N, 238, device="cuda:0")
mean, val = torch.empty(N, 12, device="cuda:0"), torch.empty(N, 1, device="cuda:0")
for _ in range(30): pk.forward(obs, priv, mean, val)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): pk.forward(obs, priv, mean, val)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("LSIM_POLICY_ROWS16", "rows32"), "policy forward us:", e0.elapsed_time(e1) / 200 * 1e3)
