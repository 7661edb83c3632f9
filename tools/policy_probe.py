"""GPU probe: lsim_policy_forward alone at N = 4096 (run under rocprofv3 for kernel durations / PMC counters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymloco_amd.learn.modules import HIMActorCritic
from isaacgymloco_amd.learn.fused_policy import PackedHimPolicy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
ac = HIMActorCritic(270, 238, 45, 12).to("cuda:0")
pk = PackedHimPolicy(ac)
obs, priv = torch.randn(N, 270, device="cuda:0"), torch.randn(N, 238, device="cuda:0")
mean, val = torch.empty(N, 12, device="cuda:0"), torch.empty(N, 1, device="cuda:0")
for _ in range(30):
    pk.forward(obs, priv, mean, val)
torch.cuda.synchronize()
