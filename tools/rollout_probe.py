"""GPU probe: is the rollout loop host-bound?  host time (no sync) vs wall time, and per-part timings."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgymloco_amd.envs import config as C
from isaacgymloco_amd.envs.legged_robot import LeggedRobot
from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner
from isaacgymloco_amd.learn.bench_train import train_cfg_dict
cfg = C.aliengo_cfg(); env = LeggedRobot(cfg, sim_device="cuda:0")
r = HIMOnPolicyRunner(env, train_cfg_dict("aliengo"), device="cuda:0")
obs, crit = env.get_observations().clone(), env.get_privileged_observations().clone()
def loop(n, what):
    global obs, crit
    torch.cuda.synchronize(); t0=time.perf_counter()
    with torch.inference_mode():
        for _ in range(n):
            if what=="full": obs, crit, *_ = r._rollout_step(obs, crit); 
            elif what=="act": a = r.alg.act(obs, crit)
            elif what=="env": env.step_device(a_fixed)
        if what=="full": r.alg.storage.clear()
    th=time.perf_counter()-t0; torch.cuda.synchronize(); tw=time.perf_counter()-t0
    print(f"{what:5s} host {1e3*th/n:.3f} ms/step  wall {1e3*tw/n:.3f} ms/step")
a_fixed = torch.randn(4096,12,device="cuda")
for w in ("full","act","env","full"):
    r.alg.storage.clear(); loop(100, w)
