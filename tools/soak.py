"""Soak check on the GPU: many steps of N(0,1) actions on every task; reports non-finite values, state bounds and reset rates.
usage: python tools/soak.py [steps] [envs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from isaacgymloco_amd.envs import config as C
from isaacgymloco_amd.envs.legged_robot import LeggedRobot

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for task in ("aliengo", "aliengo_stairs", "aliengo_amp", "aliengo_recover", "go1", "go2"):
    cfg = C.TASKS[task][0]()
    cfg.env.num_envs = N
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=3, using_amp=(task == "aliengo_amp"))
    env.reset()
    g = torch.Generator(device="cuda:0").manual_seed(0)
    bad = 0
    resets = torch.zeros((), device="cuda:0")
    zmax = torch.zeros((), device="cuda:0"); vmax = torch.zeros((), device="cuda:0"); fmax = torch.zeros((), device="cuda:0")
    pen = torch.zeros((), device="cuda:0"); qdmax = torch.zeros((), device="cuda:0")
    n_pen = torch.zeros((), device="cuda:0"); n_fast = torch.zeros((), device="cuda:0")
    hard = torch.tensor([[env.model.dof_pos_lower[j], env.model.dof_pos_upper[j]] for j in range(12)], device="cuda:0")
    vlim = torch.tensor([env.model.dof_vel_limit[j] for j in range(12)], device="cuda:0")
    t0 = time.time()
    for i in range(steps):
        obs, priv, rew, done = env.step_device(torch.randn(N, 12, device="cuda:0", generator=g) * (1.0 if i % 500 < 400 else 3.0))
        if i % 50 == 0:
            fin = torch.isfinite(obs).all() & torch.isfinite(priv).all() & torch.isfinite(rew).all() & torch.isfinite(env.root_states).all() \
                & torch.isfinite(env.dof_state).all() & torch.isfinite(env.contact_forces).all()
            bad += int(not bool(fin))
        resets += done.sum()
        zmax = torch.maximum(zmax, env.root_states[:, 2].abs().max())
        vmax = torch.maximum(vmax, env.root_states[:, 7:13].abs().max())
        fmax = torch.maximum(fmax, env.contact_forces.abs().max())
        q, qd = env.dof_state.view(N, 12, 2)[..., 0], env.dof_state.view(N, 12, 2)[..., 1]
        over = torch.maximum(hard[:, 0] - q, q - hard[:, 1])
        pen = torch.maximum(pen, over.max())
        n_pen += (over > 0.05).sum(); n_fast += (qd.abs() > 1.05 * vlim).sum()
        qdmax = torch.maximum(qdmax, qd.abs().max())
    torch.cuda.synchronize()
    print(f"{task:15s} steps {steps} non-finite checks failed {bad}  kernel non-finite counter {int(env.nonfinite_envs)}  resets/env/1000 steps {float(resets) / N / steps * 1000:.1f}  "
          f"max |z| {float(zmax):.2f} m  max |v| {float(vmax):.1f}  max |contact force| {float(fmax):.0f} N  joint stops: max overshoot {float(pen):.3f} rad, {float(n_pen) / (N * 12 * steps) * 1e6:.1f} ppm of joint-steps beyond 0.05 rad; joint speed: max {float(qdmax):.1f}, {float(n_fast) / (N * 12 * steps) * 1e6:.1f} ppm above 1.05 x limit  ({time.time() - t0:.1f} s)", flush=True)
    del env
