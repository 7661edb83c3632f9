"""A/B of two builds of liblsim.so on IDENTICAL states: step both from the same arena every step (the variant's arena is overwritten with the
product's after each comparison) and report the per-step differences of the simulator outputs -- a defect in one build shows as outliers, rounding as
a 1e-6 floor.   usage: python tools/ab_step_compare.py variants/liblsim_X.so [task] [steps] [N]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgymloco_amd import lib  # noqa: E402
from isaacgymloco_amd.envs import config as C  # noqa: E402
from isaacgymloco_amd.envs.legged_robot import LeggedRobot  # noqa: E402

variant = os.path.abspath(sys.argv[1])
task = sys.argv[2] if len(sys.argv) > 2 else "aliengo_stairs"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
N = int(sys.argv[4]) if len(sys.argv) > 4 else 4096


def make(path):
    lib._lib = None
    lib.LIB_PATH = path
    cfg = C.TASKS[task][0]()
    cfg.env.num_envs = N
    return LeggedRobot(cfg, sim_device="cuda:0", seed=3, using_amp=(task == "aliengo_amp"))


product = os.path.join(ROOT, "isaacgymloco_amd", "csrc", "liblsim.so")
a, b = make(product), make(variant)
a.reset(); b.reset()
b._arena.copy_(a._arena)
g = torch.Generator(device="cuda:0").manual_seed(0)
keys = ["root_states", "dof_state", "contact_forces", "rew", "obs"]
worst = {k: 0.0 for k in keys}
outliers = {k: 0 for k in keys}
reset_mismatch = 0
hist_nc = np.zeros(40, np.int64)
for t in range(steps):
    act = torch.randn(N, 12, device="cuda:0", generator=g) * (1.0 if t % 3 else 2.5)
    a.step_device(act); b.step_device(act)
    torch.cuda.synchronize()
    reset_mismatch += int((a.buf["reset"] != b.buf["reset"]).sum())
    same = (a.buf["reset"] == b.buf["reset"])
    for k in keys:
        d = (a.buf[k].float() - b.buf[k].float()).abs().reshape(N, -1).max(dim=1).values
        d = d[same.bool()]
        worst[k] = max(worst[k], float(d.max()))
        tol = {"root_states": 1e-3, "dof_state": 1e-2, "contact_forces": 2.0, "rew": 1e-3, "obs": 1e-2}[k]
        outliers[k] += int((d > tol).sum())
    hist_nc += np.bincount(a.buf["contact_count"][:, 1].cpu().numpy().clip(0, 39), minlength=40)
    b._arena.copy_(a._arena)
print("task", task, "steps", steps, "N", N)
print("max |product - variant| over all env-steps:", {k: f"{v:.3g}" for k, v in worst.items()})
print("env-steps beyond tolerance:", outliers, " reset flags that differ:", reset_mismatch, "of", steps * N)
print("contacts in the last sub-step, histogram:", hist_nc[:12].tolist())
