"""Learnability / stability check of the AMP configuration (HybridPolicyRunner, fused rollout): N iterations, prints the update outputs."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shutil, tempfile
tdir = tempfile.mkdtemp(prefix="lsim_tunableop_")
shutil.copy(os.path.join(ROOT, "isaacgymloco_amd", "learn", "tunableop_gfx950.csv"), os.path.join(tdir, "tuned0.csv"))
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1"); os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(tdir, "tuned.csv")); os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
import numpy as np
import torch
from isaacgymloco_amd.envs import config as C
from isaacgymloco_amd.envs.legged_robot import LeggedRobot
from isaacgymloco_amd.learn.bench_train import train_cfg_dict
from isaacgymloco_amd.learn.hybrid import HybridPolicyRunner

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
task = sys.argv[2] if len(sys.argv) > 2 else "aliengo_amp"
cfg = C.TASKS[task][0]()
env = LeggedRobot(cfg, sim_device="cuda:0", seed=1, using_amp=True)
torch.manual_seed(1); np.random.seed(1)
run = HybridPolicyRunner(env, train_cfg_dict(task), log_dir=None, device="cuda:0")
run.enable_graphs()
t0 = time.time()
for it in range(iters):
    run.learn(1, init_at_random_ep_len=(it == 0))
    if it % 10 == 0 or it == iters - 1:
        u = run.last_update
        st = run.alg.storage
        print(json.dumps(dict(it=it, value_loss=u[0], surrogate=u[1], est=u[2], swap=u[3], amp_loss=u[4], grad_pen=u[5], policy_d=u[6], expert_d=u[7],
                              mean_reward=float(st.rewards.mean()), done_rate=float(st.dones.float().mean()), lr=run.alg.learning_rate,
                              std=float(run.alg.actor_critic.std.mean()), wall_s=time.time() - t0)), flush=True)
