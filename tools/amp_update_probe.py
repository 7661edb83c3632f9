"""Diagnostics: HybridPPO.update() alone (storage + replay filled by real rollouts) so that a rocprofv3 kernel trace of this script shows the AMP
learner's kernels without the rollout's; prints the update's wall time.  usage: python3 tools/amp_update_probe.py [n_updates]
(rocprofv3 --kernel-trace ... -- python3 tools/amp_update_probe.py 2; then tools/trace_seq.py on the trace for the last minibatch)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tdir = os.path.join(ROOT, "gpurun_out", "tunableop")
os.makedirs(tdir, exist_ok=True)
import shutil
shutil.copy(os.path.join(ROOT, "isaacgymloco_amd", "learn", "tunableop_gfx950.csv"), os.path.join(tdir, "tuned0.csv"))
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(tdir, "tuned.csv"))
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
import numpy as np
import torch
from isaacgymloco_amd.envs import config as C
from isaacgymloco_amd.envs.legged_robot import LeggedRobot
from isaacgymloco_amd.learn.bench_train import train_cfg_dict
from isaacgymloco_amd.learn.hybrid import HybridPolicyRunner

n_upd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = C.TASKS["aliengo_amp"][0]()
env = LeggedRobot(cfg, sim_device="cuda:0", seed=1, using_amp=True)
torch.manual_seed(1)
np.random.seed(1)
runner = HybridPolicyRunner(env, train_cfg_dict("aliengo_amp"), log_dir=None, device="cuda:0")
runner.enable_graphs()
runner.alg.actor_critic.train()
for it in range(n_upd + 1):
    for _ in range(runner.num_steps_per_env):
        runner.graphs.step()
    with torch.inference_mode():
        runner.alg.compute_returns(env.privileged_obs_buf)
    runner.graphs.end_iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = runner.alg.update()
    torch.cuda.synchronize()
    print(f"update {it}: {1e3 * (time.perf_counter() - t0):.1f} ms  losses {[round(float(x), 5) for x in res]}", flush=True)
