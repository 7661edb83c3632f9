"""Diagnostics: time HIMPPO.update() alone (storage filled by one real rollout) so that a rocprofv3 kernel trace of this
script shows the learner's kernels without the rollout's.  usage: python3 tools/update_probe.py [n_updates]"""
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
import torch
from isaacgymloco_amd.envs import config as C
from isaacgymloco_amd.envs.legged_robot import LeggedRobot
from isaacgymloco_amd.learn.bench_train import train_cfg_dict
from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner

n_upd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if "--torch-sinkhorn" in sys.argv:      # A/B: force the torch statement of the Sinkhorn step
    import isaacgymloco_amd.learn.fused_linear as FL
    FL.sinkhorn_hip = None
    import isaacgymloco_amd.learn.modules as M
    _orig = M.sinkhorn
    def _torch_sinkhorn(scores, eps=0.05, iters=3):
        with torch.no_grad():
            Q = torch.exp(scores / eps).T
            K, B = Q.shape
            Q /= Q.sum()
            for _ in range(iters):
                Q /= Q.sum(dim=1, keepdim=True); Q /= K
                Q /= Q.sum(dim=0, keepdim=True); Q /= B
            return (Q * B).T
    M.sinkhorn = _torch_sinkhorn
if "--no-lr-sync" in sys.argv:        # A/B: what the per-minibatch .item() of the adaptive-lr rule costs (lr simply stays fixed)
    import isaacgymloco_amd.learn.him_ppo as HP
    HP.HIMPPO._adapt_lr = lambda self, *a, **k: None
if "--no-fused-elu" in sys.argv:       # A/B: Linear + ELU pairs through separate torch nodes (SkinnyLinear wgrad + elu_backward)
    import isaacgymloco_amd.learn.fused_linear as FL
    FL._eligible_fused_elu = lambda *a: False
if "--nn-linear" in sys.argv:           # A/B: plain nn.Linear backward (BLAS wgrad)
    import isaacgymloco_amd.learn.fused_linear as FL
    FL._eligible = lambda *a: False
cfg = C.TASKS["aliengo"][0]()
env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)
torch.manual_seed(1)
runner = HIMOnPolicyRunner(env, train_cfg_dict("aliengo"), log_dir=None, device="cuda:0")
runner.enable_graphs()
runner.alg.actor_critic.train()
if "--fused-adam" in sys.argv:          # A/B: torch's single-kernel Adam instead of the foreach implementation
    alg = runner.alg
    alg.optimizer = torch.optim.Adam(alg.actor_critic.parameters(), lr=alg.learning_rate, fused=True)
    est = alg.actor_critic.estimator
    est.optimizer = torch.optim.Adam(est.parameters(), lr=est.learning_rate, fused=True)
for it in range(n_upd + 1):
    for _ in range(runner.num_steps_per_env):
        runner.graphs.step()
    with torch.inference_mode():
        runner.alg.compute_returns(env.privileged_obs_buf)
    runner.graphs.end_iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.alg.update()
    torch.cuda.synchronize()
    print(f"update {it}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
