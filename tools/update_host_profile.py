"""Where the HOST time of one PPO update goes (diagnostics): cProfile around HIMPPO.update() / HybridPPO.update() on the GPU box, after warm-up.  The update
enqueues ~3000 launches without a read-back; its host time (bench line: update_host_enqueue_s, 0.034-0.047 s of a 0.066 s update) is what a rank process of an
8-rank run needs from its share of the host's CPUs.   usage: python tools/update_host_profile.py [task] [top]"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shutil, tempfile
tdir = tempfile.mkdtemp(prefix="lsim_tunableop_")
shutil.copy(os.path.join(ROOT, "isaacgymloco_amd", "learn", "tunableop_gfx950.csv"), os.path.join(tdir, "tuned0.csv"))
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(tdir, "tuned.csv"))
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
import torch
from isaacgymloco_amd.envs import config as C
from isaacgymloco_amd.envs.legged_robot import LeggedRobot
from isaacgymloco_amd.learn.bench_train import train_cfg_dict

task = sys.argv[1] if len(sys.argv) > 1 else "aliengo"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
amp = task == "aliengo_amp"
cfg = C.TASKS[task][0]()
env = LeggedRobot(cfg, sim_device="cuda:0", seed=1, using_amp=amp)
torch.manual_seed(1)
if amp:
    from isaacgymloco_amd.learn.hybrid import HybridPolicyRunner as R
else:
    from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner as R
runner = R(env, train_cfg_dict(task), log_dir=None, device="cuda:0")
runner.enable_graphs()
runner.alg.actor_critic.train()
T = runner.num_steps_per_env


def collect():
    with torch.inference_mode():
        for _ in range(T):
            runner.graphs.step()
        runner.alg.compute_returns(env.privileged_obs_buf)
    runner.graphs.end_iteration()


for _ in range(6):
    collect(); runner.alg.update()
torch.cuda.synchronize()
pr = cProfile.Profile()
enq = []
for _ in range(3):
    collect()
    torch.cuda.synchronize()
    pr.enable()
    runner.alg.update()
    pr.disable()
    enq.append(runner.alg.update_enqueue_s)
print(f"task {task}: update_enqueue_s under cProfile {[round(e, 4) for e in enq]} (the profiler roughly doubles it)")
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats("tottime").print_stats(top)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats("cumulative").print_stats(top)
print(s.getvalue())
