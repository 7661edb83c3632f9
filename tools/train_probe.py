"""Learnability check (not a benchmark): train aliengo (flat-ish terrain mix of the task config, 4096 envs) for a number of PPO
iterations with the build's runner and record mean episode reward / length / tracking reward per iteration.
usage: python tools/train_probe.py [iterations] [out.json] [task] [seed] [checkpoint.pt]
checkpoint.pt: the trained actor-critic + the simulator's curriculum state (LeggedRobot.state_dict) + what the product's own physics did under the trained
policy in a closed-loop evaluation rollout (tools/trained_policy_physics.py replays the same policy through the CPU oracle's variants)"""
import json
import os
import sys
import time

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
from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "train_curve.json")
task = sys.argv[3] if len(sys.argv) > 3 else "aliengo"
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cfg = C.TASKS[task][0]()
env = LeggedRobot(cfg, sim_device="cuda:0", seed=seed)
torch.manual_seed(seed)
runner = HIMOnPolicyRunner(env, train_cfg_dict(task), log_dir=None, device="cuda:0")
runner.enable_graphs()
runner.alg.actor_critic.train()
env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
N, T = env.num_envs, runner.num_steps_per_env
cur_r = torch.zeros(N, device="cuda:0"); cur_l = torch.zeros(N, device="cuda:0")
curve = []
t_start = time.time()
for it in range(iters):
    fin = torch.zeros(3, device="cuda:0")
    track = torch.zeros(1, device="cuda:0")
    with torch.inference_mode():
        for _ in range(T):
            runner.graphs.step()
            d = env.reset_buf.float()
            cur_r += env.rew_buf; cur_l += 1
            fin += torch.stack((d.sum(), (cur_r * d).sum(), (cur_l * d).sum()))
            cur_r *= 1 - d; cur_l *= 1 - d
            track += env.rew_buf.mean()
        runner.alg.compute_returns(env.privileged_obs_buf)
    runner.graphs.end_iteration()
    vl, sl, el, swl = runner.alg.update()[:4]
    f = fin.tolist()
    rec = dict(it=it, mean_step_reward=float(track) / T, finished=f[0], mean_ep_reward=f[1] / max(f[0], 1), mean_ep_len=f[2] / max(f[0], 1),
               value_loss=vl, surrogate_loss=sl, est_loss=el, swap_loss=swl, lr=runner.alg.learning_rate,
               action_std=float(runner.alg.actor_critic.std.mean()), terrain_level=float(env.terrain_levels.float().mean()),
               wall_s=time.time() - t_start)
    curve.append(rec)
    if it % 10 == 0 or it == iters - 1:
        print(json.dumps(rec), flush=True)
os.makedirs(os.path.dirname(out), exist_ok=True)
ckpt = sys.argv[5] if len(sys.argv) > 5 else None
evals = None
if ckpt:
    # closed loop under the TRAINED policy (mean actions, as play.py runs it): the quantities the modelling-distance tables of DESIGN.md section 4 read
    ac = runner.alg.actor_critic
    ac.eval()
    pen, term = env.penalised_contact_indices, env.termination_contact_indices
    acc = torch.zeros(6, device="cuda:0", dtype=torch.float64)
    steps_eval = 1000
    with torch.inference_mode():
        obs = env.get_observations()
        for _ in range(steps_eval):
            env.step_device(ac.act_inference(obs))
            obs = env.obs_buf
            cf = env.contact_forces
            hit = (torch.linalg.norm(cf[:, term, :], dim=-1) > 1.0).any(dim=1)
            acc += torch.stack(((env.reset_buf & ~env.time_out_buf).double().sum(), hit.double().sum(),
                                (torch.linalg.norm(cf[:, pen, :], dim=-1) > 0.1).double().sum(), (env.buf["contact_count"][:, 0] > 8).double().sum(),
                                env.reset_buf.double().sum(), env.terrain_levels.double().sum() / steps_eval))
    a = acc.tolist()
    es = N * steps_eval
    evals = dict(steps=steps_eval, num_envs=N, terminations_not_timeout_per_env_step=a[0] / es, base_contact_per_env_step=a[1] / es,
                 collision_count_per_env_step=a[2] / es, contact_cap_hits_per_env_step=a[3] / es, resets_per_env_step=a[4] / es,
                 mean_terrain_level=a[5] / N, nonfinite_envs=int(env.nonfinite_envs))
    print("eval", json.dumps(evals), flush=True)
    torch.save({"task": task, "model_state_dict": {k: v.cpu() for k, v in ac.state_dict().items()}, "env_state": env.state_dict(), "eval_hip": evals,
                "iterations": iters}, ckpt)
json.dump(dict(task=task, num_envs=N, steps_per_iteration=T, curve=curve, eval_hip=evals), open(out, "w"))
