"""Timed training loop for bench.py --mode train: the reference's total_fps definition (HIMR:179), i.e.
env-steps/s over collection (policy inference + LeggedRobot.step + storage) PLUS compute_returns + update()."""
import ctypes
import time

import torch

from ..envs import config as C
from .runner import HIMOnPolicyRunner


def train_cfg_dict(task):
    ppo = C.TASKS[task][1]().to_dict()
    return {"runner": ppo["runner"], "algorithm": ppo["algorithm"], "policy": ppo["policy"]}


def run_train_bench(env, cfg, args, dev, rank, world, barrier):
    task = args.task if args.task in C.TASKS else "aliengo"
    tc = train_cfg_dict(task)
    torch.manual_seed(1)   # identical initial policy on every rank (then broadcast anyway)
    if C.TASKS[task][1]().runner_class_name == "HybridPolicyRunner":      # AMP configuration (AGA:297, AGA:329)
        from .hybrid import HybridPolicyRunner
        import numpy as np
        np.random.seed(1 + rank)
        runner = HybridPolicyRunner(env, tc, log_dir=None, device=str(dev))
    else:
        runner = HIMOnPolicyRunner(env, tc, log_dir=None, device=str(dev))
    import os
    use_graphs = os.environ.get("LSIM_NO_GRAPHS") != "1" and runner.enable_graphs()
    T = runner.num_steps_per_env
    K, W = args.steps, args.warmup
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    obs = env.get_observations().clone()
    critic_obs = env.get_privileged_observations().clone()
    runner.alg.actor_critic.train()
    state = dict(obs=obs, critic=critic_obs, in_iter=0, coll=0.0, learn=0.0, iters=0)

    def one_step(timed):
        t0 = time.perf_counter()
        if use_graphs:
            runner.graphs.step()
            state["critic"] = env.privileged_obs_buf
        else:
            with torch.inference_mode():
                state["obs"], state["critic"], _, _, _ = runner._rollout_step(state["obs"], state["critic"])
        state["in_iter"] += 1
        if state["in_iter"] == T:
            if timed:
                torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            with torch.inference_mode():
                runner.alg.compute_returns(state["critic"])
            if use_graphs:
                runner.graphs.end_iteration()
            runner.alg.update()
            state["in_iter"] = 0
            if timed:
                torch.cuda.synchronize(dev)
                state["learn"] += time.perf_counter() - t1
                state["iters"] += 1
            return t1 - t0
        return 0.0

    for _ in range(W):
        one_step(False)
    env._L.lsim_set_profiling(env._h, K)
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        one_step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    ms_a, ms_b, n = (ctypes.c_float * K)(), (ctypes.c_float * K)(), ctypes.c_int(K)
    env._L.lsim_read_profile(env._h, ms_a, ms_b, ctypes.byref(n))
    ka = sum(ms_a[i] for i in range(n.value)) / max(n.value, 1)
    kb = sum(ms_b[i] for i in range(n.value)) / max(n.value, 1)
    learn = state["learn"]
    iters = max(state["iters"], 1)
    extra = {"kernel_a_ms": ka, "kernel_b_ms": kb, "ppo_updates_timed": state["iters"],
             "learn_s_per_update": learn / iters if state["iters"] else None,
             "collection_s_per_iteration": (elapsed - learn) / (K / T) if K >= T else None,
             "ppo_iteration_wall_s": elapsed / (K / T) if K >= T else None,
             "collection_env_steps_per_s": world * env.num_envs * K / max(elapsed - learn, 1e-9),
             "rollout_hip_graphs": bool(use_graphs)}
    workload = (f"{task}: {type(runner).__name__} loop = policy inference + LeggedRobot.step + storage for {T} steps/iteration, then GAE + "
                f"HIMPPO.update (5 epochs x 4 minibatches), {env.num_envs} envs/GPU")
    return elapsed, extra, workload
