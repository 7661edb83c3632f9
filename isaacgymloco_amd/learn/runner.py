"""HIMOnPolicyRunner: the reference's rollout + learn loop (HIMR:44-255) on top of the HIP environment.

    runner = HIMOnPolicyRunner(env, train_cfg_dict, log_dir, device)
    runner.learn(num_learning_iterations, init_at_random_ep_len=True)

Call order per iteration is the reference's: 100 x {alg.act -> env.step -> patch next_critic_obs with the termination
observations (HIMR:122-123) -> alg.process_env_step} -> alg.compute_returns -> alg.update.  Two ways to take the step:
  * `fast=True` (default with our env): `env.step_device` + a masked `torch.where` patch -- no host sync per step
    (the reference syncs at LR:225 and HIMR:133-135);
  * `fast=False`: the reference's exact 7-tuple `env.step` with index assignment, works with any VecEnv.
Checkpoints use the reference's keys (HIMR:233-240) so they interchange with the reference's play.py.
Data parallel: one process per GPU; `DistCtx` averages gradients (him_ppo.py).  Per-rank seeds = (seed, rank).
"""
import os
import time
from collections import deque

import torch

from .him_ppo import DistCtx, HIMPPO
from .modules import HIMActorCritic

_POLICIES = {"HIMActorCritic": HIMActorCritic}
_ALGOS = {"HIMPPO": HIMPPO}


class HIMOnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device="cpu", fast=None):
        self.cfg, self.alg_cfg, self.policy_cfg = train_cfg["runner"], train_cfg["algorithm"], train_cfg["policy"]
        self.device, self.env = device, env
        num_critic_obs = env.num_privileged_obs if env.num_privileged_obs is not None else env.num_obs
        self.num_actor_obs, self.num_critic_obs = env.num_obs, num_critic_obs
        self.dist_ctx = DistCtx()
        actor_critic = _POLICIES[self.cfg["policy_class_name"]](env.num_obs, num_critic_obs, env.num_one_step_obs, env.num_actions,
                                                                **self.policy_cfg).to(device)
        self.alg = _ALGOS[self.cfg["algorithm_class_name"]](actor_critic, device=device, dist_ctx=self.dist_ctx, **self.alg_cfg)
        self.num_steps_per_env = self.cfg["num_steps_per_env"]
        self.save_interval = self.cfg["save_interval"]
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [env.num_obs], [env.num_privileged_obs], [env.num_actions])
        self.log_dir, self.writer = log_dir, None
        self.tot_timesteps, self.tot_time, self.current_learning_iteration = 0, 0.0, 0
        self.fast = hasattr(env, "step_device") if fast is None else fast
        self.last_perf = {}
        self.graphs = None
        env.reset()

    graphs = None   # GraphedRollout once enable_graphs() succeeded (subclasses with their own __init__ inherit the default)
    _pending_draw_counter = None   # a checkpoint's sampler counter loaded before enable_graphs()

    def enable_graphs(self):
        """Switch the rollout step to the fused device path (graph_rollout.py): HIP kernels for the policy forward, sampling and
        storage writes, no host round trips.  Needs the build's GPU environment (step_device)."""
        if self.fast and str(self.device).startswith("cuda"):
            self.graphs = self._make_fused_rollout()
            if self.graphs is not None:
                self.alg.enable_device_lr()
                if self._pending_draw_counter is not None:
                    self.graphs.set_draw_counter(self._pending_draw_counter)
                    self._pending_draw_counter = None
        return self.graphs is not None

    def _make_fused_rollout(self):
        from .graph_rollout import GraphedRollout
        return GraphedRollout(self) if type(self) is HIMOnPolicyRunner else None

    # ------------------------------------------------------------------ rollout
    def _rollout_step(self, obs, critic_obs):
        env, alg = self.env, self.alg
        actions = alg.act(obs, critic_obs)
        if self.fast:
            obs, priv, rewards, dones = env.step_device(actions)
            infos = env.extras
            obs = obs.clone()                          # live simulator buffers: keep a copy across the next step
            critic_obs = (priv if priv is not None else obs).clone()
            next_critic_obs = torch.where(dones.unsqueeze(1), env.termination_privileged_obs_buf, critic_obs)
        else:
            obs, priv, rewards, dones, infos, term_ids, term_priv = env.step(actions)[:7]
            critic_obs = priv if priv is not None else obs
            obs, critic_obs, rewards, dones = obs.to(self.device), critic_obs.to(self.device), rewards.to(self.device), dones.to(self.device)
            next_critic_obs = critic_obs.clone().detach()
            next_critic_obs[term_ids.to(self.device)] = term_priv.to(self.device).clone().detach()
        alg.process_env_step(rewards, dones, infos, next_critic_obs)
        return obs, critic_obs, rewards, dones, infos

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        env = self.env
        if self.log_dir is not None and self.writer is None:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(log_dir=self.log_dir, flush_secs=10)
            except Exception:
                self.writer = None   # tensorboard is optional; console logging still works
        if init_at_random_ep_len:
            env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        obs = env.get_observations()
        priv = env.get_privileged_observations()
        critic_obs = priv if priv is not None else obs
        obs, critic_obs = obs.to(self.device).clone(), critic_obs.to(self.device).clone()
        self.alg.actor_critic.train()
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
        cur_reward_sum = torch.zeros(env.num_envs, dtype=torch.float, device=self.device)
        cur_episode_length = torch.zeros(env.num_envs, dtype=torch.float, device=self.device)
        tot_iter = self.current_learning_iteration + num_learning_iterations
        for it in range(self.current_learning_iteration, tot_iter):
            start = time.time()
            ep_stats = []
            fin = torch.zeros(3, device=self.device)   # finished episodes this iteration: count, sum reward, sum length
            with torch.inference_mode():
                for _ in range(self.num_steps_per_env):
                    if self.graphs is not None:
                        self.graphs.step()
                        obs, critic_obs, rewards, dones, infos = env.obs_buf, env.privileged_obs_buf, env.rew_buf, env.reset_buf, env.extras
                    else:
                        obs, critic_obs, rewards, dones, infos = self._rollout_step(obs, critic_obs)
                    if self.log_dir is not None:
                        cur_reward_sum += rewards
                        cur_episode_length += 1
                        d = dones.float()
                        fin += torch.stack((d.sum(), (cur_reward_sum * d).sum(), (cur_episode_length * d).sum()))
                        cur_reward_sum *= 1.0 - d
                        cur_episode_length *= 1.0 - d
                        if self.fast:
                            ep_stats.append(env.stats_row().clone())
                        elif "episode" in infos:
                            ep_stats.append(infos["episode"])
                if self.device != "cpu" and torch.cuda.is_available():
                    torch.cuda.synchronize()
                stop = time.time()
                collection_time = stop - start
                start = stop
                self.alg.compute_returns(critic_obs)
            if self.graphs is not None:
                self.graphs.end_iteration()
            from .. import lib as _lib
            with _lib.roctx_range("ppo_update"):
                update_out = self.alg.update()  # HIMPPO: 4 values; HybridPPO: 8 (adds AMP loss, grad penalty, policy / expert prediction)
            mean_value_loss, mean_surrogate_loss, mean_estimation_loss, mean_swap_loss = update_out[:4]
            self.last_update = update_out
            if self.device != "cpu" and torch.cuda.is_available():
                torch.cuda.synchronize()
            learn_time = time.time() - start
            self.last_perf = dict(collection_time=collection_time, learn_time=learn_time,
                                  fps=self.num_steps_per_env * env.num_envs / (collection_time + learn_time))
            if it == self.current_learning_iteration and self.graphs is not None:
                # the update is launch-bound (~2 200 launches in 78 ms); a full pass of Python's cyclic collector over everything the set-up
                # left behind stalls it for ~60 ms every few iterations.  Collect once, then park the survivors in the permanent generation.
                import gc
                gc.collect(); gc.freeze()
            if self.log_dir is not None:
                f = fin.tolist()
                if f[0] > 0:
                    rewbuffer.append(f[1] / f[0])
                    lenbuffer.append(f[2] / f[0])
                self.log(dict(it=it, tot_iter=tot_iter, collection_time=collection_time, learn_time=learn_time, ep_stats=ep_stats,
                              rewbuffer=rewbuffer, lenbuffer=lenbuffer, mean_value_loss=mean_value_loss,
                              mean_surrogate_loss=mean_surrogate_loss, mean_estimation_loss=mean_estimation_loss,
                              mean_swap_loss=mean_swap_loss))
                if it % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
        self.current_learning_iteration += num_learning_iterations
        if self.log_dir is not None:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    # ------------------------------------------------------------------ logging (HIMR:159-231)
    def _episode_means(self, ep_stats):
        from .. import abi
        out = {}
        if not ep_stats:
            return out
        if isinstance(ep_stats[0], dict):
            for key in ep_stats[0]:
                vals = [torch.as_tensor(e[key], dtype=torch.float32).reshape(-1).to(self.device) for e in ep_stats]
                out[key] = float(torch.cat(vals).mean())
            return out
        st = torch.stack(ep_stats)                              # (steps, LSIM_STATS_SIZE)
        S = abi.STATS
        cnt = st[:, S["reset_count"]]
        valid = cnt > 0
        if not bool(valid.any()):
            return out
        per_step = st[valid][:, S["episode_sums"]:S["episode_sums"] + abi.NUM_REWARD_TERMS] / cnt[valid].unsqueeze(1) / self.env.dt
        means = per_step.mean(dim=0).tolist()
        for name in self.env.reward_scales:
            out["rew_" + name] = means[abi.REWARD_IDS[name]]
        out["terrain_level"] = float(self.env.terrain_levels.float().mean())
        out["max_command_x"] = float(st[-1, S["cmd_ranges"] + 1])
        return out

    def log(self, locs, width=80, pad=35):
        self.tot_timesteps += self.num_steps_per_env * self.env.num_envs * self.dist_ctx.world
        iteration_time = locs["collection_time"] + locs["learn_time"]
        self.tot_time += iteration_time
        fps = int(self.num_steps_per_env * self.env.num_envs * self.dist_ctx.world / iteration_time)
        ep = self._episode_means(locs["ep_stats"])
        mean_std = float(self.alg.actor_critic.std.mean())
        scalars = {"Loss/value_function": locs["mean_value_loss"], "Loss/surrogate": locs["mean_surrogate_loss"],
                   "Loss/Estimation Loss": locs["mean_estimation_loss"], "Loss/Swap Loss": locs["mean_swap_loss"],
                   "Loss/learning_rate": self.alg.learning_rate, "Policy/mean_noise_std": mean_std, "Perf/total_fps": fps,
                   "Perf/collection time": locs["collection_time"], "Perf/learning_time": locs["learn_time"]}
        if len(locs["rewbuffer"]) > 0:
            scalars["Train/mean_reward"] = sum(locs["rewbuffer"]) / len(locs["rewbuffer"])
            scalars["Train/mean_episode_length"] = sum(locs["lenbuffer"]) / len(locs["lenbuffer"])
        if self.writer is not None:
            for k, v in ep.items():
                self.writer.add_scalar("Episode/" + k, v, locs["it"])
            for k, v in scalars.items():
                self.writer.add_scalar(k, v, locs["it"])
        if self.dist_ctx.enabled and self.dist_ctx.dist.get_rank() != 0:
            return
        head = f" Learning iteration {locs['it']}/{locs['tot_iter']} "
        lines = ["#" * width, head.center(width), "",
                 f"{'Computation:':>{pad}} {fps:.0f} steps/s (collection: {locs['collection_time']:.3f}s, learning {locs['learn_time']:.3f}s)",
                 f"{'Value function loss:':>{pad}} {locs['mean_value_loss']:.4f}", f"{'Surrogate loss:':>{pad}} {locs['mean_surrogate_loss']:.4f}",
                 f"{'Estimation loss:':>{pad}} {locs['mean_estimation_loss']:.4f}", f"{'Swap loss:':>{pad}} {locs['mean_swap_loss']:.4f}",
                 f"{'Mean action noise std:':>{pad}} {mean_std:.2f}"]
        if "Train/mean_reward" in scalars:
            lines += [f"{'Mean reward:':>{pad}} {scalars['Train/mean_reward']:.2f}", f"{'Mean episode length:':>{pad}} {scalars['Train/mean_episode_length']:.2f}"]
        lines += [f"{f'Mean episode {k}:':>{pad}} {v:.4f}" for k, v in ep.items()]
        done = locs["it"] + 1 - self.current_learning_iteration
        eta = self.tot_time / max(done, 1) * (locs["tot_iter"] - locs["it"])
        lines += ["-" * width, f"{'Total timesteps:':>{pad}} {self.tot_timesteps}", f"{'Iteration time:':>{pad}} {iteration_time:.2f}s",
                  f"{'Total time:':>{pad}} {format_time(self.tot_time)}", f"{'ETA:':>{pad}} {format_time(eta)}"]
        print("\n".join(lines))

    # ------------------------------------------------------------------ checkpoints (HIMR:233-255)
    def save(self, path, infos=None):
        if self.graphs is not None:
            self.graphs.flush()
        if self.dist_ctx.enabled and self.dist_ctx.dist.get_rank() != 0:
            return
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        d = {"model_state_dict": self.alg.actor_critic.state_dict(), "optimizer_state_dict": self._portable_optimizer_state(self.alg.optimizer),
             "estimator_optimizer_state_dict": self._portable_optimizer_state(self.alg.actor_critic.estimator.optimizer),
             "iter": self.current_learning_iteration, "infos": infos}            # the reference's keys (HIMR:233-240): play.py loads these
        d.update(self._extra_checkpoint_state())                                 # ... plus what it forgets; extra keys are ignored by its load()
        torch.save(d, path)

    def _portable_optimizer_state(self, opt):
        """optimizer.state_dict() as the reference's runner.load(load_optimizer=True) can adopt it: on the fused path the learning rate of the
        param groups is a CUDA tensor and `fused=True` is set (enable_device_lr); a checkpoint carries the plain float and no backend flag"""
        sd = opt.state_dict()
        groups = []
        for g in sd["param_groups"]:
            g = {k: v for k, v in g.items() if k != "fused"}
            g["lr"] = float(self.alg.learning_rate)
            groups.append(g)
        return {"state": sd["state"], "param_groups": groups}

    def _extra_checkpoint_state(self):
        out = {"learning_rate": self.alg.learning_rate}
        if hasattr(self.env, "state_dict"):
            out["env_state_dict"] = self.env.state_dict()
        if self.graphs is not None:
            out["rollout_draw_counter"] = self.graphs.get_draw_counter()
        elif self._pending_draw_counter is not None:
            out["rollout_draw_counter"] = int(self._pending_draw_counter)
        return out

    def _load_extra_checkpoint_state(self, d):
        if "learning_rate" in d:
            self.alg._set_learning_rate(d["learning_rate"])
        if "env_state_dict" in d and hasattr(self.env, "load_state_dict"):
            self.env.load_state_dict(d["env_state_dict"])
        if "rollout_draw_counter" in d:           # the sampler's Philox step word: applied now, or when enable_graphs() creates the rollout
            if self.graphs is not None:
                self.graphs.set_draw_counter(d["rollout_draw_counter"])
            else:
                self._pending_draw_counter = int(d["rollout_draw_counter"])

    def load(self, path, load_optimizer=True):
        d = torch.load(path, map_location=self.device, weights_only=False)   # holds optimizer state and plain-python extras
        self.alg.actor_critic.load_state_dict(d["model_state_dict"])
        if load_optimizer:
            self.alg.optimizer.load_state_dict(d["optimizer_state_dict"])
            self.alg.actor_critic.estimator.optimizer.load_state_dict(d["estimator_optimizer_state_dict"])
            for opt in (self.alg.optimizer, self.alg.actor_critic.estimator.optimizer):
                for g in opt.param_groups:            # a portable checkpoint (or the reference's) carries no backend flags: keep this optimiser's own.
                    for k in ("fused", "foreach", "capturable", "differentiable"):   # ASSIGN them: Adam.__setstate__ (inside load_state_dict) has
                        if k in opt.defaults:                                         # already filled the missing keys with None / False, so a
                            g[k] = opt.defaults[k]                                    # setdefault() would leave a resumed run on the per-tensor Adam
                    for k, v in opt.defaults.items():
                        g.setdefault(k, v)
            self.alg._relink_lr()
        self.current_learning_iteration = d["iter"]
        self._load_extra_checkpoint_state(d)
        return d["infos"]

    def get_inference_policy(self, device=None):
        if self.graphs is not None:
            self.graphs.flush()                    # a deferred post-step store must not see buffers an evaluation rollout is about to overwrite
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_inference


def format_time(seconds):
    s = int(seconds)
    return f"{s // 3600:02d}:{(s % 3600) // 60:02d}:{s % 60:02d}"
