"""AMP training path (BASELINE config 4): HybridPPO + HybridPolicyRunner, the live AMP pair of the reference
(rsl_rl/algorithms/hybrid_ppo.py:39-306, rsl_rl/runners/hybrid_runner.py:54-366; SURVEY.md headline 5).

Per rollout step the task reward is replaced by the discriminator's style reward lerped with the task reward
(HYBR:194-195, DISC:55-72); terminal AMP states of resetting envs patch the next AMP observation (HYBR:191-192); policy
transition pairs go to a 1 M-entry replay buffer (HYBP:144-145).  The update adds the LSGAN loss and the gradient penalty
(lambda 10) to the PPO loss in ONE Adam over {actor_critic, discriminator trunk (wd 1e-3), head (wd 1e-1)} (HYBP:86-92,
HYBP:252-273), clamps std >= min_std (HYBP:276-277) and feeds the running normaliser with the *normalised* policy /
expert states exactly like the reference does (HYBP:236-241, HYBP:279-281).
"""
import time
import torch
import torch.nn as nn

from .amp import AMPDiscriminator, AMPLoader, Normalizer, ReplayBuffer, default_motion_files
from .him_ppo import HIMPPO
from .modules import HIMActorCritic
from .runner import HIMOnPolicyRunner
from .storage import HIMRolloutStorage


class HybridPPO(HIMPPO):
    def __init__(self, actor_critic, discriminator, amp_data, amp_normalizer, amp_replay_buffer_size=100000, min_std=None,
                 device="cpu", dist_ctx=None, **kw):
        super().__init__(actor_critic, device=device, dist_ctx=dist_ctx, **kw)
        self.min_std = min_std
        self.discriminator = discriminator.to(device)
        self.amp_transition = HIMRolloutStorage.Transition()
        self.amp_storage = ReplayBuffer(discriminator.input_dim // 2, amp_replay_buffer_size, device)
        self.amp_data, self.amp_normalizer = amp_data, amp_normalizer
        self.optimizer = torch.optim.Adam([
            {"params": self.actor_critic.parameters(), "name": "actor_critic"},
            {"params": self.discriminator.trunk.parameters(), "weight_decay": 10e-4, "name": "amp_trunk"},
            {"params": self.discriminator.amp_linear.parameters(), "weight_decay": 10e-2, "name": "amp_head"}], lr=self.learning_rate)
        if dist_ctx is not None and dist_ctx.enabled:
            dist_ctx.broadcast_module(self.discriminator)
            if amp_normalizer is not None:
                amp_normalizer.moment_sync = lambda a, b, c: tuple(t.to(torch.float64) for t in dist_ctx.sum_triple_vec(a, b, c))

    def act(self, obs, critic_obs, amp_obs):
        self.amp_transition.observations = amp_obs
        return super().act(obs, critic_obs)

    def process_env_step(self, rewards, dones, infos, amp_obs, next_critic_obs):
        self.amp_storage.insert(self.amp_transition.observations, amp_obs)
        self.amp_transition.clear()
        super().process_env_step(rewards, dones, infos, next_critic_obs)

    def update(self):
        ac, disc, dev = self.actor_critic, self.discriminator, self.device
        n_updates = self.num_learning_epochs * self.num_mini_batches
        mb = self.storage.num_envs * self.storage.num_transitions_per_env // self.num_mini_batches
        sums = torch.zeros(6, device=dev)
        est = swap = None
        t_enqueue = time.perf_counter()
        self._grad_arena()
        gens = zip(self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs),
                   self.amp_storage.feed_forward_generator(n_updates, mb), self.amp_data.feed_forward_generator(n_updates, mb))
        for sample, (pol_s, pol_ns), (exp_s_raw, exp_ns_raw) in gens:
            obs, critic_obs, actions, next_critic_obs, target_values, advantages, returns, old_logp, old_mu, old_sigma = sample
            ac.estimator.prime(obs)        # one encoder forward serves the policy features and the estimator loss below
            # the reference calls act() here (HIMP:141) and throws the sample away; torch.normal(mean, std) validates std >= 0 with a
            # host read-back, i.e. one pipeline drain per minibatch on the GPU: only the distribution is needed
            if obs.is_cuda:
                ac.update_distribution(obs)
            else:
                ac.act(obs)
            value = ac.evaluate(critic_obs)
            mu, sigma = ac.action_mean, ac.action_std
            ppo_loss, surrogate_loss, value_loss, kl_mean = self._ppo_loss(ac, mu, sigma, value, actions, old_logp, advantages, returns, target_values,
                                                                           old_mu, old_sigma)
            adaptive = self.desired_kl is not None and self.schedule == "adaptive"
            dist_on = self.dist_ctx is not None and self.dist_ctx.enabled
            if not dist_on:
                if adaptive:
                    self._adapt_lr(mu, sigma, old_mu, old_sigma, kl_mean)
                est, swap = ac.estimator.update(obs, next_critic_obs, lr=None if self._lr_t is not None else self.learning_rate)
            # normalise + concatenate the sampled pairs (HYBP:247-251, DISC:57): three launches on the GPU (AMPDiscriminator.pair_inputs)
            expert_in, policy_in, expert_raw, exp_s, pol_s = disc.pair_inputs(exp_s_raw, exp_ns_raw, pol_s, pol_ns, self.amp_normalizer)
            # (exp_s, pol_s: the NORMALISED states -- what the reference feeds its normaliser at the end of the minibatch, HYBP:279-281)
            amp_loss, policy_d_mean, expert_d_mean = disc.lsgan_loss(expert_in, policy_in)   # HYBP:252-261
            grad_pen = disc.compute_grad_pen(exp_s_raw, exp_ns_raw, lambda_=10, pair=expert_raw)     # on the un-normalised expert pair (HYBP:262-263)
            loss = ppo_loss + amp_loss + grad_pen
            if dist_on:      # two collectives per minibatch, the estimator's in flight during this backward (him_ppo.py)
                est, swap = self._step_minibatch_data_parallel(ac, obs, next_critic_obs, loss, mu, sigma, old_mu, old_sigma, kl_mean, adaptive,
                                                               more_params=list(disc.parameters()))
            else:
                self.optimizer.zero_grad()
                from . import fused_linear as FL
                FL.grad_cycle()
                with FL.deferred_wgrad_reduce():
                    loss.backward()
                if FL._arena is not None:
                    FL._arena.bucket("ppo", [p for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]).adopt()
                self._clip_and_step(self.optimizer, ac.parameters(), self.max_grad_norm)     # HYBP:270-273: clipping over the actor-critic only
            if self.min_std is not None:
                ac.std.data = ac.std.data.clamp(min=self.min_std)
            if self.amp_normalizer is not None:
                self.amp_normalizer.update(pol_s)
                self.amp_normalizer.update(exp_s)
            sums += torch.stack((value_loss.detach(), surrogate_loss.detach(), amp_loss.detach(), grad_pen.detach(),
                                 policy_d_mean, expert_d_mean))
        self.update_enqueue_s = time.perf_counter() - t_enqueue      # (him_ppo.HIMPPO.update: host time to enqueue, no read-back before here)
        if self._lr_t is not None:
            self.learning_rate = float(self._lr_t)
        s = (sums / n_updates).tolist()
        self.storage.clear()
        return s[0], s[1], float(est), float(swap), s[2], s[3], s[4], s[5]


class HybridPolicyRunner(HIMOnPolicyRunner):
    def __init__(self, env, train_cfg, log_dir=None, device="cpu", fast=None):
        self.cfg, self.alg_cfg, self.policy_cfg = train_cfg["runner"], dict(train_cfg["algorithm"]), train_cfg["policy"]
        self.device, self.env = device, env
        from .him_ppo import DistCtx
        self.dist_ctx = DistCtx()
        num_critic_obs = env.num_privileged_obs if env.num_privileged_obs is not None else env.num_obs
        self.num_actor_obs, self.num_critic_obs = env.num_obs, num_critic_obs
        actor_critic = HIMActorCritic(env.num_obs, num_critic_obs, env.num_one_step_obs, env.num_actions, **self.policy_cfg).to(device)
        files = self.cfg.get("amp_motion_files") or default_motion_files()
        if not all(str(f).startswith("/") or __import__("os").path.exists(f) for f in files):
            files = default_motion_files()
        amp_data = AMPLoader(device, time_between_frames=env.dt, preload_transitions=True,
                             num_preload_transitions=self.cfg["amp_num_preload_transitions"], motion_files=files)
        amp_normalizer = Normalizer(amp_data.observation_dim, device=device)
        discriminator = AMPDiscriminator(amp_data.observation_dim * 2, self.cfg["amp_reward_coef"], self.cfg["amp_discr_hidden_dims"],
                                         device, self.cfg["amp_task_reward_lerp"]).to(device)
        min_std = torch.tensor(self.cfg["min_normalized_std"], device=device) * torch.abs(env.dof_pos_limits[:, 1] - env.dof_pos_limits[:, 0]).to(device)
        self.alg = HybridPPO(actor_critic, discriminator, amp_data, amp_normalizer, device=device, min_std=min_std,
                             dist_ctx=self.dist_ctx, **self.alg_cfg)
        self.num_steps_per_env, self.save_interval = self.cfg["num_steps_per_env"], self.cfg["save_interval"]
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [env.num_obs], [env.num_privileged_obs], [env.num_actions])
        self.log_dir, self.writer = log_dir, None
        self.tot_timesteps, self.tot_time, self.current_learning_iteration = 0, 0.0, 0
        self.fast = hasattr(env, "step_device") if fast is None else fast
        self.last_perf = {}
        env.reset()
        self._amp_obs = None

    def _rollout_step(self, obs, critic_obs):
        env, alg = self.env, self.alg
        if self._amp_obs is None:
            self._amp_obs = env.get_amp_observations().to(self.device).clone()
        amp_obs = self._amp_obs
        actions = alg.act(obs, critic_obs, amp_obs)
        if self.fast:
            obs, priv, rewards, dones = env.step_device(actions)
            infos = env.extras
            obs = obs.clone()
            critic_obs = (priv if priv is not None else obs).clone()
            next_amp_obs = env.get_amp_observations().clone()
            mask = dones.unsqueeze(1)
            next_amp_with_term = torch.where(mask, env.terminal_amp_states_buf, next_amp_obs)
            next_critic_obs = torch.where(mask, env.termination_privileged_obs_buf, critic_obs)
        else:
            obs, priv, rewards, dones, infos, ids, term_priv, term_amp = env.step(actions)
            next_amp_obs = env.get_amp_observations().to(self.device).clone()
            critic_obs = priv if priv is not None else obs
            obs, critic_obs, rewards, dones = obs.to(self.device), critic_obs.to(self.device), rewards.to(self.device), dones.to(self.device)
            next_amp_with_term = torch.clone(next_amp_obs)
            next_amp_with_term[ids] = term_amp
            next_critic_obs = critic_obs.clone().detach()
            next_critic_obs[ids.to(self.device)] = term_priv.to(self.device).clone().detach()
        rewards = alg.discriminator.predict_amp_reward(amp_obs, next_amp_with_term, rewards, normalizer=alg.amp_normalizer)[0]
        self._amp_obs = next_amp_obs
        alg.process_env_step(rewards, dones, infos, next_amp_with_term, next_critic_obs)
        return obs, critic_obs, rewards, dones, infos

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        self._amp_obs = None
        return super().learn(num_learning_iterations, init_at_random_ep_len)

    def _make_fused_rollout(self):
        from .graph_rollout import HybridFusedRollout
        return HybridFusedRollout(self)

    # the reference's HybridPolicyRunner.save (HYBR:334-345) drops the discriminator and the AMP normaliser: a resumed run restarts them
    def _extra_checkpoint_state(self):
        out = super()._extra_checkpoint_state()
        nz = self.alg.amp_normalizer
        out["discriminator_state_dict"] = self.alg.discriminator.state_dict()
        out["amp_normalizer"] = {"mean": nz._mean.cpu(), "var": nz._var.cpu(), "count": nz._count.cpu()}
        return out

    def _load_extra_checkpoint_state(self, d):
        super()._load_extra_checkpoint_state(d)
        if "discriminator_state_dict" in d:
            self.alg.discriminator.load_state_dict(d["discriminator_state_dict"])
        if "amp_normalizer" in d:
            nz, st = self.alg.amp_normalizer, d["amp_normalizer"]
            nz._mean, nz._var, nz._count = st["mean"].to(nz._mean.device), st["var"].to(nz._var.device), st["count"].to(nz._count.device)
