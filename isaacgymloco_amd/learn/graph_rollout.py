"""Device-side rollout step for HIMOnPolicyRunner: three launches per step (policy + sample + storage, simulator kernels A and B).

The eager rollout step issues ~100 tiny torch kernels (policy / estimator / critic forward at batch N, Gaussian sampling,
log-prob, the time-out bootstrap, eleven storage copies, HIMR:110-127 + HIMP:90-118 + HST:92-106).  At N = 4096 each takes
a few microseconds on the GPU but ~10 us of host dispatch, so collection is host-bound (1.2 ms/step against 0.76 ms of GPU
work).  Here

    lsim_policy_act_post_at    :  encoder + normalise + actor + critic in one MFMA kernel (learn/fused_policy.py) whose blocks also do
                                  a = mean + std * z, log-prob, storage[t] <- (obs, critic_obs, actions, values, log-prob, mu, sigma)
                                  and the PREVIOUS step's post-step store: storage[t - 1] <- (where(done, termination_obs, critic_obs),
                                  reward + gamma * value * time_out, done) -- the critic input of step t is that next critic observation
    env.step_device(actions)      (HIP kernels A + B)
    lsim_rollout_post_at       :  the post-step store of the rollout's LAST step only (and of flush())
    [other network topologies: torch's forward, captured once in a HIP graph, then lsim_rollout_act / lsim_rollout_post with device-side counters]

With the fused policy kernel the launches are direct, so the storage row and the sampler's draw counter go by value (lsim_rollout_*_at);
when torch's forward is replayed from a captured graph they live in device memory and a one-thread kernel advances them.  Either way
nothing in the loop reads back to the host.
The sampler is the library's counter-based Philox (keyed like the simulator by (seed, rank)), not torch's generator: the
action distribution is the reference's N(mean, std); the draws are not torch's draws.
"""
import ctypes

import torch

from .. import lib


class GraphedRollout:
    def __init__(self, runner):
        self.runner, self.env, self.alg = runner, runner.env, runner.alg
        self.storage = self.alg.storage
        self.dev = self.env.obs_buf.device
        self._L = lib.load()
        N, A = self.env.num_envs, self.env.num_actions
        self.idx = torch.zeros(1, dtype=torch.long, device=self.dev)       # storage row of the current step (device resident)
        self.draws = torch.zeros(1, dtype=torch.long, device=self.dev)     # Philox step word of the action sampler
        self._draw_host = 0                                                # the same counter on the host (fused-policy path: counters go by value)
        self._pending_post = None                                          # post-step store deferred into the next policy launch
        self.actions = torch.zeros(N, A, device=self.dev)
        self.values = torch.zeros(N, 1, device=self.dev)
        self.mean = torch.zeros(N, A, device=self.dev)
        self._S = self.storage.c_struct()
        self._seed, self._rank = int(self.env.lcfg.seed), int(self.env.lcfg.rank)
        self.graph_a = None
        # preferred: the library's fused policy kernel (11 Linear + 8 ELU + glue in one launch); otherwise capture torch's forward
        from .fused_policy import PackedHimPolicy
        self.packed = PackedHimPolicy(self.alg.actor_critic) if PackedHimPolicy.supported(self.alg.actor_critic) else None
        self.by_value = self.packed is not None         # captured graphs need device-side counters; the direct launches do not
        if self.packed is None:
            self._capture()
        self._weights_stale = True
        self.storage.step = 0
        # the deferred post-step store keeps references to live simulator buffers: anything else that steps / resets the env (an evaluation
        # rollout, env.reset()) or reads the storage in the middle of a rollout flushes it first
        self.env.before_external_step = self.flush
        self.storage.pending_store = lambda: self._pending_post is not None

    # ---- HIMP:90-103 written against static tensors; the elementwise tail is one HIP kernel --------------------------
    def _act(self):
        env, ac = self.env, self.alg.actor_critic
        if self.by_value:           # networks + sample + storage row (+ the previous step's post-step store) in one launch
            prev, self._pending_post = self._pending_post, None
            self.packed.forward_act(self._S, self.storage.step, self._draw_host, env.obs_buf, env.privileged_obs_buf, ac.std, self._seed, self._rank,
                                    self.mean, self.values, self.actions, prev=prev)
            return
        if self.packed is not None:
            self.packed.forward(env.obs_buf, env.privileged_obs_buf, self.mean, self.values)
        else:
            ac.update_distribution(env.obs_buf)
            self.mean.copy_(ac.action_mean)
            self.values.copy_(ac.evaluate(env.privileged_obs_buf))
        s = torch.cuda.current_stream(self.dev).cuda_stream
        lib.check(self._L.lsim_rollout_act(ctypes.byref(self._S), self.idx.data_ptr(), self.draws.data_ptr(), self.mean.data_ptr(),
                                           ac.std.data_ptr(), self.values.data_ptr(), env.obs_buf.data_ptr(),
                                           env.privileged_obs_buf.data_ptr(), self._seed, self._rank, self.actions.data_ptr(), s),
                  what="lsim_rollout_act")

    # ---- HIMR:119-121 + HIMP:105-118 + HST:92-106 ------------------------------------------------------------------------
    def _post(self, rewards=None):
        env = self.env
        to = env.extras.get("time_outs")
        rewards = env.rew_buf if rewards is None else rewards
        s = torch.cuda.current_stream(self.dev).cuda_stream
        if self.by_value:
            self._draw_host += 1
            if self.storage.step + 1 < self.storage.num_transitions_per_env:
                # not the rollout's last step: the next step's policy launch does this store (its critic blocks stage the same privileged
                # observation, and the buffers read here stay untouched until the simulator steps again); flush() forces it
                self._pending_post = (int(self.storage.step), env.reset_buf, to, rewards, env.termination_privileged_obs_buf, float(self.alg.gamma))
                return
            self._post_now(int(self.storage.step), env.reset_buf, to, rewards, env.termination_privileged_obs_buf, float(self.alg.gamma))
            return
        lib.check(self._L.lsim_rollout_post(ctypes.byref(self._S), self.idx.data_ptr(), self.draws.data_ptr(), env.reset_buf.data_ptr(),
                                            to.data_ptr() if to is not None else None, rewards.data_ptr(), self.values.data_ptr(),
                                            env.privileged_obs_buf.data_ptr(), env.termination_privileged_obs_buf.data_ptr(),
                                            float(self.alg.gamma), s), what="lsim_rollout_post")

    def _capture(self):
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(3):          # warm-up on a side stream (allocator / lazy init) before capture
                self._act()
                self._post()
            self.idx.zero_()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph_a):
            self._act()
        self.idx.zero_()
        self.storage.step = 0

    def step(self):
        """one rollout step: networks + fused sample/store (two launches, or one graph replay), simulator step, fused post-step store"""
        self._sync_weights()
        with lib.roctx_range("policy_forward+sample+store"):
            if self.packed is not None:
                self._act()
            else:
                self.graph_a.replay()
        self.env.step_device(self.actions)
        with lib.roctx_range("rollout_post"):
            self._post()
        self.storage.step += 1

    def _post_now(self, step, dones, to, rewards, term, gamma):
        lib.check(self._L.lsim_rollout_post_at(ctypes.byref(self._S), step, dones.data_ptr(), to.data_ptr() if to is not None else None,
                                               rewards.data_ptr(), self.values.data_ptr(), self.env.privileged_obs_buf.data_ptr(), term.data_ptr(),
                                               gamma, torch.cuda.current_stream(self.dev).cuda_stream), what="lsim_rollout_post_at")

    def flush(self):
        """write a post-step store that is still waiting for the next policy launch (callers that read the storage in the middle of a rollout)"""
        if getattr(self, "_pending_post", None) is not None:
            p, self._pending_post = self._pending_post, None
            self._post_now(*p)

    def get_draw_counter(self):
        """Philox step word of the action sampler (checkpointed by the runner)"""
        return self._draw_host if self.by_value else int(self.draws.item())

    def set_draw_counter(self, v):
        self._draw_host = int(v)
        self.draws.fill_(int(v))

    def _sync_weights(self):
        if self._weights_stale and self.packed is not None:
            self.packed.refresh()
        self._weights_stale = False

    def end_iteration(self):
        """after compute_returns (callers rewind before or after update()): rewind the device-side step index.  The packed weights of
        the fused policy kernel are re-copied lazily at the first step of the next rollout, i.e. always AFTER the optimiser steps."""
        self.flush()
        self.idx.zero_()
        self._weights_stale = True


class HybridFusedRollout(GraphedRollout):
    """The same device-side rollout step for HybridPolicyRunner (AMP, HYBR:118-152): between the simulator step and the storage write
    the task reward is blended with the discriminator's style reward on (amp_obs, next_amp_obs) -- terminal AMP states patched in for
    resetting envs (HYBR:136-140) -- and the pair goes into the AMP replay buffer (HYBP:121-124).  ONE launch: lsim_amp_step
    (learn/fused_amp.py; csrc/ls_amp.h) normalises, runs the discriminator on the matrix cores, writes the blended reward row the next policy
    launch stores, the replay ring rows and the next step's AMP observation.  Discriminator shapes the kernel does not take (or
    LSIM_AMP_FUSED_STEP=0, the A/B switch) run the same statements as torch ops."""

    def __init__(self, runner):
        super().__init__(runner)
        import os
        from .fused_amp import PackedAmpDisc
        N = self.env.num_envs
        first = self.env.get_amp_observations()
        self._amp_bufs = [first.clone(), torch.empty_like(first)]      # ping-pong: the kernel reads one as amp_obs and fills the other with next_amp_obs
        self._cur = 0
        self.rewards = torch.zeros(N, device=self.dev)
        self.disc_out = torch.zeros(N, device=self.dev)
        disc = self.alg.discriminator
        ok = os.environ.get("LSIM_AMP_FUSED_STEP") != "0" and PackedAmpDisc.supported(disc) and self.alg.amp_storage.buffer_size >= N
        self.packed_disc = PackedAmpDisc(disc, self.alg.amp_normalizer, N) if ok else None
        self._amp_stale = False

    def flush(self):
        """(also the env's before_external_step hook) something else is about to step / reset the env: the AMP observation this rollout carries from step to
        step is then no longer the env's current one -- the next rollout step re-reads it (the eager runner does the same at the start of learn(), HYBR:118)"""
        super().flush()
        self._amp_stale = True

    @property
    def _amp_obs(self):
        return self._amp_bufs[self._cur]

    def _sync_weights(self):
        if self._weights_stale and self.packed_disc is not None:
            self.packed_disc.refresh()
        super()._sync_weights()

    def step(self):
        env, alg = self.env, self.alg
        amp_obs = self._amp_bufs[self._cur]
        if self._amp_stale:           # (after an ordinary end of iteration the two already hold the same values: the copy changes nothing)
            amp_obs.copy_(env.get_amp_observations())
            self._amp_stale = False
        self._sync_weights()
        if self.packed is not None:
            self._act()
        else:
            self.graph_a.replay()
        env.step_device(self.actions)
        if self.packed_disc is not None:
            with lib.roctx_range("amp_step"):
                self.packed_disc.step(amp_obs, env.get_amp_observations(), env.reset_buf, env.terminal_amp_states_buf, env.rew_buf, self.rewards,
                                      disc_out=self.disc_out, carry=self._amp_bufs[self._cur ^ 1], replay=alg.amp_storage)
        else:
            next_amp = env.get_amp_observations()
            next_with_term = torch.where(env.reset_buf.unsqueeze(1), env.terminal_amp_states_buf, next_amp)
            self.rewards.copy_(alg.discriminator.predict_amp_reward(amp_obs, next_with_term, env.rew_buf, normalizer=alg.amp_normalizer)[0])
            alg.amp_storage.insert(amp_obs, next_with_term)
            self._amp_bufs[self._cur ^ 1].copy_(next_amp)
        self._cur ^= 1
        self._post(self.rewards)
        self.storage.step += 1
