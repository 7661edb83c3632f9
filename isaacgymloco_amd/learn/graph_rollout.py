"""hipGraph-captured rollout step for HIMOnPolicyRunner (launch-bound inner loop -> two graph replays + one env launch).

The eager rollout step issues ~100 tiny torch kernels (policy / estimator / critic forward at batch N, Gaussian sampling,
log-prob, the time-out bootstrap, eleven storage copies, HIMR:110-127 + HIMP:90-118 + HST:92-106).  At N = 4096 each takes
a few microseconds on the GPU but ~10 us of host dispatch, so collection is host-bound (1.2 ms/step against 0.76 ms of GPU
work).  Here the same torch ops are captured once into two HIP graphs that read the simulator's live buffers (static
addresses inside the arena) and write the rollout storage at a device-resident step index:

    graph A:  act(obs, critic_obs) -> actions, values, log-prob, mu, sigma;   storage[idx] <- (obs, critic_obs, actions, ...)
    env.step_device(actions)                                                  (HIP kernels A + B, not captured: per-call args)
    graph B:  next_critic_obs = where(done, termination_obs, critic_obs);  reward += gamma * value * time_out;
              storage[idx] <- (next_critic_obs, reward, done);  idx += 1

Numerics are the eager path's (same ops, same order); only the Philox offsets of the sampler differ.
"""
import torch


class GraphedRollout:
    def __init__(self, runner):
        self.runner, self.env, self.alg = runner, runner.env, runner.alg
        self.storage = self.alg.storage
        self.dev = self.env.obs_buf.device
        self.idx = torch.zeros(1, dtype=torch.long, device=self.dev)
        self.one = torch.ones(1, dtype=torch.long, device=self.dev)
        N = self.env.num_envs
        self.actions = torch.zeros(N, self.env.num_actions, device=self.dev)
        self.values = torch.zeros(N, 1, device=self.dev)
        self.graph_a = self.graph_b = None
        self._capture()

    # ---- the two halves of HIMR:110-127, written against static tensors -------------------------------------------
    def _act(self):
        env, ac, st, i = self.env, self.alg.actor_critic, self.storage, self.idx
        obs, critic_obs = env.obs_buf, env.privileged_obs_buf
        # ac.act(obs) without torch.normal(mean, std): that overload checks std >= 0 on the host (a sync, illegal while
        # capturing); N(0,1) * std + mean is what it computes internally
        ac.update_distribution(obs)
        mean, std = ac.action_mean, ac.action_std
        actions = torch.randn_like(mean) * std + mean
        values = ac.evaluate(critic_obs)
        logp = ac.get_actions_log_prob(actions)
        self.actions.copy_(actions)
        self.values.copy_(values)
        st.observations.index_copy_(0, i, obs.unsqueeze(0))
        st.privileged_observations.index_copy_(0, i, critic_obs.unsqueeze(0))
        st.actions.index_copy_(0, i, actions.unsqueeze(0))
        st.values.index_copy_(0, i, values.unsqueeze(0))
        st.actions_log_prob.index_copy_(0, i, logp.view(1, -1, 1))
        st.mu.index_copy_(0, i, ac.action_mean.unsqueeze(0))
        st.sigma.index_copy_(0, i, ac.action_std.unsqueeze(0))

    def _post(self):
        env, st, i = self.env, self.storage, self.idx
        dones = env.reset_buf
        next_critic = torch.where(dones.unsqueeze(1), env.termination_privileged_obs_buf, env.privileged_obs_buf)
        rewards = env.rew_buf.clone()
        if "time_outs" in env.extras:   # HIMP:110-111
            rewards += self.alg.gamma * torch.squeeze(self.values * env.extras["time_outs"].unsqueeze(1), 1)
        st.next_privileged_observations.index_copy_(0, i, next_critic.unsqueeze(0))
        st.rewards.index_copy_(0, i, rewards.view(1, -1, 1))
        st.dones.index_copy_(0, i, dones.view(1, -1, 1).to(torch.uint8))
        i.add_(self.one)

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
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.no_grad():
            with torch.cuda.graph(self.graph_a):
                self._act()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
                self._post()
        self.idx.zero_()
        self.storage.step = 0

    def step(self):
        """one rollout step: graph A, simulator step, graph B"""
        self.graph_a.replay()
        self.env.step_device(self.actions)
        self.graph_b.replay()
        self.storage.step += 1

    def end_iteration(self):
        """after compute_returns/update (which call storage.clear()): rewind the device-side step index"""
        self.idx.zero_()
