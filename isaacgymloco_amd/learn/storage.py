"""Rollout storage for the HIM PPO learner: (T, N, .) device tensors, GAE(lambda) sweep, minibatch sampling.
Same contract as rsl_rl.storage.HIMRolloutStorage (HST:36-177): field names, add_transitions, compute_returns,
mini_batch_generator (one randperm reused for every epoch, HST:140-164)."""
import torch


class Transition:
    __slots__ = ("observations", "critic_observations", "actions", "rewards", "dones", "values", "actions_log_prob",
                 "action_mean", "action_sigma", "next_critic_observations")

    def __init__(self):
        self.clear()

    def clear(self):
        for k in self.__slots__:
            setattr(self, k, None)


class HIMRolloutStorage:
    Transition = Transition

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cpu"):
        T, N = num_transitions_per_env, num_envs
        self.device = device
        self.obs_shape, self.privileged_obs_shape, self.actions_shape = obs_shape, privileged_obs_shape, actions_shape
        self.num_transitions_per_env, self.num_envs = T, N

        def z(*shape, dtype=torch.float32):
            return torch.zeros(T, N, *shape, device=device, dtype=dtype)
        self.observations = z(*obs_shape)
        if privileged_obs_shape[0] is not None:
            self.privileged_observations = z(*privileged_obs_shape)
            self.next_privileged_observations = z(*privileged_obs_shape)
        else:
            self.privileged_observations = self.next_privileged_observations = None
        self.rewards, self.actions = z(1), z(*actions_shape)
        self.dones = z(1, dtype=torch.uint8)
        self.actions_log_prob, self.values, self.returns, self.advantages = z(1), z(1), z(1), z(1)
        self.mu, self.sigma = z(*actions_shape), z(*actions_shape)
        self.step = 0
        self.advantage_sync = None   # data-parallel hook: callable(sum, sumsq, count) -> reduced triple

    def add_transitions(self, t):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        i = self.step
        self.observations[i].copy_(t.observations)
        if self.privileged_observations is not None:
            self.privileged_observations[i].copy_(t.critic_observations)
            self.next_privileged_observations[i].copy_(t.next_critic_observations)
        self.actions[i].copy_(t.actions)
        self.rewards[i].copy_(t.rewards.view(-1, 1))
        self.dones[i].copy_(t.dones.view(-1, 1))
        self.values[i].copy_(t.values)
        self.actions_log_prob[i].copy_(t.actions_log_prob.view(-1, 1))
        self.mu[i].copy_(t.action_mean)
        self.sigma[i].copy_(t.action_sigma)
        self.step += 1

    def _no_pending_store(self):
        """a device-side rollout may still owe this storage the post-step row of its latest step (learn/graph_rollout.py: flush())"""
        pending = getattr(self, "pending_store", None)
        assert pending is None or not pending(), "rollout storage read / rewound while a deferred post-step store is pending: call flush()"

    def clear(self):
        self._no_pending_store()
        self.step = 0

    def c_struct(self):
        """lsim_rollout_storage (include/lsim.h) over this storage's device tensors, for the fused HIP rollout kernels"""
        from .. import abi
        if getattr(self, "_c_struct", None) is None:
            S = abi.LsimRolloutStorage()
            for name in ("observations", "privileged_observations", "next_privileged_observations", "actions", "values",
                         "actions_log_prob", "mu", "sigma", "rewards", "dones"):
                t = getattr(self, name)
                assert t is not None and t.is_cuda and t.is_contiguous(), name
                setattr(S, name, t.data_ptr())
            assert self.dones.dtype == torch.uint8 and self.rewards.dtype == torch.float32
            S.num_steps, S.num_envs = self.num_transitions_per_env, self.num_envs
            S.num_obs, S.num_priv_obs = self.observations.shape[2], self.privileged_observations.shape[2]
            S.num_actions = self.actions.shape[2]
            self._c_struct = S
        return self._c_struct

    def compute_returns(self, last_values, gamma, lam):
        """GAE(lambda) reverse sweep (HST:113-123) and advantage normalisation over the whole batch (HST:126-127)."""
        T = self.num_transitions_per_env
        self._no_pending_store()
        if self.values.is_cuda and self.privileged_observations is not None:
            # one HIP launch (one thread per env walks the T steps) instead of ~8 torch kernels per step
            import ctypes
            from .. import lib
            a = torch.empty_like(self.returns)
            lv = last_values.detach().contiguous()
            lib.check(lib.load().lsim_rollout_gae(ctypes.byref(self.c_struct()), lv.data_ptr(), float(gamma), float(lam),
                                                  self.returns.data_ptr(), a.data_ptr(),
                                                  torch.cuda.current_stream(self.values.device).cuda_stream), what="lsim_rollout_gae")
        else:
            not_done = 1.0 - self.dones.float()
            adv = torch.zeros_like(last_values)
            nxt = last_values
            for s in range(T - 1, -1, -1):
                delta = self.rewards[s] + not_done[s] * gamma * nxt - self.values[s]
                adv = delta + not_done[s] * gamma * lam * adv
                self.returns[s] = adv + self.values[s]
                nxt = self.values[s]
            a = self.returns - self.values
        if self.advantage_sync is None:
            self.advantages = (a - a.mean()) / (a.std() + 1e-8)
        else:   # global statistics over all ranks (unbiased std, like Tensor.std())
            s1, s2, n = self.advantage_sync(a.sum(), (a * a).sum(), torch.tensor(float(a.numel()), device=a.device))
            mean = s1 / n
            var = (s2 - n * mean * mean) / (n - 1.0)
            self.advantages = (a - mean) / (var.clamp_min(0).sqrt() + 1e-8)

    def get_statistics(self):
        done = self.dones.clone()
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        return (idx[1:] - idx[:-1]).float().mean(), self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches, num_epochs=8):
        """HST:129-177.  ALIASING CONTRACT (GPU): the minibatch tensors are slices of persistent per-storage buffers (_shuffle), not fresh tensors -- they
        are overwritten by the NEXT call's shuffle, so a caller that keeps one across update() calls must clone it, and only ONE generator of a
        storage may be live at a time (a second one re-shuffles under the first: asserted on every yield).  release_shuffle_buffers() frees them."""
        self._no_pending_store()
        gen_id = self._shuffle_generation = getattr(self, "_shuffle_generation", 0) + 1
        batch = self.num_envs * self.num_transitions_per_env
        mb = batch // num_mini_batches
        perm = torch.randperm(num_mini_batches * mb, requires_grad=False, device=self.device)
        obs = self.observations.flatten(0, 1)
        if self.privileged_observations is not None:
            critic, next_critic = self.privileged_observations.flatten(0, 1), self.next_privileged_observations.flatten(0, 1)
        else:
            critic = next_critic = obs
        fields = (obs, critic, self.actions.flatten(0, 1), next_critic, self.values.flatten(0, 1), self.advantages.flatten(0, 1),
                  self.returns.flatten(0, 1), self.actions_log_prob.flatten(0, 1), self.mu.flatten(0, 1), self.sigma.flatten(0, 1))
        # The reference draws ONE permutation and reuses it for every epoch (HST:140, HST:159-164), so minibatch i holds the same
        # rows in all epochs: gather the whole batch through the permutation once and hand out contiguous slices, instead of
        # re-gathering ~800 floats per sample for each of the epochs x minibatches (same values, 1/num_epochs of the gather traffic).
        shuffled = self._shuffle(fields, perm)
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                if self._shuffle_generation != gen_id:
                    raise RuntimeError("a second mini_batch_generator of this storage re-shuffled the buffers this one hands out (one live generator per storage)")
                yield tuple(f[i * mb:(i + 1) * mb] for f in shuffled)

    def release_shuffle_buffers(self):
        """free the persistent shuffle destinations (~0.4 GB at 4096 x 100): after the last update of a run, before an evaluation-only phase"""
        self._shuffled = None

    def _shuffle(self, fields, perm):
        """every field gathered through the permutation.  On the GPU the destinations are PERSISTENT buffers (allocated at the first call):
        minibatch i of every update lives at the same addresses and the update stops allocating 0.4 GB per call.
        Round 6: the rows of 2-D fields whose width is not a multiple of four floats (the 270-wide observation history, the 238-wide privileged
        observations) are laid out 16-byte aligned -- row pitch rounded up to 4 floats, the padding zero and never written -- and handed out as
        [B, width] VIEWS of the padded buffer: the first layers of the networks then read aligned rows and join the library's fused Linear + ELU
        forward (fused_linear.linear_elu_forward with a zero-padded weight copy) and the 16-byte form of the weight-gradient kernel.  (Round 5
        measured padded rows with the BLAS forward kept: -1 %, the BLAS kernels are TunableOp-selected per leading dimension.)
        LSIM_PAD_SHUFFLED=0: contiguous rows (A/B switch)."""
        if not fields[0].is_cuda:
            return tuple(_gather_rows(f, perm) for f in fields)
        import os
        pad_on = os.environ.get("LSIM_PAD_SHUFFLED", "0") == "1"
        pitch = lambda f: (f.shape[1] + 3) // 4 * 4 if (pad_on and f.dim() == 2 and f.shape[1] % 4 != 0 and f.shape[1] >= 64 and f.element_size() == 4) else None
        bufs = getattr(self, "_shuffled", None)
        ok = bufs is not None and len(bufs) == len(fields) and all(
            b.shape[0] == perm.numel() and b.dtype == f.dtype and b.device == f.device and
            (b.shape[1:] == f.shape[1:] if pitch(f) is None else (b.dim() == 2 and b.shape[1] == pitch(f))) for b, f in zip(bufs, fields))
        if not ok:
            bufs = self._shuffled = tuple(torch.empty((perm.numel(),) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device) if pitch(f) is None
                                          else torch.zeros(perm.numel(), pitch(f), dtype=f.dtype, device=f.device) for f in fields)
        out = []
        for f, b in zip(fields, bufs):
            if pitch(f) is None:
                out.append(_gather_rows(f, perm, out=b))
            else:
                out.append(_gather_rows(f, perm, out=b[:, :f.shape[1]]))          # rows pitch(f) floats apart; columns beyond the width stay zero
        return tuple(out)


def _gather_rows(f, perm, out=None):
    """f[perm] for a contiguous tensor of 4-byte elements on the GPU through lsim_gather_rows (rows at copy bandwidth); anything else: f[perm].
    `out`: destination to fill (and return) instead of a fresh tensor; a 2-D `out` may have rows further apart than its width"""
    if not (f.is_cuda and f.is_contiguous() and f.element_size() == 4 and f.dim() >= 1 and perm.dtype == torch.int64 and perm.is_contiguous()):
        if out is None:
            return f[perm]
        torch.index_select(f, 0, perm, out=out)
        return out
    from .. import lib
    cols = f[0].numel() if f.dim() > 1 else 1
    if out is None:
        out = torch.empty((perm.numel(),) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device)
    if out.is_contiguous():
        ld = cols
    else:
        assert out.dim() == 2 and out.stride(1) == 1 and out.stride(0) >= cols, "destination rows must be dense"
        ld = out.stride(0)
    lib.check(lib.load().lsim_gather_rows_ld(f.data_ptr(), cols, perm.data_ptr(), perm.numel(), out.data_ptr(), ld, torch.cuda.current_stream(f.device).cuda_stream),
              what="lsim_gather_rows_ld")
    return out
