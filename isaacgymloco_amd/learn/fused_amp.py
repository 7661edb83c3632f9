"""The AMP rollout step through the library's fused kernel (include/lsim.h, lsim_amp_step): terminal-state patch, running-moment
normalisation, discriminator 2 D -> H1 -> H2 -> 1 on fp32 MFMA, style reward lerped with the task reward, replay-ring insert and the carry
of the next AMP observation in ONE launch -- what HybridPolicyRunner does between env.step() and process_env_step() (HYBR:183-200,
DISC:55-72, UT:124-130, RB:52-68).

`PackedAmpDisc` keeps the discriminator's parameters in the kernel's layout (zero padded to multiples of 16, trunk weights cut into
16 x 16 blocks, like fused_policy.PackedHimPolicy) at fixed device addresses; `refresh()` re-copies them after a policy update.
"""
import ctypes

import torch
import torch.nn as nn

from .. import abi, lib


def _pad16(v):
    return (v + 15) // 16 * 16


class PackedAmpDisc:
    @staticmethod
    def supported(disc):
        mods = list(disc.trunk)
        if len(mods) != 4 or not (isinstance(mods[0], nn.Linear) and isinstance(mods[1], nn.ReLU) and isinstance(mods[2], nn.Linear) and isinstance(mods[3], nn.ReLU)):
            return False
        l0, l1 = mods[0], mods[2]
        if l0.bias is None or l1.bias is None or disc.amp_linear.bias is None or disc.amp_linear.out_features != 1:
            return False
        return (l0.in_features % 2 == 0 and l0.in_features // 2 <= 32 and _pad16(l0.out_features) <= 1024 and _pad16(l1.out_features) <= 1024
                and l0.weight.is_cuda and l0.weight.dtype == torch.float32)

    def __init__(self, disc, normalizer, num_envs):
        assert self.supported(disc)
        self.disc, self.normalizer = disc, normalizer
        self.dev = disc.trunk[0].weight.device
        self._L = lib.load()
        self.layers = [disc.trunk[0], disc.trunk[2]]
        self._rows = [torch.zeros(_pad16(l.out_features), _pad16(l.in_features), device=self.dev) for l in self.layers]
        self.w = [torch.zeros(r.shape[0] // 16, r.shape[1] // 16, 16, 16, device=self.dev) for r in self._rows]
        self.b = [torch.zeros(r.shape[0], device=self.dev) for r in self._rows]
        self.head_w = torch.zeros(self._rows[1].shape[0], device=self.dev)
        self.head_b = torch.zeros(1, device=self.dev)
        D = abi.LsimAmpDisc()
        for i, l in enumerate(self.layers):
            h = D.hidden[i]
            h.weight, h.bias = self.w[i].data_ptr(), self.b[i].data_ptr()
            h.k_pad, h.n_pad, h.k_in, h.n_out = self._rows[i].shape[1], self._rows[i].shape[0], l.in_features, l.out_features
        D.head_weight, D.head_bias = self.head_w.data_ptr(), self.head_b.data_ptr()
        D.amp_dim = self.layers[0].in_features // 2
        D.reward_coef, D.task_reward_lerp = float(disc.amp_reward_coef), float(disc.task_reward_lerp)
        if normalizer is not None:
            D.norm_eps, D.norm_clip = float(normalizer.epsilon), float(normalizer.clip_obs)
        self._D = D
        n = ctypes.c_size_t(0)
        lib.check(self._L.lsim_amp_step_workspace(int(num_envs), ctypes.byref(n)), what="lsim_amp_step_workspace")
        self.workspace = torch.zeros((n.value + 3) // 4, dtype=torch.int32, device=self.dev)      # zero-filled once: the kernel leaves it ready
        self.num_envs = int(num_envs)
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """copy the current parameters into the padded buffers (after every optimiser step the rollout should see)"""
        for l, r, w, b in zip(self.layers, self._rows, self.w, self.b):
            r[:l.out_features, :l.in_features].copy_(l.weight)
            w.copy_(r.view(r.shape[0] // 16, 16, r.shape[1] // 16, 16).permute(0, 2, 1, 3))
            b[:l.out_features].copy_(l.bias)
        hl = self.disc.amp_linear
        self.head_w[:hl.in_features].copy_(hl.weight.view(-1))
        self.head_b.copy_(hl.bias.view(-1))

    def step(self, amp_obs, next_amp_obs, dones, terminal_amp_states, task_rewards, rewards_out, disc_out=None, carry=None, replay=None):
        """one lsim_amp_step; `replay`: a learn.amp.ReplayBuffer whose ring receives the pair (its cursor advances as insert() would)"""
        n = amp_obs.shape[0]
        if n > self.num_envs:
            raise ValueError("more rows than the workspace was sized for")
        D = self._D
        nz = self.normalizer
        if nz is not None:      # the running moments are re-bound by Normalizer.update: read the live addresses
            D.norm_mean, D.norm_var = nz._mean.data_ptr(), nz._var.data_ptr()
        else:
            D.norm_mean = D.norm_var = None
        rs = rns = None
        cap = cur = 0
        if replay is not None:
            cur = replay.reserve(n)
            rs, rns, cap = replay.states.data_ptr(), replay.next_states.data_ptr(), replay.buffer_size
        lib.check(self._L.lsim_amp_step(ctypes.byref(D), amp_obs.data_ptr(), next_amp_obs.data_ptr(), dones.data_ptr() if dones is not None else None,
                                        terminal_amp_states.data_ptr() if terminal_amp_states is not None else None,
                                        task_rewards.data_ptr() if task_rewards is not None else None, n, rewards_out.data_ptr(),
                                        disc_out.data_ptr() if disc_out is not None else None, carry.data_ptr() if carry is not None else None,
                                        rs, rns, cap, cur, self.workspace.data_ptr(), self.workspace.numel() * 4,
                                        torch.cuda.current_stream(self.dev).cuda_stream), what="lsim_amp_step")
