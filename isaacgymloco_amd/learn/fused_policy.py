"""Rollout-time forward of the HIM policy through the library's fused MFMA kernel (include/lsim.h, lsim_policy_forward).

`PackedHimPolicy` keeps zero-padded copies of the eleven nn.Linear parameter pairs in the layout the kernel wants (both dimensions
multiples of 16, cut into 16 x 16 blocks) at fixed device addresses; `refresh()` re-copies them after every policy update (22 small copies per PPO
iteration).  Same arithmetic as HIMActorCritic.update_distribution / evaluate (HAC:136-163) up to fp32 summation order.
"""
import ctypes

import torch
import torch.nn as nn

from .. import abi, lib


def _linears(seq):
    mods = list(seq)
    lin = [m for m in mods if isinstance(m, nn.Linear)]
    act_ok = all(isinstance(m, (nn.Linear, nn.ELU)) for m in mods) and all(getattr(m, "alpha", 1.0) == 1.0 for m in mods if isinstance(m, nn.ELU))
    # Linear / ELU alternate and the last module is a Linear
    shape_ok = len(mods) == 2 * len(lin) - 1 and all(isinstance(m, nn.Linear) for m in mods[0::2])
    return lin if (act_ok and shape_ok) else None


class PackedHimPolicy:
    @staticmethod
    def supported(ac):
        enc, act, cri = _linears(ac.estimator.encoder), _linears(ac.actor), _linears(ac.critic)
        if enc is None or act is None or cri is None or (len(enc), len(act), len(cri)) != (3, 4, 4):
            return False
        wide = max(l.out_features for l in enc + act + cri)
        second = max(enc[1].out_features, act[1].out_features, cri[1].out_features)     # these land in the narrower LDS buffer
        return wide <= 512 and second <= 272 and ac.num_actor_obs <= 272 and cri[0].in_features <= 272 and ac.num_actions <= 16 and next(ac.parameters()).is_cuda

    def __init__(self, ac):
        assert self.supported(ac)
        self.ac = ac
        self.dev = next(ac.parameters()).device
        self._L = lib.load()
        self.layers = _linears(ac.estimator.encoder) + _linears(ac.actor) + _linears(ac.critic)
        pad = lambda v: (v + 15) // 16 * 16
        # weights as the kernel reads them: 16 x 16 blocks, [n_pad / 16][k_pad / 16][16][16] (include/lsim.h) -- self.w holds the blocks, the
        # row-major padded image they are cut from is scratch
        self._rows = [torch.zeros(pad(l.out_features), pad(l.in_features), device=self.dev) for l in self.layers]
        self.w = [torch.zeros(r.shape[0] // 16, r.shape[1] // 16, 16, 16, device=self.dev) for r in self._rows]
        self.b = [torch.zeros(pad(l.out_features), device=self.dev) for l in self.layers]
        P = abi.LsimHimPolicy()
        for idx, l in enumerate(self.layers):
            dst = P.encoder[idx] if idx < 3 else (P.actor[idx - 3] if idx < 7 else P.critic[idx - 7])
            dst.weight, dst.bias = self.w[idx].data_ptr(), self.b[idx].data_ptr()
            dst.k_pad, dst.n_pad, dst.k_in, dst.n_out = self._rows[idx].shape[1], self._rows[idx].shape[0], l.in_features, l.out_features
        P.num_obs, P.num_priv_obs = ac.num_actor_obs, self.layers[7].in_features
        P.num_one_step_obs, P.num_actions = ac.num_one_step_obs, ac.num_actions
        self._P = P
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """copy the current parameters into the padded buffers (call after every optimiser step that the rollout should see)"""
        for l, r, w, b in zip(self.layers, self._rows, self.w, self.b):
            r[:l.out_features, :l.in_features].copy_(l.weight)
            w.copy_(r.view(r.shape[0] // 16, 16, r.shape[1] // 16, 16).permute(0, 2, 1, 3))
            b[:l.out_features].copy_(l.bias)

    def forward(self, obs, priv_obs, mean_out, values_out):
        n = obs.shape[0]
        lib.check(self._L.lsim_policy_forward(ctypes.byref(self._P), obs.data_ptr(), priv_obs.data_ptr(), n, mean_out.data_ptr(), values_out.data_ptr(),
                                              torch.cuda.current_stream(self.dev).cuda_stream), what="lsim_policy_forward")

    def forward_act(self, storage_struct, step, draw, obs, priv_obs, std, seed, rank, mean_out, values_out, actions_out, prev=None):
        """lsim_policy_act_at: the networks, the action sample and the storage row of rollout step `step` in one launch; with
        prev = (step, dones, time_outs or None, rewards, term_priv_obs, gamma) also the previous step's post-step store (lsim_policy_act_post_at)"""
        s = torch.cuda.current_stream(self.dev).cuda_stream
        if prev is None:
            lib.check(self._L.lsim_policy_act_at(ctypes.byref(self._P), ctypes.byref(storage_struct), int(step), int(draw), obs.data_ptr(), priv_obs.data_ptr(),
                                                 std.data_ptr(), seed, rank, mean_out.data_ptr(), values_out.data_ptr(), actions_out.data_ptr(), s),
                      what="lsim_policy_act_at")
            return
        pstep, dones, touts, rewards, term, gamma = prev
        lib.check(self._L.lsim_policy_act_post_at(ctypes.byref(self._P), ctypes.byref(storage_struct), int(step), int(draw), obs.data_ptr(), priv_obs.data_ptr(),
                                                  std.data_ptr(), seed, rank, mean_out.data_ptr(), values_out.data_ptr(), actions_out.data_ptr(),
                                                  int(pstep), dones.data_ptr(), touts.data_ptr() if touts is not None else None, rewards.data_ptr(),
                                                  term.data_ptr(), float(gamma), s), what="lsim_policy_act_post_at")
