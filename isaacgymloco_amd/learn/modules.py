"""HIM actor-critic and estimator (PyTorch-ROCm; stays in torch per the north-star).

Same architecture, parameter names and public methods as rsl_rl.modules.HIMActorCritic (HAC:43-163) and
rsl_rl.modules.HIMEstimator (HES:11-133), so checkpoints interchange with the reference
(`model_state_dict` keys `actor.N.*`, `critic.N.*`, `estimator.encoder.N.*`, `estimator.target.N.*`,
`estimator.proto.weight`, `std`) and layers are created in the same order (same weights from the same seed).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions import Normal

from .fused_linear import HimMLP, SkinnyLinear

_ACTIVATIONS = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "crelu": nn.ReLU, "silu": nn.SiLU,
                "lrelu": nn.LeakyReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}


def get_activation(name):
    if name not in _ACTIVATIONS:
        raise ValueError(f"invalid activation function '{name}'")
    return _ACTIVATIONS[name]()


def mlp(sizes, activation, last_activation=False):
    """Linear(sizes[0], sizes[1]) -> act -> ... -> Linear(sizes[-2], sizes[-1]) [-> act]"""
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(SkinnyLinear(sizes[i], sizes[i + 1]))   # an nn.Linear; narrow layers get the HIP weight-gradient kernel on the GPU
        if i < len(sizes) - 2 or last_activation:
            layers.append(activation)
    return HimMLP(*layers)      # an nn.Sequential; fuses (Linear, ELU) backward passes on the GPU (fused_linear.py)


@torch.no_grad()
def sinkhorn(scores, eps=0.05, iters=3):
    """Sinkhorn-Knopp assignment of a batch of prototype scores (HES:119-133); returns (B, K)."""
    if scores.is_cuda and scores.dim() == 2 and scores.dtype == torch.float32 and scores.shape[1] <= 64 and scores.shape[0] >= 4096:
        from .fused_linear import sinkhorn_hip
        return sinkhorn_hip(scores, eps, iters)       # 2 * iters + 1 launches instead of ~26 torch kernels (include/lsim.h)
    Q = torch.exp(scores / eps).T
    K, B = Q.shape
    Q /= Q.sum()
    for _ in range(iters):
        Q /= Q.sum(dim=1, keepdim=True)
        Q /= K
        Q /= Q.sum(dim=0, keepdim=True)
        Q /= B
    return (Q * B).T


class HIMEstimator(nn.Module):
    """History encoder -> (velocity estimate, unit latent); contrastive target on the next observation (HES:11-116)."""

    def __init__(self, temporal_steps, num_one_step_obs, enc_hidden_dims=(128, 64, 16), tar_hidden_dims=(128, 64),
                 activation="elu", learning_rate=1e-3, max_grad_norm=10.0, num_prototype=32, temperature=3.0, **kwargs):
        super().__init__()
        act = get_activation(activation)
        self.temporal_steps = temporal_steps
        self.num_one_step_obs = num_one_step_obs
        self.num_latent = enc_hidden_dims[-1]
        self.max_grad_norm = max_grad_norm
        self.fused_step = False
        self.temperature = temperature
        self.encoder = mlp([temporal_steps * num_one_step_obs, *enc_hidden_dims[:-1], enc_hidden_dims[-1] + 3], act)
        self.target = mlp([num_one_step_obs, *tar_hidden_dims, enc_hidden_dims[-1]], act)
        self.proto = nn.Embedding(num_prototype, enc_hidden_dims[-1])
        self.learning_rate = learning_rate
        self.optimizer = torch.optim.Adam(self.parameters(), lr=learning_rate)
        self.grad_sync = None   # set by the data-parallel runner: callable(list_of_params) averaging .grad over ranks
        self._primed = None     # (obs_history tensor, encoder output with autograd graph) shared by encode() and losses()

    def prime(self, obs_history):
        """Run the encoder ONCE for a minibatch, with autograd, and let both consumers of that minibatch use it: the policy's
        no-grad input features (HAC:136-141) and the estimator's own loss (HES:76-108).  The reference runs the same forward
        twice with unchanged weights; the values are identical."""
        self._primed = (obs_history, self.encoder(obs_history.detach()))

    def _encoder_out(self, obs_history, want_grad):
        if self._primed is not None and self._primed[0] is obs_history:
            out = self._primed[1]
            return out if want_grad else out.detach()
        return self.encoder(obs_history if want_grad else obs_history.detach())

    def encode(self, obs_history):
        out = self._encoder_out(obs_history, want_grad=False)
        return out[..., :3], F.normalize(out[..., 3:], dim=-1, p=2)

    def forward(self, obs_history):
        vel, z = self.encode(obs_history)
        return vel.detach(), z.detach()

    get_latent = forward

    def losses(self, obs_history, next_critic_obs):
        """estimation (MSE on base velocity) and swap (SwAV-style) losses of HES:76-108 -> (est, swap, est + swap to back-propagate)."""
        n = self.num_one_step_obs
        vel = next_critic_obs[:, n:n + 3].detach()
        next_obs = next_critic_obs.detach()[:, 3:n + 3]
        out = self._encoder_out(obs_history, want_grad=True)
        tgt = self.target(next_obs)
        w = self.proto.weight
        if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.numel() <= 4096:
            from .. import lib      # the same statement in one launch instead of five (include/lsim.h: lsim_normalize_rows)
            lib.check(lib.load().lsim_normalize_rows(w.data_ptr(), w.shape[0], w.shape[1], 1e-12, torch.cuda.current_stream(w.device).cuda_stream),
                      what="lsim_normalize_rows")
        else:
            with torch.no_grad():
                w.copy_(F.normalize(w.data.clone(), dim=-1, p=2))
        if out.is_cuda and out.dim() == 2:
            from .fused_linear import estimator_loss_hip, estimator_loss_supported
            if estimator_loss_supported(tgt.shape[1], self.proto.weight.shape[0]):
                # normalise + scores + both Sinkhorn chains + log-softmax + losses and their backward: 10 launches (include/lsim.h)
                total, parts = estimator_loss_hip(out, tgt, self.proto.weight, vel, self.temperature)
                return parts[0], parts[1], total
        pred_vel, z_s = out[..., :3], F.normalize(out[..., 3:], dim=-1, p=2)
        z_t = F.normalize(tgt, dim=-1, p=2)
        score_s = z_s @ self.proto.weight.T
        score_t = z_t @ self.proto.weight.T
        with torch.no_grad():
            q_s, q_t = sinkhorn(score_s), sinkhorn(score_t)
        log_p_s = F.log_softmax(score_s / self.temperature, dim=-1)
        log_p_t = F.log_softmax(score_t / self.temperature, dim=-1)
        swap = -0.5 * (q_s * log_p_t + q_t * log_p_s).mean()
        est = F.mse_loss(pred_vel, vel)
        return est, swap, est + swap

    def update(self, obs_history, next_critic_obs, lr=None):
        if lr is not None:
            self.learning_rate = lr
            for g in self.optimizer.param_groups:
                g["lr"] = lr
        est, swap, total = self.losses(obs_history, next_critic_obs)
        self._primed = None
        self.optimizer.zero_grad()
        from . import fused_linear as FL
        FL.grad_cycle()
        with FL.deferred_wgrad_reduce():
            FL.backward_losses(total)
        if self.grad_sync is not None:
            self.grad_sync(list(self.parameters()))
        elif FL._arena is not None and next(self.parameters()).is_cuda:      # stable gradient pointers for the fused optimiser step
            FL._arena.bucket("estimator", [p for p in self.parameters() if p.grad is not None]).adopt()
        if self.fused_step:       # set by HIMPPO.enable_device_lr: clip + Adam in one C-ABI call (two launches)
            from .fused_linear import adam_clip_step_hip
            if adam_clip_step_hip(self.optimizer, self.max_grad_norm):
                return est.detach(), swap.detach()
        nn.utils.clip_grad_norm_(self.parameters(), self.max_grad_norm)
        self.optimizer.step()
        return est.detach(), swap.detach()


class HIMActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_one_step_obs, num_actions, actor_hidden_dims=(512, 256, 128),
                 critic_hidden_dims=(512, 256, 128), activation="elu", init_noise_std=1.0, **kwargs):
        super().__init__()
        act = get_activation(activation)
        self.history_size = int(num_actor_obs / num_one_step_obs)
        self.num_actor_obs, self.num_actions, self.num_one_step_obs = num_actor_obs, num_actions, num_one_step_obs
        self.estimator = HIMEstimator(temporal_steps=self.history_size, num_one_step_obs=num_one_step_obs)
        self.actor = mlp([num_one_step_obs + 3 + 16, *actor_hidden_dims, num_actions], act)
        self.critic = mlp([num_critic_obs, *critic_hidden_dims, 1], act)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None

    def reset(self, dones=None):
        pass

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def _actor_input(self, obs_history):
        if obs_history.is_cuda and obs_history.dim() == 2 and obs_history.dtype == torch.float32 and obs_history.stride(1) == 1:
            from .. import lib        # one launch (lsim_actor_input) instead of norm + clamp + div + cat
            with torch.no_grad():
                enc = self.estimator._encoder_out(obs_history, want_grad=False)
            n1, nl = self.num_one_step_obs, enc.shape[1] - 3
            out = torch.empty(obs_history.shape[0], n1 + 3 + nl, device=obs_history.device, dtype=torch.float32)
            lib.check(lib.load().lsim_actor_input(obs_history.data_ptr(), obs_history.stride(0), n1, enc.data_ptr(), enc.stride(0), nl,
                                                  obs_history.shape[0], out.data_ptr(), torch.cuda.current_stream(obs_history.device).cuda_stream),
                      what="lsim_actor_input")
            return out
        with torch.no_grad():
            vel, latent = self.estimator(obs_history)
        return torch.cat((obs_history[:, :self.num_one_step_obs], vel, latent), dim=-1)

    def update_distribution(self, obs_history):
        mean = self.actor(self._actor_input(obs_history))
        # validate_args=False: the reference intends this (HAC:103 assigns Normal.set_default_validate_args = False, a no-op);
        # argument validation is a host sync per call and cannot be captured into a HIP graph
        self.distribution = Normal(mean, mean * 0.0 + self.std, validate_args=False)

    def act(self, obs_history=None, **kwargs):
        self.update_distribution(obs_history)
        return self.distribution.sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, obs_history, observations=None):
        return self.actor(self._actor_input(obs_history))

    def evaluate(self, critic_observations, **kwargs):
        return self.critic(critic_observations)
