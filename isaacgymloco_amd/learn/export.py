"""Deployment export of a trained policy: the counterpart of legged_gym.utils.helpers.export_policy_as_jit / PolicyExporterHIM
(HLP:201-212, HLP:248-264), which play.py calls (PLAY:71-74) to write `<log_dir>/exported/policies/policy.pt`.

The exported TorchScript module maps an observation history [B, 270] to action means [B, 12]:
    encoder(obs)[:, :19] -> (velocity[3], L2-normalised latent[16]);  actor(cat(obs[:, :45], velocity, latent)).
The build's networks are `HimMLP` / `SkinnyLinear` modules (nn.Sequential / nn.Linear subclasses whose forward dispatches to the HIP
weight-gradient kernels under autograd), which TorchScript cannot script; the exporter therefore re-materialises the two networks as plain
`nn.Sequential(nn.Linear, nn.ELU, ...)` with the SAME weights -- exactly the module tree the reference scripts, so a file written here loads
wherever the reference's file does (torch.jit.load, C++ libtorch on the robot) and gives the same outputs.
"""
import copy
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


def plain_sequential(seq):
    """nn.Sequential of plain nn.Linear / activation modules with copies of `seq`'s parameters (state_dict keys unchanged)."""
    layers = []
    for m in seq:
        if isinstance(m, nn.Linear):
            lin = nn.Linear(m.in_features, m.out_features, bias=m.bias is not None)
            with torch.no_grad():
                lin.weight.copy_(m.weight.detach().cpu())
                if m.bias is not None:
                    lin.bias.copy_(m.bias.detach().cpu())
            layers.append(lin)
        else:
            layers.append(copy.deepcopy(m).cpu())
    return nn.Sequential(*layers)


class PolicyExporterHIM(nn.Module):
    """HLP:248-264: actor + estimator encoder, forward(obs_history) -> action means."""

    def __init__(self, actor_critic):
        super().__init__()
        self.actor = plain_sequential(actor_critic.actor)
        self.estimator = plain_sequential(actor_critic.estimator.encoder)
        self.num_one_step_obs = int(actor_critic.num_one_step_obs)
        self.num_enc_out = int(actor_critic.estimator.num_latent) + 3

    def forward(self, obs_history: torch.Tensor) -> torch.Tensor:
        parts = self.estimator(obs_history)[:, 0:self.num_enc_out]
        vel, z = parts[..., :3], parts[..., 3:]
        z = F.normalize(z, dim=-1, p=2.0)
        return self.actor(torch.cat((obs_history[:, 0:self.num_one_step_obs], vel, z), dim=1))

    def export(self, path):
        os.makedirs(path, exist_ok=True)
        path = os.path.join(path, "policy.pt")
        self.to("cpu")
        torch.jit.script(self).save(path)
        return path


def export_policy_as_jit(actor_critic, path):
    """HLP:201-212.  `path` is a directory; returns the file written (policy.pt for HIM policies, policy_1.pt for a bare actor)."""
    if hasattr(actor_critic, "estimator"):
        return PolicyExporterHIM(actor_critic).export(path)
    os.makedirs(path, exist_ok=True)
    out = os.path.join(path, "policy_1.pt")
    torch.jit.script(plain_sequential(actor_critic.actor)).save(out)
    return out
