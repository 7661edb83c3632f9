"""Deployment export: export_policy_as_jit / PolicyExporterHIM (HLP:201-264) and checkpoint interchange with the
reference's own classes.  CPU; the reference-side half runs only where /root/reference exists (the build container)."""
import os
import sys

import numpy as np
import pytest
import torch

from isaacgymloco_amd.learn.export import PolicyExporterHIM, export_policy_as_jit
from isaacgymloco_amd.learn.modules import HIMActorCritic

REF = "/root/reference/rsl_rl"


def _policy(seed=3):
    torch.manual_seed(seed)
    ac = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    with torch.no_grad():      # move away from the initialisation so that every parameter matters
        for p in ac.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return ac


def test_exported_torchscript_reproduces_act_inference(tmp_path):
    ac = _policy()
    path = export_policy_as_jit(ac, str(tmp_path / "exported" / "policies"))
    assert os.path.basename(path) == "policy.pt"                       # the file name play.py's consumers expect (HLP:259)
    mod = torch.jit.load(path)
    obs = torch.randn(37, 270)
    with torch.no_grad():
        want = ac.act_inference(obs)
        got = mod(obs)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-6)
    assert sorted(mod.state_dict()) == sorted(PolicyExporterHIM(ac).state_dict())
    assert all(k.startswith(("actor.", "estimator.")) for k in mod.state_dict())      # the reference exporter's module tree


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_classes_load_a_build_checkpoint(tmp_path):
    """a checkpoint saved with the runner's keys loads STRICTLY into the reference's HIMActorCritic (what its play.py / runner.load do) and
    gives the same act_inference / evaluate outputs to 1e-6; the reference's own exporter and ours write modules with identical outputs"""
    sys.path.insert(0, REF)
    try:
        from rsl_rl.modules import HIMActorCritic as RefAC
    finally:
        sys.path.remove(REF)
    ac = _policy(seed=5)
    ck = str(tmp_path / "model_7.pt")
    opt = torch.optim.Adam(ac.parameters(), lr=1e-3)
    torch.save({"model_state_dict": ac.state_dict(), "optimizer_state_dict": opt.state_dict(),
                "estimator_optimizer_state_dict": ac.estimator.optimizer.state_dict(), "iter": 7, "infos": None}, ck)   # HIMR:233-240
    d = torch.load(ck, weights_only=False)
    ref = RefAC(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    ref.load_state_dict(d["model_state_dict"])               # strict: same keys, same shapes
    ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    ref_opt.load_state_dict(d["optimizer_state_dict"])        # HIMR:247-249 load_optimizer=True
    obs, crit = torch.randn(29, 270), torch.randn(29, 238)
    with torch.no_grad():
        np.testing.assert_allclose(ref.act_inference(obs).numpy(), ac.act_inference(obs).numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(ref.evaluate(crit).numpy(), ac.evaluate(crit).numpy(), rtol=0, atol=1e-6)
    # the reference's exporter (restated here from HLP:248-264 because legged_gym.utils.helpers imports isaacgym) scripted on ITS module tree
    import copy
    import torch.nn.functional as F

    class RefExporter(torch.nn.Module):
        def __init__(self, actor_critic):
            super().__init__()
            self.actor = copy.deepcopy(actor_critic.actor)
            self.estimator = copy.deepcopy(actor_critic.estimator.encoder)

        def forward(self, obs_history):
            parts = self.estimator(obs_history)[:, 0:19]
            vel, z = parts[..., :3], parts[..., 3:]
            z = F.normalize(z, dim=-1, p=2.0)
            return self.actor(torch.cat((obs_history[:, 0:45], vel, z), dim=1))
    ref_script = torch.jit.script(RefExporter(ref))
    mine = torch.jit.load(export_policy_as_jit(ac, str(tmp_path / "exp")))
    with torch.no_grad():
        np.testing.assert_allclose(mine(obs).numpy(), ref_script(obs).numpy(), rtol=0, atol=1e-6)
    assert sorted(mine.state_dict()) == sorted(ref_script.state_dict())
