"""Learner parity: our HIMActorCritic / HIMEstimator / HIMRolloutStorage / HIMPPO against golden
vectors captured from the reference rsl_rl with the same seeds (tools/gen_golden_learner.py).  CPU, fp32."""
import os

import numpy as np
import torch

from helpers import ROOT
from isaacgymloco_amd.learn.him_ppo import HIMPPO
from isaacgymloco_amd.learn.modules import HIMActorCritic, sinkhorn

FX = os.path.join(ROOT, "tests", "golden", "learner_himppo.npz")
ALG = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=2,
           num_mini_batches=2, learning_rate=1e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0)


def _ck(module):
    return {k: np.array([float(v.double().sum()), float(v.double().abs().sum())]) for k, v in module.state_dict().items()}


def test_himppo_rollout_gae_update_match_reference():
    fx = np.load(FX)
    T, N = fx["rew_seq"].shape
    torch.manual_seed(0)
    ac = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    for k, v in _ck(ac).items():      # same layer creation order => same weights from the same seed, same state_dict keys
        np.testing.assert_allclose(v, fx["init/" + k], rtol=1e-12, err_msg=k)
    assert sum(p.numel() for p in ac.parameters()) == 545660
    alg = HIMPPO(ac, device="cpu", **ALG)
    alg.init_storage(N, T, [270], [238], [12])
    obs, crit = torch.from_numpy(fx["obs_seq"]), torch.from_numpy(fx["crit_seq"])
    rew, done, tout = torch.from_numpy(fx["rew_seq"]), torch.from_numpy(fx["done_seq"]), torch.from_numpy(fx["tout_seq"])
    torch.manual_seed(1)
    with torch.inference_mode():
        for t in range(T):
            a = alg.act(obs[t], crit[t])
            np.testing.assert_allclose(a.numpy(), fx["actions"][t], rtol=1e-6, atol=1e-6)
            alg.process_env_step(rew[t], done[t], {"time_outs": tout[t]}, crit[t + 1])
        alg.compute_returns(crit[T])
    np.testing.assert_allclose(alg.storage.values.numpy(), fx["values"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(alg.storage.returns.numpy(), fx["returns"], rtol=1e-6, atol=1e-6)          # GAE, HST:113-123
    np.testing.assert_allclose(alg.storage.advantages.numpy(), fx["advantages"], rtol=1e-5, atol=1e-6)    # HST:126-127
    torch.manual_seed(2)
    losses = alg.update()
    np.testing.assert_allclose(np.array(losses), fx["losses"], rtol=1e-4, atol=1e-6)
    assert abs(alg.learning_rate - float(fx["final_lr"])) < 1e-12
    for k, v in _ck(ac).items():
        np.testing.assert_allclose(v, fx["final/" + k], rtol=1e-4, atol=1e-5, err_msg=k)


def test_sinkhorn_matches_reference():
    fx = np.load(FX)
    out = sinkhorn(torch.from_numpy(fx["sinkhorn_in"]).clone()).numpy()
    np.testing.assert_allclose(out, fx["sinkhorn_out"], rtol=1e-6, atol=1e-8)
