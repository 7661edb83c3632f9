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


def synth_inputs(n, t, seed):
    """the synthetic rollout of tools/gen_golden_learner.py, regenerated from its seed (torch's CPU generator is deterministic)"""
    g = torch.Generator().manual_seed(seed)
    obs_seq = torch.randn(t + 1, n, 270, generator=g)
    crit_seq = torch.randn(t + 1, n, 238, generator=g)
    rew_seq = torch.randn(t, n, generator=g)
    done_seq = torch.rand(t, n, generator=g) < 0.2
    tout_seq = done_seq & (torch.rand(t, n, generator=g) < 0.5)
    return obs_seq, crit_seq, rew_seq, done_seq, tout_seq


def load_large():
    fx = np.load(os.path.join(ROOT, "tests", "golden", "learner_himppo_large.npz"))
    ins = synth_inputs(int(fx["n"]), int(fx["t"]), int(fx["input_seed"]))
    sums = np.array([float(x.double().sum()) for x in (ins[0], ins[1], ins[2], ins[3].float(), ins[4].float())])
    np.testing.assert_allclose(sums, fx["input_checksums"], rtol=1e-12)      # the regenerated inputs ARE the fixture's inputs
    return fx, ins


def test_himppo_large_batch_matches_reference():
    """N = 128, T = 64 (minibatches of 4096 rows): same procedure as above against the reference at the batch size where the GPU
    build's weight-gradient kernels engage; this CPU leg pins the torch statement the GPU leg (test_gpu_learner_golden.py) shares"""
    fx, (obs, crit, rew, done, tout) = load_large()
    T, N = int(fx["t"]), int(fx["n"])
    torch.manual_seed(0)
    ac = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    alg = HIMPPO(ac, device="cpu", **ALG)
    alg.init_storage(N, T, [270], [238], [12])
    torch.manual_seed(1)
    with torch.inference_mode():
        for t in range(T):
            a = alg.act(obs[t], crit[t])
            np.testing.assert_allclose(a.numpy(), fx["actions"][t], rtol=1e-5, atol=1e-5)
            alg.process_env_step(rew[t], done[t], {"time_outs": tout[t]}, crit[t + 1])
        alg.compute_returns(crit[T])
    np.testing.assert_allclose(alg.storage.returns.numpy(), fx["returns"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(alg.storage.advantages.numpy(), fx["advantages"], rtol=1e-4, atol=1e-5)
    torch.manual_seed(2)
    losses = alg.update()
    np.testing.assert_allclose(np.array(losses), fx["losses"], rtol=1e-4, atol=1e-6)
    assert abs(alg.learning_rate - float(fx["final_lr"])) < 1e-12
    for k, v in _ck(ac).items():
        np.testing.assert_allclose(v, fx["final/" + k], rtol=1e-4, atol=1e-4, err_msg=k)
