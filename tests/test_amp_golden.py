"""AMP path parity against golden vectors from the reference rsl_rl (tools/gen_golden_amp.py):
AMPLoader pre-sampling, Normalizer, AMPDiscriminator reward / gradient penalty, ReplayBuffer, one HybridPPO update.  CPU."""
import os

import numpy as np
import torch

from helpers import ROOT
from isaacgymloco_amd.learn import amp
from isaacgymloco_amd.learn.hybrid import HybridPPO
from isaacgymloco_amd.learn.modules import HIMActorCritic

FX = os.path.join(ROOT, "tests", "golden", "learner_amp.npz")
BUNDLE = os.path.join(ROOT, "isaacgymloco_amd", "data", "mocap_aliengo.npz")
ALG = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=2,
           num_mini_batches=2, learning_rate=1e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0,
           amp_replay_buffer_size=64)


def _ck(module):
    return {k: np.array([float(v.double().sum()), float(v.double().abs().sum())]) for k, v in module.state_dict().items()}


def _loader(n=4000):
    np.random.seed(1)
    return amp.AMPLoader("cpu", time_between_frames=0.02, preload_transitions=True, num_preload_transitions=n, motion_files=[BUNDLE])


def test_loader_presampling_matches_reference():
    fx = np.load(FX)
    ld = _loader()
    assert ld.observation_dim == 30 and ld.num_motions == 7
    np.testing.assert_allclose(ld.preloaded_s.numpy(), fx["pre_s"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(ld.preloaded_s_next.numpy(), fx["pre_s_next"], rtol=0, atol=1e-7)
    gen = ld.feed_forward_generator(2, 16)
    for i in range(2):
        s, sn = next(gen)
        np.testing.assert_allclose(s.numpy(), fx["ff_s"][i], atol=1e-7)
        np.testing.assert_allclose(sn.numpy(), fx["ff_s_next"][i], atol=1e-7)


def test_normalizer_discriminator_replay_match_reference():
    fx = np.load(FX)
    nz = amp.Normalizer(30)
    nz.update(fx["nz_x1"]); nz.update(fx["nz_x2"])
    np.testing.assert_allclose(nz.mean, fx["nz_mean"], rtol=2e-6, atol=1e-7)     # float32 batch moments, summation order differs
    np.testing.assert_allclose(nz.var, fx["nz_var"], rtol=2e-6, atol=1e-7)
    assert abs(nz.count - float(fx["nz_count"])) < 1e-9
    np.testing.assert_allclose(nz.normalize_torch(torch.from_numpy(fx["nz_probe"])).numpy(), fx["nz_probe_out"], rtol=1e-6, atol=1e-6)
    torch.manual_seed(3)
    disc = amp.AMPDiscriminator(60, 0.5 * 0.02, [1024, 512], "cpu", 0.3)
    assert sum(p.numel() for p in disc.parameters()) == 587777
    r, d = disc.predict_amp_reward(torch.from_numpy(fx["disc_s"]), torch.from_numpy(fx["disc_ns"]), torch.from_numpy(fx["disc_task"]), normalizer=nz)
    np.testing.assert_allclose(d.numpy(), fx["disc_d"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r.numpy(), fx["disc_reward"], rtol=1e-5, atol=1e-7)
    gp = disc.compute_grad_pen(torch.from_numpy(fx["disc_s"]), torch.from_numpy(fx["disc_ns"]), lambda_=10).item()
    assert abs(gp - float(fx["disc_gp"])) < 1e-5 * max(1.0, abs(gp))
    rb = amp.ReplayBuffer(30, 20, "cpu")
    for c in torch.from_numpy(fx["rb_chunks"]):
        rb.insert(c, c + 1)
    np.testing.assert_array_equal(rb.states.numpy(), fx["rb_states"])
    np.testing.assert_array_equal(rb.next_states.numpy(), fx["rb_next"])
    assert rb.step == int(fx["rb_step"]) and rb.num_samples == int(fx["rb_num"])


def test_hybrid_ppo_update_matches_reference():
    fx = np.load(FX)
    ld = _loader()
    list(ld.feed_forward_generator(2, 16))        # the generator consumed two np.random draws before the update
    g = None
    N, T = 8, 6
    torch.manual_seed(3)
    _ = amp.AMPDiscriminator(60, 0.5 * 0.02, [1024, 512], "cpu", 0.3)   # the fixture created one discriminator before (RNG order)
    torch.manual_seed(0)
    ac = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    disc = amp.AMPDiscriminator(60, 0.5 * 0.02, [1024, 512], "cpu", 0.3)
    nz = amp.Normalizer(30)
    alg = HybridPPO(ac, disc, ld, nz, device="cpu", min_std=torch.tensor([0.05, 0.02, 0.05] * 4) * 1.5, **ALG)
    alg.init_storage(N, T, [270], [238], [12])
    obs, crit, ampo = (torch.from_numpy(fx[k]) for k in ("hy_obs", "hy_crit", "hy_amp"))
    rew, done = torch.from_numpy(fx["hy_rew"]), torch.from_numpy(fx["hy_done"])
    torch.manual_seed(1)
    np.random.seed(7)
    with torch.inference_mode():
        for t in range(T):
            alg.act(obs[t], crit[t], ampo[t])
            r = alg.discriminator.predict_amp_reward(ampo[t], ampo[t + 1], rew[t], normalizer=alg.amp_normalizer)[0]
            alg.process_env_step(r, done[t], {"time_outs": done[t] & False}, ampo[t + 1], crit[t + 1])
        alg.compute_returns(crit[T])
    torch.manual_seed(2)
    res = alg.update()
    np.testing.assert_allclose(np.array(res), fx["hy_losses"], rtol=2e-4, atol=1e-6)
    assert abs(alg.learning_rate - float(fx["hy_lr"])) < 1e-12
    np.testing.assert_allclose(nz.mean, fx["hy_nz_mean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(nz.var, fx["hy_nz_var"], rtol=1e-6, atol=1e-7)
    for k, v in _ck(ac).items():
        np.testing.assert_allclose(v, fx["hy_ac/" + k], rtol=1e-4, atol=1e-5, err_msg=k)
    for k, v in _ck(disc).items():
        np.testing.assert_allclose(v, fx["hy_disc/" + k], rtol=1e-4, atol=1e-5, err_msg=k)


def test_closed_form_gradient_penalty_equals_autograd():
    """amp._GradPenFn (the GPU path of compute_grad_pen: the penalty and its gradients in closed form, no double backward) against the
    reference's autograd statement (DISC:36-53) on the CPU: value and every parameter gradient"""
    torch.manual_seed(0)
    d = amp.AMPDiscriminator(60, 0.01, [64, 32], "cpu", 0.3)
    s, ns = torch.randn(50, 30), torch.randn(50, 30)
    ref = d.compute_grad_pen(s, ns, 10)                     # CPU: autograd.grad(create_graph=True) + backward
    ref.backward()
    want = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in d.named_parameters()}
    d.zero_grad()
    l1, l2 = d.trunk[0], d.trunk[2]
    v = amp._GradPenFn.apply(torch.cat([s, ns], -1), l1.weight, l1.bias, l2.weight, l2.bias, d.amp_linear.weight, 10.0)
    v.backward()
    assert abs(float(v.detach()) - float(ref.detach())) < 1e-6 * max(1.0, abs(float(ref.detach())))
    for n, p in d.named_parameters():
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        torch.testing.assert_close(got, want[n], rtol=1e-5, atol=1e-7, msg=n)
