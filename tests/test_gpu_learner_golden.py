"""GPU: the learner's golden fixtures (captured from the reference rsl_rl, tools/gen_golden_learner.py / gen_golden_amp.py) replayed on
the HIP path -- fused GAE (lsim_rollout_gae), PPO loss (lsim_ppo_loss), estimator loss (lsim_estimator_loss), device learning rate
(lsim_adaptive_lr), Adam + clipping (lsim_adam_clip_step) and, at the large batch, the MFMA weight-gradient kernels
(lsim_linear_wgrad / lsim_linear_elu_wgrad).  HIMP:125-198, HYBP:117-306, HST:113-127.

What is injected and why: the actions (the reference sampled them from torch's CPU generator; the value of a sample is not what is
pinned here) and the minibatch permutation (torch.randperm on the GPU draws from a different generator than on the CPU).
Tolerances (fp32, different summation order of GPU GEMMs vs the CPU reference) are written at each assert.
"""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT
from test_learner_golden import ALG, FX, _ck, load_large

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inject_cpu_randperm(monkeypatch):
    orig = torch.randperm
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: orig(n).to(kw.get("device", "cpu")))


def _replay_rollout(alg, obs, crit, rew, done, tout, actions, amp=None):
    """HIMP:90-118 with the fixture's actions in place of a fresh sample"""
    ac = alg.actor_critic
    T = rew.shape[0]
    with torch.inference_mode():
        for t in range(T):
            tr = alg.transition
            ac.update_distribution(obs[t])
            tr.actions = actions[t]
            tr.values = ac.evaluate(crit[t]).detach()
            tr.actions_log_prob = ac.get_actions_log_prob(actions[t]).detach()
            tr.action_mean, tr.action_sigma = ac.action_mean.detach(), ac.action_std.detach()
            tr.observations, tr.critic_observations = obs[t], crit[t]
            if amp is None:
                alg.process_env_step(rew[t], done[t], {"time_outs": tout[t]}, crit[t + 1])
            else:
                alg.amp_transition.observations = amp[t]
                r = alg.discriminator.predict_amp_reward(amp[t], amp[t + 1], rew[t], normalizer=alg.amp_normalizer)[0]
                alg.process_env_step(r, done[t], {"time_outs": tout[t]}, amp[t + 1], crit[t + 1])
        alg.compute_returns(crit[T])


def _himppo_on_gpu():
    from isaacgymloco_amd.learn.him_ppo import HIMPPO
    from isaacgymloco_amd.learn.modules import HIMActorCritic
    torch.manual_seed(0)
    ac = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    alg = HIMPPO(ac, device=DEV, **ALG)         # weights drawn on the CPU from the fixture's seed, then moved
    assert alg.enable_device_lr()               # the fused optimiser side: device lr, lsim_adam_clip_step
    return ac, alg


def test_himppo_fixture_on_hip_path(monkeypatch):
    fx = np.load(FX)
    T, N = fx["rew_seq"].shape
    ac, alg = _himppo_on_gpu()
    for k, v in _ck(ac).items():
        np.testing.assert_allclose(v, fx["init/" + k], rtol=1e-12, err_msg=k)
    alg.init_storage(N, T, [270], [238], [12])
    dv = lambda k: torch.from_numpy(fx[k]).to(DEV)
    _replay_rollout(alg, dv("obs_seq"), dv("crit_seq"), dv("rew_seq"), dv("done_seq"), dv("tout_seq"), dv("actions"))
    st = alg.storage
    np.testing.assert_allclose(st.values.cpu().numpy(), fx["values"], rtol=1e-5, atol=2e-6)          # critic forward, GPU GEMM vs CPU
    np.testing.assert_allclose(st.returns.cpu().numpy(), fx["returns"], rtol=1e-5, atol=1e-5)        # lsim_rollout_gae vs HST:113-123
    np.testing.assert_allclose(st.advantages.cpu().numpy(), fx["advantages"], rtol=1e-4, atol=1e-5)  # HST:126-127
    _inject_cpu_randperm(monkeypatch)
    torch.manual_seed(2)
    losses = alg.update()
    np.testing.assert_allclose(np.array(losses), fx["losses"], rtol=2e-4, atol=2e-6)
    assert abs(alg.learning_rate - float(fx["final_lr"])) < 1e-9       # the adaptive-KL decisions (HIMP:144-156) taken on the device
    for k, v in _ck(ac).items():
        np.testing.assert_allclose(v, fx["final/" + k], rtol=2e-4, atol=2e-5, err_msg=k)


def test_himppo_large_fixture_on_hip_path_uses_mfma_wgrad(monkeypatch):
    """N = 128, T = 64: minibatches of 4096 rows, where SkinnyLinear / HimMLP route the weight gradients through the MFMA kernels"""
    from isaacgymloco_amd.learn import fused_linear
    fx, (obs, crit, rew, done, tout) = load_large()
    T, N = int(fx["t"]), int(fx["n"])
    ac, alg = _himppo_on_gpu()
    alg.init_storage(N, T, [270], [238], [12])
    to = lambda x: x.to(DEV)
    _replay_rollout(alg, to(obs), to(crit), to(rew), to(done), to(tout), torch.from_numpy(fx["actions"]).to(DEV))
    st = alg.storage
    np.testing.assert_allclose(st.values.cpu().numpy(), fx["values"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(st.returns.cpu().numpy(), fx["returns"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(st.advantages.cpu().numpy(), fx["advantages"], rtol=1e-4, atol=2e-5)
    perm = torch.from_numpy(fx["perm"])
    orig = torch.randperm
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: perm.to(kw.get("device", "cpu")) if n == perm.numel() else orig(n, **kw))
    calls = {"wgrad": 0, "elu": 0}
    for name, key in (("_SkinnyLinearFn", "wgrad"), ("_LinearEluFn", "elu")):
        fn = getattr(fused_linear, name)
        orig_apply = fn.apply
        monkeypatch.setattr(fn, "apply", (lambda oa, k: (lambda *a: (calls.__setitem__(k, calls[k] + 1), oa(*a))[1]))(orig_apply, key))
    torch.manual_seed(2)
    losses = alg.update()
    assert calls["wgrad"] + calls["elu"] > 0, "the MFMA weight-gradient path did not run at minibatch 4096"
    np.testing.assert_allclose(np.array(losses), fx["losses"], rtol=5e-4, atol=5e-6)
    assert abs(alg.learning_rate - float(fx["final_lr"])) < 1e-9
    for k, v in _ck(ac).items():
        np.testing.assert_allclose(v, fx["final/" + k], rtol=5e-4, atol=2e-4, err_msg=k)


def test_hybrid_ppo_fixture_on_hip_path(monkeypatch):
    """HybridPPO (AMP) update of the reference fixture on the GPU: PPO + LSGAN + gradient penalty in one Adam (HYBP:117-306)"""
    from isaacgymloco_amd.learn import amp
    from isaacgymloco_amd.learn.hybrid import HybridPPO
    from isaacgymloco_amd.learn.modules import HIMActorCritic
    from test_amp_golden import ALG as AMP_ALG, BUNDLE
    fx = np.load(os.path.join(ROOT, "tests", "golden", "learner_amp.npz"))
    np.random.seed(1)
    ld = amp.AMPLoader(DEV, time_between_frames=0.02, preload_transitions=True, num_preload_transitions=4000, motion_files=[BUNDLE])
    np.testing.assert_allclose(ld.preloaded_s.cpu().numpy(), fx["pre_s"], rtol=0, atol=1e-6)       # expert pre-sampling on the device
    list(ld.feed_forward_generator(2, 16))        # the fixture's generator consumed two np.random draws before the update
    N, T = 8, 6
    torch.manual_seed(3)
    _ = amp.AMPDiscriminator(60, 0.5 * 0.02, [1024, 512], "cpu", 0.3)   # RNG order of the fixture
    torch.manual_seed(0)
    ac = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    disc = amp.AMPDiscriminator(60, 0.5 * 0.02, [1024, 512], "cpu", 0.3)
    nz = amp.Normalizer(30, device=DEV)
    alg = HybridPPO(ac, disc, ld, nz, device=DEV, min_std=(torch.tensor([0.05, 0.02, 0.05] * 4) * 1.5).to(DEV), **AMP_ALG)
    alg.discriminator.device = DEV
    alg.init_storage(N, T, [270], [238], [12])
    dv = lambda k: torch.from_numpy(fx[k]).to(DEV)
    obs, crit, ampo, rew, done = dv("hy_obs"), dv("hy_crit"), dv("hy_amp"), dv("hy_rew"), dv("hy_done")
    # the fixture's actions are not stored: sample them with the reference's CPU stream on a CPU twin of the (identical) initial policy
    torch.manual_seed(0)
    ac_cpu = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    torch.manual_seed(1)
    with torch.inference_mode():
        actions = torch.stack([ac_cpu.act(fx_obs) for fx_obs in torch.from_numpy(fx["hy_obs"])[:T]]).to(DEV)
    np.random.seed(7)
    _replay_rollout(alg, obs, crit, rew, done, done & False, actions, amp=ampo)
    _inject_cpu_randperm(monkeypatch)
    torch.manual_seed(2)
    res = alg.update()
    np.testing.assert_allclose(np.array(res), fx["hy_losses"], rtol=5e-4, atol=5e-6)
    assert abs(alg.learning_rate - float(fx["hy_lr"])) < 1e-9
    np.testing.assert_allclose(nz.mean, fx["hy_nz_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(nz.var, fx["hy_nz_var"], rtol=1e-5, atol=1e-6)
    for k, v in _ck(ac).items():
        np.testing.assert_allclose(v, fx["hy_ac/" + k], rtol=5e-4, atol=5e-5, err_msg=k)
    for k, v in _ck(disc).items():
        np.testing.assert_allclose(v, fx["hy_disc/" + k], rtol=5e-4, atol=5e-5, err_msg=k)


def _make_amp_runner():
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    from isaacgymloco_amd.learn.bench_train import train_cfg_dict
    from isaacgymloco_amd.learn.hybrid import HybridPolicyRunner
    cfg = C.TASKS["aliengo_amp"][0]()
    cfg.env.num_envs = 256
    cfg.env.episode_length_s = 0.1              # time-outs (and therefore terminal AMP states) inside the 8-step window
    cfg.terrain.terrain_proportions = [1.0, 0.0, 0.0, 0.0]
    env = LeggedRobot(cfg, sim_device=DEV, seed=5, using_amp=True)
    tc = train_cfg_dict("aliengo_amp")
    tc["runner"]["num_steps_per_env"] = 8
    tc["runner"]["amp_num_preload_transitions"] = 20000
    torch.manual_seed(0)
    np.random.seed(1)
    return env, HybridPolicyRunner(env, tc, log_dir=None, device=DEV)


def test_hybrid_fused_rollout_matches_eager_storage():
    """HybridFusedRollout (policy kernel, fused sample/store, style reward + replay insert on the device) fills the rollout storage AND the
    AMP replay buffer exactly like the eager HybridPolicyRunner step (HYBR:118-152) driven with the same actions"""
    env_e, run_e = _make_amp_runner()
    env_g, run_g = _make_amp_runner()
    assert run_g.enable_graphs()
    run_g.alg.actor_critic.load_state_dict(run_e.alg.actor_critic.state_dict())
    run_g.alg.discriminator.load_state_dict(run_e.alg.discriminator.state_dict())
    obs, crit = env_e.get_observations().clone(), env_e.get_privileged_observations().clone()
    amp_obs = env_e.get_amp_observations().clone()
    alg, ac = run_e.alg, run_e.alg.actor_critic
    saw_reset = False
    with torch.inference_mode():
        for t in range(8):
            run_g.graphs.step()
            a = run_g.graphs.actions.clone()
            tr = alg.transition
            ac.update_distribution(obs)
            tr.actions, tr.values = a, ac.evaluate(crit).detach()
            tr.actions_log_prob = ac.get_actions_log_prob(a).detach()
            tr.action_mean, tr.action_sigma = ac.action_mean.detach(), ac.action_std.detach()
            tr.observations, tr.critic_observations = obs, crit
            alg.amp_transition.observations = amp_obs
            o, p, r, d = env_e.step_device(a)
            obs, crit = o.clone(), p.clone()
            next_amp = env_e.get_amp_observations().clone()
            mask = d.unsqueeze(1)
            saw_reset |= bool(d.any())
            nxt_amp = torch.where(mask, env_e.terminal_amp_states_buf, next_amp)
            nxt_crit = torch.where(mask, env_e.termination_privileged_obs_buf, crit)
            rew = alg.discriminator.predict_amp_reward(amp_obs, nxt_amp, r, normalizer=alg.amp_normalizer)[0]
            alg.process_env_step(rew, d, env_e.extras, nxt_amp, nxt_crit)
            amp_obs = next_amp
    run_g.graphs.flush()                # the last step's post-step store waits for the next policy launch
    torch.cuda.synchronize()
    assert saw_reset
    se, sg = run_e.alg.storage, run_g.alg.storage
    for name in ("observations", "privileged_observations", "next_privileged_observations", "actions", "rewards", "dones", "values",
                 "actions_log_prob", "mu", "sigma"):
        torch.testing.assert_close(getattr(sg, name).float(), getattr(se, name).float(), rtol=1e-5, atol=1e-5, msg=name)
    be, bg = run_e.alg.amp_storage, run_g.alg.amp_storage
    assert (be.step, be.num_samples) == (bg.step, bg.num_samples) == (8 * 256, 8 * 256)
    torch.testing.assert_close(bg.states, be.states, rtol=0, atol=0)
    torch.testing.assert_close(bg.next_states, be.next_states, rtol=0, atol=0)
