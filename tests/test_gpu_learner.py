"""GPU: the hipGraph-captured rollout step writes the same storage as the eager runner step (same ops); full PPO iteration runs."""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(seed=1):
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    from isaacgymloco_amd.learn.bench_train import train_cfg_dict
    from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner
    cfg = C.aliengo_cfg()
    cfg.env.num_envs = 256
    cfg.terrain.terrain_proportions = [1.0, 0.0, 0.0, 0.0]
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=seed)
    tc = train_cfg_dict("aliengo")
    tc["runner"]["num_steps_per_env"] = 8
    torch.manual_seed(0)
    return env, HIMOnPolicyRunner(env, tc, log_dir=None, device="cuda:0")


def test_graph_rollout_matches_eager_storage():
    env_e, run_e = _make()
    env_g, run_g = _make()
    assert run_g.enable_graphs()
    run_g.alg.actor_critic.load_state_dict(run_e.alg.actor_critic.state_dict())
    obs, crit = env_e.get_observations().clone(), env_e.get_privileged_observations().clone()
    # drive both with the SAME actions: take them from the graphed runner and replay them through the eager storage path
    with torch.inference_mode():
        for t in range(8):
            run_g.graphs.step()
            a = run_g.graphs.actions.clone()
            tr = run_e.alg.transition
            ac = run_e.alg.actor_critic
            ac.update_distribution(obs)
            tr.actions = a
            tr.values = ac.evaluate(crit).detach()
            tr.actions_log_prob = ac.get_actions_log_prob(a).detach()
            tr.action_mean, tr.action_sigma = ac.action_mean.detach(), ac.action_std.detach()
            tr.observations, tr.critic_observations = obs, crit
            o, p, r, d = env_e.step_device(a)
            obs, crit = o.clone(), p.clone()
            nxt = torch.where(d.unsqueeze(1), env_e.termination_privileged_obs_buf, crit)
            run_e.alg.process_env_step(r, d, env_e.extras, nxt)
    run_g.graphs.flush()                # the last step's post-step store waits for the next policy launch
    torch.cuda.synchronize()
    se, sg = run_e.alg.storage, run_g.alg.storage
    for name in ("observations", "privileged_observations", "next_privileged_observations", "actions", "rewards", "dones", "values",
                 "actions_log_prob", "mu", "sigma"):
        torch.testing.assert_close(getattr(sg, name).float(), getattr(se, name).float(), rtol=1e-5, atol=1e-5, msg=name)


def test_full_iteration_runs_and_learns_something():
    env, run = _make(seed=3)
    run.enable_graphs()
    before = {k: v.clone() for k, v in run.alg.actor_critic.state_dict().items()}
    run.learn(2, init_at_random_ep_len=True)
    after = run.alg.actor_critic.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before)
    assert all(torch.isfinite(v).all() for v in after.values())
    assert run.last_perf["fps"] > 0


def test_amp_hybrid_runner_iteration_on_gpu():
    """aliengo_amp: HybridPolicyRunner (policy + AMP discriminator, style reward from lsim's amp_obs buffers) for two iterations."""
    import numpy as np
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    from isaacgymloco_amd.learn.bench_train import train_cfg_dict
    from isaacgymloco_amd.learn.hybrid import HybridPolicyRunner
    cfg = C.TASKS["aliengo_amp"][0]()
    cfg.env.num_envs = 256
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=2, using_amp=True)
    tc = train_cfg_dict("aliengo_amp")
    tc["runner"]["num_steps_per_env"] = 8
    torch.manual_seed(0); np.random.seed(0)
    run = HybridPolicyRunner(env, tc, log_dir=None, device="cuda:0")
    before = {k: v.clone() for k, v in run.alg.discriminator.state_dict().items()}
    run.learn(1, init_at_random_ep_len=True)              # eager rollout (reference call order)
    assert run.enable_graphs()                            # fused device-side rollout incl. style reward + replay insert
    n0 = run.alg.amp_storage.num_samples
    run.learn(2)
    assert run.alg.amp_storage.num_samples > n0
    st = run.alg.storage
    assert torch.isfinite(st.rewards).all() and float(st.rewards.abs().sum()) > 0
    after = run.alg.discriminator.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before)
    assert all(torch.isfinite(v).all() for v in after.values())
    assert all(torch.isfinite(v).all() for v in run.alg.actor_critic.state_dict().values())


def test_fused_rollout_kernels_match_torch():
    """lsim_rollout_act / lsim_rollout_post (include/lsim.h) against the torch ops they replace (HIMP:90-118, HST:92-106):
    fp32 tolerance 1e-5 on log-prob / bootstrap, exact copies, N(0,1) statistics and (seed, counter) determinism of the sampler."""
    import ctypes
    from torch.distributions import Normal
    from isaacgymloco_amd import abi, lib
    L = lib.load()
    dev = "cuda:0"
    T, N, A, O, P = 3, 1000, 12, 270, 238
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    st = {k: torch.zeros(T, N, d, device=dev) for k, d in (("observations", O), ("privileged_observations", P),
          ("next_privileged_observations", P), ("actions", A), ("values", 1), ("actions_log_prob", 1), ("mu", A), ("sigma", A), ("rewards", 1))}
    st["dones"] = torch.zeros(T, N, 1, device=dev, dtype=torch.uint8)
    S = abi.LsimRolloutStorage()
    for k, t in st.items():
        setattr(S, k, t.data_ptr())
    S.num_steps, S.num_envs, S.num_obs, S.num_priv_obs, S.num_actions = T, N, O, P, A
    idx = torch.ones(1, dtype=torch.long, device=dev)              # write row 1
    draws = torch.full((1,), 7, dtype=torch.long, device=dev)
    mean, std, values = rnd(N, A), torch.rand(A, device=dev, generator=g) + 0.3, rnd(N, 1)
    obs, priv, term = rnd(N, O), rnd(N, P), rnd(N, P)
    acts = torch.zeros(N, A, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    call = lambda out: L.lsim_rollout_act(ctypes.byref(S), idx.data_ptr(), draws.data_ptr(), mean.data_ptr(), std.data_ptr(), values.data_ptr(),
                                          obs.data_ptr(), priv.data_ptr(), 5, 0, out.data_ptr(), s)
    assert call(acts) == 0
    torch.cuda.synchronize()
    assert torch.equal(st["observations"][1], obs) and torch.equal(st["privileged_observations"][1], priv)
    assert torch.equal(st["actions"][1], acts) and torch.equal(st["mu"][1], mean) and torch.equal(st["values"][1], values)
    assert torch.equal(st["sigma"][1], std.expand(N, A))
    assert st["observations"][0].abs().sum() == 0 and st["observations"][2].abs().sum() == 0
    ref_lp = Normal(mean, std.expand(N, A)).log_prob(acts).sum(-1, keepdim=True)
    torch.testing.assert_close(st["actions_log_prob"][1], ref_lp, rtol=1e-5, atol=1e-5)
    z = ((acts - mean) / std).flatten()
    assert abs(float(z.mean())) < 0.03 and abs(float(z.var()) - 1.0) < 0.05 and float(z.abs().max()) < 6.0
    assert abs(float((z[:-1] * z[1:]).mean())) < 0.03          # neighbouring draws (the two halves of a Box-Muller pair) uncorrelated
    again = torch.zeros_like(acts)
    assert call(again) == 0
    torch.cuda.synchronize()
    assert torch.equal(again, acts)                             # same (seed, rank, counter) -> same draws
    # ---- post
    dones = torch.rand(N, device=dev, generator=g) < 0.3
    touts = dones & (torch.rand(N, device=dev, generator=g) < 0.5)
    rew = rnd(N)
    assert L.lsim_rollout_post(ctypes.byref(S), idx.data_ptr(), draws.data_ptr(), dones.data_ptr(), touts.data_ptr(), rew.data_ptr(), values.data_ptr(),
                               priv.data_ptr(), term.data_ptr(), ctypes.c_float(0.99), s) == 0
    torch.cuda.synchronize()
    assert int(idx) == 2 and int(draws) == 8
    assert torch.equal(st["next_privileged_observations"][1], torch.where(dones.unsqueeze(1), term, priv))
    assert torch.equal(st["dones"][1, :, 0], dones.to(torch.uint8))
    torch.testing.assert_close(st["rewards"][1, :, 0], rew + 0.99 * values[:, 0] * touts.float(), rtol=1e-6, atol=1e-6)
    other = torch.zeros_like(acts)
    assert call(other) == 0                                     # counter advanced -> fresh draws, now into row 2
    torch.cuda.synchronize()
    assert not torch.equal(other, acts) and torch.equal(st["actions"][2], other)
    # a full storage must not be written past its end (HST:93-94 raises in the reference)
    idx.fill_(T)
    before = st["actions"].clone()
    assert call(other) == 0
    torch.cuda.synchronize()
    assert torch.equal(st["actions"], before)
    assert L.lsim_rollout_act(None, idx.data_ptr(), draws.data_ptr(), mean.data_ptr(), std.data_ptr(), values.data_ptr(), obs.data_ptr(),
                              priv.data_ptr(), 5, 0, acts.data_ptr(), s) == abi.E_INVALID
    # ---- the by-value forms (lsim_rollout_act_at / _post_at): row 0 with counter 7 must reproduce what the device-counter form wrote into row 1
    by_val = torch.zeros_like(acts)
    assert L.lsim_rollout_act_at(ctypes.byref(S), 0, 7, mean.data_ptr(), std.data_ptr(), values.data_ptr(), obs.data_ptr(), priv.data_ptr(), 5, 0,
                                 by_val.data_ptr(), s) == 0
    assert L.lsim_rollout_post_at(ctypes.byref(S), 0, dones.data_ptr(), touts.data_ptr(), rew.data_ptr(), values.data_ptr(), priv.data_ptr(),
                                  term.data_ptr(), ctypes.c_float(0.99), s) == 0
    torch.cuda.synchronize()
    assert torch.equal(by_val, acts)
    for k in ("observations", "privileged_observations", "next_privileged_observations", "actions", "values", "actions_log_prob", "mu", "sigma", "rewards", "dones"):
        assert torch.equal(st[k][0], st[k][1]), k
    assert L.lsim_rollout_act_at(ctypes.byref(S), T, 7, mean.data_ptr(), std.data_ptr(), values.data_ptr(), obs.data_ptr(), priv.data_ptr(), 5, 0,
                                 by_val.data_ptr(), s) == abi.E_INVALID                     # row past the storage


def test_fused_gae_matches_torch_sweep():
    """lsim_rollout_gae against the reference's reverse sweep (HST:113-127) as written in storage.compute_returns' torch branch"""
    from isaacgymloco_amd.learn.storage import HIMRolloutStorage
    T, N = 24, 777
    g = torch.Generator().manual_seed(3)
    gpu = HIMRolloutStorage(N, T, [270], [238], [12], device="cuda:0")
    cpu = HIMRolloutStorage(N, T, [270], [238], [12], device="cpu")
    for name in ("rewards", "values"):
        v = torch.randn(T, N, 1, generator=g)
        getattr(cpu, name).copy_(v); getattr(gpu, name).copy_(v)
    d = (torch.rand(T, N, 1, generator=g) < 0.05).to(torch.uint8)
    cpu.dones.copy_(d); gpu.dones.copy_(d)
    last = torch.randn(N, 1, generator=g)
    cpu.compute_returns(last, 0.99, 0.95)
    gpu.compute_returns(last.to("cuda:0"), 0.99, 0.95)
    torch.testing.assert_close(gpu.returns.cpu(), cpu.returns, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(gpu.advantages.cpu(), cpu.advantages, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("k_in,n_out", [(128, 12), (128, 1), (64, 19), (45, 128), (64, 16), (128, 64), (16, 32), (7, 3),
                                        (64, 512), (256, 128), (270, 128), (130, 70), (250, 150), (512, 256)])
def test_linear_wgrad_kernel_matches_blas(k_in, n_out):
    """lsim_linear_wgrad (MFMA, csrc/ls_learn.h) against g^T x and g.sum(0) in fp64; strided x (a column slice, like the
    estimator's next_obs = critic_obs[:, 3:48]) and a ragged batch that does not fill the last wave."""
    from isaacgymloco_amd.learn.fused_linear import linear_wgrad
    g_ = torch.Generator(device="cuda:0").manual_seed(k_in * 131 + n_out)
    B = 102400 - 37
    big = torch.randn(B, k_in + 5, device="cuda:0", generator=g_)
    x = big[:, 3:3 + k_in]
    g = torch.randn(B, n_out, device="cuda:0", generator=g_)
    dw, db = linear_wgrad(x, g)
    ref_w = (g.double().t() @ x.double())
    ref_b = g.double().sum(0)
    scale = (B ** 0.5)
    assert float((dw.double() - ref_w).abs().max()) < 2e-5 * scale       # fp32 accumulation over 1e5 terms of O(1)
    assert float((db.double() - ref_b).abs().max()) < 2e-5 * scale
    dw2, db2 = linear_wgrad(x, g)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                   # fixed reduction order


def test_skinny_linear_trains_like_nn_linear():
    """SkinnyLinear is an nn.Linear (same init stream, same keys); its GPU backward matches torch's within fp32 summation error"""
    import torch.nn as nn
    from isaacgymloco_amd.learn.fused_linear import SkinnyLinear
    torch.manual_seed(0); a = nn.Linear(128, 12).to("cuda:0")
    torch.manual_seed(0); b = SkinnyLinear(128, 12).to("cuda:0")
    assert torch.equal(a.weight, b.weight) and list(a.state_dict()) == list(b.state_dict())
    x = torch.randn(8192, 128, device="cuda:0", requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    (a(x).tanh().sum()).backward()
    (b(x2).tanh().sum()).backward()
    torch.testing.assert_close(b.weight.grad, a.weight.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(b.bias.grad, a.bias.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(x2.grad, x.grad, rtol=1e-5, atol=1e-6)


def test_fused_sinkhorn_matches_torch():
    """lsim_sinkhorn against the torch statement of HES:119-133 (modules.sinkhorn's small-batch branch), fp32 tolerance 1e-4 relative"""
    from isaacgymloco_amd.learn import modules as M
    g = torch.Generator(device="cuda:0").manual_seed(5)
    for B, K in ((102400, 32), (5000, 7), (6001, 50), (4097, 64)):      # K <= 32 / K <= 64 kernels, vector and scalar row loads, ragged blocks
        z = torch.nn.functional.normalize(torch.randn(B, 16, device="cuda:0", generator=g), dim=-1)
        proto = torch.nn.functional.normalize(torch.randn(K, 16, device="cuda:0", generator=g), dim=-1)
        scores = z @ proto.T                                   # cosine similarities in [-1, 1], as in HES:96-97
        if K == 64:
            scores = torch.cat([scores, scores], dim=1)[:, :K]   # a strided view: row stride 128
        got = M.sinkhorn(scores)                               # HIP path (B >= 4096)
        Q = torch.exp(scores.double() / 0.05).T                # the reference arithmetic in fp64
        Kk, Bb = Q.shape
        Q /= Q.sum()
        for _ in range(3):
            Q /= Q.sum(dim=1, keepdim=True); Q /= Kk
            Q /= Q.sum(dim=0, keepdim=True); Q /= Bb
        ref = (Q * Bb).T
        torch.testing.assert_close(got.double(), ref, rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(got.sum(1), torch.ones(B, device="cuda:0"), rtol=1e-4, atol=1e-5)   # columns of Q*B sum to one
        assert torch.equal(got, M.sinkhorn(scores))            # deterministic


def test_checkpoint_round_trip_restores_env_curricula(tmp_path):
    """save() writes the reference's keys plus the simulator-side state the reference forgets; load() restores both"""
    env, run = _make(seed=4)
    run.enable_graphs()
    run.learn(2, init_at_random_ep_len=True)
    env.terrain_levels.fill_(3)
    path = str(tmp_path / "model_2.pt")
    run.save(path)
    d = torch.load(path, map_location="cpu", weights_only=False)
    assert {"model_state_dict", "optimizer_state_dict", "estimator_optimizer_state_dict", "iter", "infos"} <= set(d)     # HIMR:233-240
    st = env.state_dict()
    env2, run2 = _make(seed=9)
    run2.enable_graphs()
    run2.load(path)
    st2 = env2.state_dict()
    assert st2["step_counter"] == st["step_counter"] and torch.equal(st2["terrain_levels"], st["terrain_levels"])
    assert torch.equal(st2["command_ranges"], st["command_ranges"]) and run2.graphs.get_draw_counter() == run.graphs.get_draw_counter()
    for k, v in run.alg.actor_critic.state_dict().items():
        assert torch.equal(v, run2.alg.actor_critic.state_dict()[k])
    run2.learn(1)                                           # and training continues from there
    # the checkpoint is portable: plain-float learning rates, no backend flag, CPU-loadable by the reference's runner.load
    for key in ("optimizer_state_dict", "estimator_optimizer_state_dict"):
        for g in d[key]["param_groups"]:
            assert isinstance(g["lr"], float) and "fused" not in g
    cpu_opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros_like(p, device="cpu")) for p in run.alg.actor_critic.parameters()], lr=1e-3)
    cpu_opt.load_state_dict(d["optimizer_state_dict"])      # what the reference's load(load_optimizer=True) does (HIMR:246-247)
    # a load BEFORE enable_graphs() keeps the sampler's counter and applies it when the fused rollout is created
    env3, run3 = _make(seed=11)
    run3.load(path)
    assert run3.graphs is None and run3._pending_draw_counter == run.graphs.get_draw_counter()
    run3.enable_graphs()
    assert run3.graphs.get_draw_counter() == run.graphs.get_draw_counter()
    run3.learn(1)


@pytest.mark.parametrize("N", [4096, 37])
def test_fused_policy_forward_matches_torch(N):
    """lsim_policy_forward (one MFMA kernel: encoder, normalise, actor, critic) against HIMActorCritic's own forward (HAC:136-163);
    fp32 tolerance 2e-5 relative to the output scale; ragged batch that does not fill the last 16-row block"""
    from isaacgymloco_amd.learn.modules import HIMActorCritic
    from isaacgymloco_amd.learn.fused_policy import PackedHimPolicy
    torch.manual_seed(3)
    ac = HIMActorCritic(270, 238, 45, 12).to("cuda:0")
    with torch.no_grad():
        for p in ac.parameters():
            p.add_(0.05 * torch.randn_like(p))          # move away from the init's symmetry
    assert PackedHimPolicy.supported(ac)
    pk = PackedHimPolicy(ac)
    obs, priv = 2.0 * torch.randn(N, 270, device="cuda:0"), 2.0 * torch.randn(N, 238, device="cuda:0")
    mean, val = torch.empty(N, 12, device="cuda:0"), torch.empty(N, 1, device="cuda:0")
    pk.forward(obs, priv, mean, val)
    with torch.no_grad():
        ac.update_distribution(obs)
        ref_mean, ref_val = ac.action_mean, ac.evaluate(priv)
    torch.testing.assert_close(mean, ref_mean, rtol=2e-4, atol=2e-5 * float(ref_mean.abs().max()))
    torch.testing.assert_close(val, ref_val, rtol=2e-4, atol=2e-5 * float(ref_val.abs().max()))
    with torch.no_grad():                                # refresh() after a parameter change
        ac.actor[0].weight.mul_(1.5)
    pk.refresh()
    pk.forward(obs, priv, mean, val)
    with torch.no_grad():
        ac.update_distribution(obs)
    torch.testing.assert_close(mean, ac.action_mean, rtol=2e-4, atol=2e-5 * float(ac.action_mean.abs().max()))


def test_policy_act_in_one_launch_equals_forward_then_act():
    """lsim_policy_act_at (networks + sample + storage row in one kernel) against lsim_policy_forward followed by lsim_rollout_act_at:
    identical means / values / actions / storage rows, bit for bit (same Philox draws, same log-probability summation tree)"""
    import ctypes
    from isaacgymloco_amd import abi, lib
    from isaacgymloco_amd.learn import modules as M
    from isaacgymloco_amd.learn.fused_policy import PackedHimPolicy
    L = lib.load()
    dev = "cuda:0"
    for N in (4096, 37):                        # the 32-row and the 16-row kernels
        torch.manual_seed(N)
        ac = M.HIMActorCritic(270, 238, 45, 12).to(dev)
        pk = PackedHimPolicy(ac)
        T, A, O, P = 3, 12, 270, 238

        def storage():
            st = {k: torch.zeros(T, N, d, device=dev) for k, d in (("observations", O), ("privileged_observations", P),
                  ("next_privileged_observations", P), ("actions", A), ("values", 1), ("actions_log_prob", 1), ("mu", A), ("sigma", A), ("rewards", 1))}
            st["dones"] = torch.zeros(T, N, 1, device=dev, dtype=torch.uint8)
            S = abi.LsimRolloutStorage()
            for k, t in st.items():
                setattr(S, k, t.data_ptr())
            S.num_steps, S.num_envs, S.num_obs, S.num_priv_obs, S.num_actions = T, N, O, P, A
            return st, S
        obs, priv = torch.randn(N, O, device=dev), torch.randn(N, P, device=dev)
        std = torch.rand(A, device=dev) + 0.3
        s = torch.cuda.current_stream().cuda_stream
        st1, S1 = storage()
        m1, v1, a1 = torch.zeros(N, A, device=dev), torch.zeros(N, 1, device=dev), torch.zeros(N, A, device=dev)
        pk.forward(obs, priv, m1, v1)
        assert L.lsim_rollout_act_at(ctypes.byref(S1), 2, 11, m1.data_ptr(), std.data_ptr(), v1.data_ptr(), obs.data_ptr(), priv.data_ptr(), 5, 1,
                                     a1.data_ptr(), s) == 0
        st2, S2 = storage()
        m2, v2, a2 = torch.zeros(N, A, device=dev), torch.zeros(N, 1, device=dev), torch.zeros(N, A, device=dev)
        pk.forward_act(S2, 2, 11, obs, priv, std, 5, 1, m2, v2, a2)
        torch.cuda.synchronize()
        assert torch.equal(m2, m1) and torch.equal(v2, v1) and torch.equal(a2, a1)
        for k in st1:
            assert torch.equal(st2[k], st1[k]), (N, k)
        assert st2["actions"][2].abs().sum() > 0 and st2["observations"][0].abs().sum() == 0
        # ---- the previous step's post-step store folded into the next launch (lsim_policy_act_post_at) against the separate calls
        obs_n, priv_n, term = torch.randn(N, O, device=dev), torch.randn(N, P, device=dev), torch.randn(N, P, device=dev)
        dones = torch.rand(N, device=dev) < 0.3
        touts = dones & (torch.rand(N, device=dev) < 0.5)
        rew = torch.randn(N, device=dev)
        stA, SA = storage(); stB, SB = storage()
        mA, vA, aA = torch.zeros(N, A, device=dev), torch.zeros(N, 1, device=dev), torch.zeros(N, A, device=dev)
        mB, vB, aB = torch.zeros(N, A, device=dev), torch.zeros(N, 1, device=dev), torch.zeros(N, A, device=dev)
        pk.forward_act(SA, 0, 3, obs, priv, std, 5, 1, mA, vA, aA)
        assert L.lsim_rollout_post_at(ctypes.byref(SA), 0, dones.data_ptr(), touts.data_ptr(), rew.data_ptr(), vA.data_ptr(), priv_n.data_ptr(),
                                      term.data_ptr(), ctypes.c_float(0.99), s) == 0
        pk.forward_act(SA, 1, 4, obs_n, priv_n, std, 5, 1, mA, vA, aA)
        pk.forward_act(SB, 0, 3, obs, priv, std, 5, 1, mB, vB, aB)
        pk.forward_act(SB, 1, 4, obs_n, priv_n, std, 5, 1, mB, vB, aB, prev=(0, dones, touts, rew, term, 0.99))
        torch.cuda.synchronize()
        assert torch.equal(vB, vA) and torch.equal(aB, aA)
        for k in stA:
            assert torch.equal(stB[k], stA[k]), (N, k)
        assert stB["dones"][0].sum() > 0 and stB["next_privileged_observations"][0].abs().sum() > 0 and stB["rewards"][1].abs().sum() == 0


@pytest.mark.parametrize("clipped", [True, False])
def test_fused_ppo_loss_matches_torch_autograd(clipped):
    """lsim_ppo_loss (forward + backward + KL in one pass) against the torch statement of HIMP:136-176 and its autograd gradients"""
    from torch.distributions import Normal
    from isaacgymloco_amd.learn.fused_linear import ppo_loss_hip
    g = torch.Generator(device="cuda:0").manual_seed(11)
    B, A, clip, cv, ce = 20000, 12, 0.2, 1.0, 0.01
    rnd = lambda *s: torch.randn(*s, device="cuda:0", generator=g)
    mu = (0.5 * rnd(B, A)).requires_grad_(True)
    std = (0.5 + torch.rand(A, device="cuda:0", generator=g)).requires_grad_(True)
    value = rnd(B, 1).requires_grad_(True)
    old_mu, old_sigma = mu.detach() + 0.1 * rnd(B, A), (std.detach() * (1 + 0.1 * rnd(A).clamp(-2, 2))).expand(B, A).contiguous()
    actions = old_mu + old_sigma * rnd(B, A)
    old_logp = Normal(old_mu, old_sigma).log_prob(actions).sum(-1, keepdim=True)
    adv, returns, tv = rnd(B, 1), rnd(B, 1), value.detach() + 0.3 * rnd(B, 1)

    def torch_loss(mu, std, value):
        sigma = mu * 0.0 + std
        dist = Normal(mu, sigma)
        logp, ent = dist.log_prob(actions).sum(-1), dist.entropy().sum(-1)
        a = adv.squeeze()
        ratio = torch.exp(logp - old_logp.squeeze())
        sur = torch.max(-a * ratio, -a * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
        if clipped:
            vc = tv + (value - tv).clamp(-clip, clip)
            vl = torch.max((value - returns).pow(2), (vc - returns).pow(2)).mean()
        else:
            vl = (returns - value).pow(2).mean()
        kl = torch.sum(torch.log(sigma / old_sigma + 1e-5) + (old_sigma ** 2 + (old_mu - mu) ** 2) / (2 * sigma ** 2) - 0.5, -1).mean()
        return sur + cv * vl - ce * ent.mean(), sur, vl, ent.mean(), kl

    ref = torch_loss(mu, std, value)
    ref[0].backward()
    ref_g = (mu.grad.clone(), std.grad.clone(), value.grad.clone())
    for t in (mu, std, value):
        t.grad = None
    sigma = mu * 0.0 + std
    loss, st = ppo_loss_hip(mu, sigma, value, actions, old_logp, adv, returns, tv, old_mu, old_sigma, clip, cv, ce, clipped)
    loss.backward()
    torch.testing.assert_close(loss, ref[0], rtol=2e-5, atol=2e-6)
    for got, want in zip(st, ref[1:]):
        torch.testing.assert_close(got, want.detach(), rtol=2e-5, atol=2e-6)
    # the loss is piecewise: a sample whose ratio (or value error) sits within fp32 rounding of a clip boundary may take the other
    # branch in the two implementations, so those few rows are compared on the loss only
    with torch.no_grad():
        sg = (mu * 0.0 + std)
        ratio = torch.exp(Normal(mu, sg).log_prob(actions).sum(-1) - old_logp.squeeze())
        near = ((ratio - (1 - clip)).abs() < 1e-4) | ((ratio - (1 + clip)).abs() < 1e-4)
        if clipped:
            dv = (value - tv).squeeze()
            vc = tv + (value - tv).clamp(-clip, clip)
            l_gap = ((value - returns) ** 2 - (vc - returns) ** 2).squeeze().abs()
            near |= ((dv.abs() - clip).abs() < 1e-4) | ((l_gap < 1e-5) & (dv.abs() > clip))     # exact ties (unclipped rows) agree
    keep = ~near
    assert int(near.sum()) < 0.01 * B
    scale = float(ref_g[0].abs().max())
    torch.testing.assert_close(mu.grad[keep], ref_g[0][keep], rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(value.grad[keep], ref_g[2][keep], rtol=1e-4, atol=1e-5 * float(ref_g[2].abs().max()))
    if int(near.sum()) == 0:
        torch.testing.assert_close(std.grad, ref_g[1], rtol=2e-4, atol=1e-5 * float(ref_g[1].abs().max()))
    else:   # std's gradient sums over the batch: bound the contribution of the excluded rows
        torch.testing.assert_close(std.grad, ref_g[1], rtol=1e-2, atol=float(near.sum()) * scale)


def test_fused_rollout_uses_the_updated_policy():
    """after update() the next rollout must act with the NEW weights (the fused policy kernel reads packed copies of them)"""
    env, run = _make(seed=6)
    assert run.enable_graphs() and run.graphs.packed is not None
    run.learn(2, init_at_random_ep_len=True)
    ac = run.alg.actor_critic
    run.graphs.step()
    torch.cuda.synchronize()
    with torch.no_grad():
        ac.update_distribution(run.alg.storage.observations[0])
        want = ac.action_mean
    torch.testing.assert_close(run.alg.storage.mu[0], want, rtol=2e-4, atol=2e-5 * float(want.abs().max()))


def test_device_lr_and_fused_adam_match_the_host_rule():
    """enable_device_lr(): lsim_adaptive_lr + fused Adam reading a device scalar give the same learning-rate trajectory and (to fp32
    rounding of a different Adam kernel) the same parameters as the reference-style host rule with .item() per minibatch"""
    env_a, run_a = _make(seed=8)
    env_b, run_b = _make(seed=8)
    assert run_a.enable_graphs() and run_a.alg._lr_t is not None
    run_b.alg.enable_device_lr = lambda: False
    assert run_b.enable_graphs() and run_b.alg._lr_t is None
    run_b.alg.actor_critic.load_state_dict(run_a.alg.actor_critic.state_dict())
    for it in range(3):
        torch.manual_seed(100 + it); run_a.learn(1)
        torch.manual_seed(100 + it); run_b.learn(1)
        assert abs(run_a.alg.learning_rate - run_b.alg.learning_rate) <= 1e-6 * run_b.alg.learning_rate, it
    pa, pb = run_a.alg.actor_critic.state_dict(), run_b.alg.actor_critic.state_dict()
    for k in pa:
        torch.testing.assert_close(pa[k], pb[k], rtol=5e-3, atol=5e-4, msg=k)
    path = "/tmp/lsim_ckpt_lr.pt"
    run_a.save(path); run_a.load(path)
    assert all(g["lr"] is run_a.alg._lr_t for g in run_a.alg.optimizer.param_groups)
    run_a.learn(1)


@pytest.mark.parametrize("k_in,n_out", [(64, 512), (512, 256), (256, 128), (270, 128), (45, 128), (128, 64)])
def test_fused_linear_elu_backward_matches_torch(k_in, n_out):
    """_LinearEluFn (lsim_linear_elu_wgrad: elu_backward folded into the weight-gradient kernel) against nn.Linear + nn.ELU autograd"""
    import torch.nn as nn
    from isaacgymloco_amd.learn.fused_linear import HimMLP
    torch.manual_seed(k_in + n_out)
    B = 20480 + 12
    ref = nn.Sequential(nn.Linear(k_in, n_out), nn.ELU(), nn.Linear(n_out, 5)).to("cuda:0")
    fus = HimMLP(nn.Linear(k_in, n_out), nn.ELU(), nn.Linear(n_out, 5)).to("cuda:0")
    fus.load_state_dict(ref.state_dict())
    big = torch.randn(B, k_in + 3, device="cuda:0")
    xa = big[:, 1:1 + k_in].clone().requires_grad_(True) if k_in % 2 else big[:, :k_in].clone().requires_grad_(True)
    xb = xa.detach().clone().requires_grad_(True)
    w = torch.randn(B, 5, device="cuda:0")
    (ref(xa) * w).sum().backward()
    (fus(xb) * w).sum().backward()
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), fus.named_parameters()):
        scale = float(p1.grad.abs().max())
        torch.testing.assert_close(p2.grad, p1.grad, rtol=2e-4, atol=2e-5 * scale, msg=n1)
    torch.testing.assert_close(xb.grad, xa.grad, rtol=2e-4, atol=2e-5 * float(xa.grad.abs().max()))
    # a first layer: the input needs no gradient, so the kernel is called with grad_pre = NULL and never writes it -- same parameter gradients
    fus.zero_grad()
    (fus(xb.detach()) * w).sum().backward()
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), fus.named_parameters()):
        torch.testing.assert_close(p2.grad, p1.grad, rtol=2e-4, atol=2e-5 * float(p1.grad.abs().max()), msg=n1 + " (no input gradient)")


@pytest.mark.parametrize("rows,k_in,n_out", [(102400, 64, 512), (20480 + 12, 512, 256), (20480 + 12, 256, 128), (4096, 238, 512), (4099, 270, 128), (1000, 45, 128),
                                             (161, 128, 64), (7, 33, 12)])
def test_fused_linear_elu_forward_matches_torch(rows, k_in, n_out):
    """lsim_linear_elu_forward (fp32 MFMA product with bias and ELU applied to the accumulators) against F.elu(F.linear) in fp64, through the C-ABI: every
    hidden-layer shape of the learner (16-byte, 8-byte and 4-byte aligned rows; partial sample, feature and K tiles), strided input rows, with and without bias.
    Tolerance: 2e-5 absolute on O(1) activations (an fp32 sum of <= 512 products; the order of the additions differs from BLAS)"""
    import torch.nn.functional as F
    from isaacgymloco_amd import abi, lib
    from isaacgymloco_amd.learn.fused_linear import linear_elu_forward
    L = lib.load()
    g = torch.Generator(device="cuda:0").manual_seed(rows + k_in + n_out)
    big = torch.randn(rows, k_in + 5, device="cuda:0", generator=g)
    W = torch.randn(n_out, k_in, device="cuda:0", generator=g) / k_in ** 0.5
    b = torch.randn(n_out, device="cuda:0", generator=g) * 0.3
    padded = torch.full((rows, (k_in + 3) // 4 * 4 + 4), float("nan"), device="cuda:0")       # rows further apart than their width: what lies behind a row is not read as data
    padded[:, :k_in] = big[:, :k_in]
    for x, bias in ((big[:, :k_in].contiguous(), b), (big[:, 1:1 + k_in], None), (padded[:, :k_in], b)):    # the second: rows k_in + 5 apart, starting 4 bytes into the allocation
        out = torch.full((rows, n_out), float("nan"), device="cuda:0")
        rc = L.lsim_linear_elu_forward(x.data_ptr(), x.stride(0), W.data_ptr(), bias.data_ptr() if bias is not None else None, rows, k_in, n_out, out.data_ptr(),
                                       out.stride(0), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        want = F.elu(F.linear(x.double(), W.double(), bias.double() if bias is not None else None))
        assert torch.isfinite(out).all()
        assert float((out.double() - want).abs().max()) < 2e-5
    # the learner's entry point: the library kernel under LSIM_ELU_FORWARD=all, BLAS + ELU under =0, the same values either way
    os.environ["LSIM_ELU_FORWARD"] = "all"
    try:
        z1 = linear_elu_forward(big[:, :k_in], W, b)
        os.environ["LSIM_ELU_FORWARD"] = "0"
        z0 = linear_elu_forward(big[:, :k_in], W, b)
    finally:
        os.environ.pop("LSIM_ELU_FORWARD", None)
    assert float((z1 - z0).abs().max()) < 2e-5
    # n_out % 4 != 0 is the caller's to run through BLAS
    assert L.lsim_linear_elu_forward(big.data_ptr(), big.stride(0), W.data_ptr(), None, rows, k_in, 10, out.data_ptr(), 12, torch.cuda.current_stream().cuda_stream) == abi.E_UNSUPPORTED


@pytest.mark.parametrize("rows,cols", [(32, 16), (64, 32), (5, 300), (1, 4096), (3, 1)])
def test_normalize_rows_matches_torch(rows, cols):
    """lsim_normalize_rows = F.normalize(w, dim=-1) in place (HIMEstimator's prototype normalisation before every loss, HES:83-86), one wave per row"""
    import torch.nn.functional as F
    from isaacgymloco_amd import lib
    g = torch.Generator(device="cuda:0").manual_seed(rows * 131 + cols)
    w = torch.randn(rows, cols, device="cuda:0", generator=g) * 3.0
    if rows > 2:
        w[1].zero_()                                   # a zero row stays zero (the 1e-12 floor of F.normalize)
    want = F.normalize(w.double(), dim=-1, p=2, eps=1e-12)
    lib.check(lib.load().lsim_normalize_rows(w.data_ptr(), rows, cols, 1e-12, torch.cuda.current_stream().cuda_stream), what="lsim_normalize_rows")
    assert float((w.double() - want).abs().max()) < 5e-7


def test_reference_style_step_tuple_and_strict_runner_path():
    """LeggedRobot.step (LR:122-176): the 7-tuple (8 with AMP) with data-dependent shapes, consistent with step_device on a twin env; and
    the runner's strict path (fast=False: the reference's call sequence incl. the next_critic_obs[termination_ids] patch, HIMR:119-123)
    fills the same storage as the fast path."""
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    from isaacgymloco_amd.learn.bench_train import train_cfg_dict
    from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner

    def mk(amp=False):
        cfg = C.TASKS["aliengo_amp" if amp else "aliengo"][0]()
        cfg.env.num_envs = 128
        cfg.env.episode_length_s = 0.3          # time-outs inside the window
        return LeggedRobot(cfg, sim_device="cuda:0", seed=12, using_amp=amp)
    a, b = mk(), mk()
    a.reset(); b.reset()
    g = torch.Generator(device="cuda:0").manual_seed(2)
    saw_reset = False
    for _ in range(30):
        act = torch.randn(128, 12, device="cuda:0", generator=g)
        obs, priv, rew, done, extras, ids, term_priv = a.step(act)
        o2, p2, r2, d2 = b.step_device(act)
        assert obs.shape == (128, 270) and priv.shape == (128, 238) and rew.shape == (128,) and done.dtype == torch.bool
        assert torch.equal(obs, o2) and torch.equal(priv, p2) and torch.equal(rew, r2) and torch.equal(done, d2)
        assert ids.dtype == torch.int64 and torch.equal(ids, d2.nonzero(as_tuple=False).flatten())
        assert term_priv.shape == (len(ids), 238) and torch.equal(term_priv, b.termination_privileged_obs_buf[ids])
        assert "time_outs" in extras
        saw_reset |= len(ids) > 0
    assert saw_reset
    amp = mk(amp=True)
    amp.reset()
    out = amp.step(torch.zeros(128, 12, device="cuda:0"))
    assert len(out) == 8 and out[7].shape == (len(out[5]), 30)
    # strict vs fast runner path: same storage after one rollout (same seeds -> same simulator trajectory given the same actions)
    tc = train_cfg_dict("aliengo")
    tc["runner"]["num_steps_per_env"] = 6
    runs = []
    for fast in (True, False):
        env = mk()
        torch.manual_seed(0)
        run = HIMOnPolicyRunner(env, tc, log_dir=None, device="cuda:0", fast=fast)
        torch.manual_seed(1)
        obs, crit = env.get_observations().clone(), env.get_privileged_observations().clone()
        with torch.inference_mode():
            for _ in range(6):
                obs, crit, *_ = run._rollout_step(obs, crit)
        runs.append(run.alg.storage)
    for name in ("observations", "privileged_observations", "next_privileged_observations", "actions", "rewards", "dones", "values"):
        torch.testing.assert_close(getattr(runs[0], name).float(), getattr(runs[1], name).float(), rtol=1e-6, atol=1e-6, msg=name)


def test_fused_estimator_loss_matches_torch_autograd():
    """lsim_estimator_loss (normalise, scores, Sinkhorn, log-softmax, swap + regression losses, backward) against the fp64 torch statement
    of HES:76-108 with autograd; losses to 1e-5 relative, gradients to 1e-3 of their scale"""
    from isaacgymloco_amd.learn.fused_linear import estimator_loss_hip
    g = torch.Generator(device="cuda:0").manual_seed(11)
    for B, D, K in ((102400, 16, 32), (5000, 16, 32), (777, 9, 50)):
        big = torch.randn(B, 3 + D + 5, device="cuda:0", generator=g)
        enc = big[:, :3 + D].clone().requires_grad_(True)
        tgt = torch.randn(B, D, device="cuda:0", generator=g).requires_grad_(True)
        proto = torch.nn.functional.normalize(torch.randn(K, D, device="cuda:0", generator=g), dim=-1).requires_grad_(True)
        vel = big[:, 3 + D:3 + D + 3]                               # a strided view, as in HIMEstimator.losses
        total, parts = estimator_loss_hip(enc, tgt, proto, vel, 3.0)
        total.backward()
        e64, t64, p64 = (t.detach().double().requires_grad_(True) for t in (enc, tgt, proto))
        z_s = torch.nn.functional.normalize(e64[:, 3:], dim=-1)
        z_t = torch.nn.functional.normalize(t64, dim=-1)
        S_s, S_t = z_s @ p64.T, z_t @ p64.T

        def sk(scores):
            Q = torch.exp(scores.detach() / 0.05).T
            Q /= Q.sum()
            for _ in range(3):
                Q /= Q.sum(dim=1, keepdim=True); Q /= Q.shape[0]
                Q /= Q.sum(dim=0, keepdim=True); Q /= Q.shape[1]
            return (Q * Q.shape[1]).T
        swap = -0.5 * (sk(S_s) * torch.log_softmax(S_t / 3.0, -1) + sk(S_t) * torch.log_softmax(S_s / 3.0, -1)).mean()
        est = torch.nn.functional.mse_loss(e64[:, :3], vel.double())
        (est + swap).backward()
        torch.testing.assert_close(parts.double(), torch.stack([est, swap]).detach(), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(total.double(), (est + swap).detach(), rtol=1e-5, atol=1e-7)
        for got, ref in ((enc.grad, e64.grad), (tgt.grad, t64.grad), (proto.grad, p64.grad)):
            scale = ref.abs().max()
            assert ((got.double() - ref).abs().max() / scale) < 1e-3, ((got.double() - ref).abs().max(), scale)
        enc2 = enc.detach().clone().requires_grad_(True)
        total2, _ = estimator_loss_hip(enc2, tgt.detach(), proto.detach(), vel, 3.0)
        total2.backward()
        assert torch.equal(total2, total) and torch.equal(enc2.grad, enc.grad)        # deterministic


def test_estimator_update_uses_the_fused_loss_and_matches_the_torch_path():
    """HIMEstimator.update on the GPU (fused loss head) moves the parameters like the torch statement of the same losses"""
    import copy
    from isaacgymloco_amd.learn import modules as M
    torch.manual_seed(3)
    est = M.HIMEstimator(6, 45).to("cuda:0")
    ref = copy.deepcopy(est)
    g = torch.Generator(device="cuda:0").manual_seed(2)
    hist, nxt = torch.randn(8192, 270, device="cuda:0", generator=g), torch.randn(8192, 45 + 3 + 187, device="cuda:0", generator=g)
    e1, s1 = est.update(hist, nxt)
    # the torch statement, on the copy
    import torch.nn.functional as F
    n = ref.num_one_step_obs
    vel, next_obs = nxt[:, n:n + 3], nxt[:, 3:n + 3]
    out = ref.encoder(hist)
    z_s, z_t = F.normalize(out[:, 3:], dim=-1), F.normalize(ref.target(next_obs), dim=-1)
    with torch.no_grad():
        ref.proto.weight.copy_(F.normalize(ref.proto.weight.data.clone(), dim=-1))
    S_s, S_t = z_s @ ref.proto.weight.T, z_t @ ref.proto.weight.T
    with torch.no_grad():
        q_s, q_t = M.sinkhorn(S_s), M.sinkhorn(S_t)
    swap = -0.5 * (q_s * F.log_softmax(S_t / ref.temperature, -1) + q_t * F.log_softmax(S_s / ref.temperature, -1)).mean()
    e2 = F.mse_loss(out[:, :3], vel)
    ref.optimizer.zero_grad()
    (e2 + swap).backward()
    torch.nn.utils.clip_grad_norm_(ref.parameters(), ref.max_grad_norm)
    ref.optimizer.step()
    torch.testing.assert_close(e1, e2.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(s1, swap.detach(), rtol=1e-4, atol=1e-7)
    for (k, a), b in zip(est.named_parameters(), ref.parameters()):     # the (clipped) gradients both optimisers consumed
        if k.startswith("target.") or k.startswith("encoder.") or k.startswith("proto."):
            assert a.grad is not None and b.grad is not None, k
            assert (a.grad - b.grad).abs().max() <= 1e-3 * b.grad.abs().max() + 1e-9, k
    moved = sum(float((a - b).abs().gt(2e-5).float().mean()) for a, b in zip(est.parameters(), ref.parameters()))
    assert moved < 1e-2        # one Adam step of lr 1e-3 (sign-like): only gradients at the noise floor may step the other way


def test_fused_clip_adam_step_matches_torch():
    """lsim_adam_clip_step on the optimizer's own tensors against clip_grad_norm_ + torch.optim.Adam(fused=True).step(): parameters, moments,
    step counters and the clipped gradients, with the clip active (max_norm 1) and inactive (1e9), host and device learning rate"""
    import copy
    from isaacgymloco_amd.learn.fused_linear import adam_clip_step_hip
    torch.manual_seed(0)
    for max_norm, dev_lr in ((1.0, True), (1e9, False)):
        net = torch.nn.Sequential(torch.nn.Linear(270, 128), torch.nn.ELU(), torch.nn.Linear(128, 64), torch.nn.ELU(), torch.nn.Linear(64, 19)).to("cuda:0")
        ref = copy.deepcopy(net)
        lr_a = torch.tensor(1e-3, device="cuda:0") if dev_lr else 1e-3
        lr_b = torch.tensor(1e-3, device="cuda:0") if dev_lr else 1e-3
        oa = torch.optim.Adam(net.parameters(), lr=lr_a, fused=True)
        ob = torch.optim.Adam(ref.parameters(), lr=lr_b, fused=True)
        g = torch.Generator(device="cuda:0").manual_seed(1)
        used = 0
        for it in range(6):
            x = torch.randn(4096, 270, device="cuda:0", generator=g)
            for n_, o_ in ((net, oa), (ref, ob)):
                o_.zero_grad()
                (n_(x).square().mean() * 50.0).backward()
            if dev_lr and it == 3:
                lr_a.mul_(0.5); lr_b.mul_(0.5)                       # the adaptive rule rewrites the device scalar between steps
            if adam_clip_step_hip(oa, max_norm):
                used += 1
            else:                                                    # first step: torch creates the state
                torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm)
                oa.step()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm)
            ob.step()
            # the two networks drift apart by rounding (different summation order of the norm, then Adam's normalised step amplifies it where
            # the second moment is tiny): tight for almost all elements, bounded for the rest
            def close(a, b, tight, loose):
                d = (a - b).abs()
                scale = b.abs().max().clamp_min(1e-30)
                assert float(d.max() / scale) < loose, (float(d.max()), float(scale))
                assert float((d / scale > tight).float().mean()) < 0.02 or a.numel() < 64 and float(d.max() / scale) < 10 * tight
            for pa, pb in zip(net.parameters(), ref.parameters()):
                close(pa.grad.detach(), pb.grad.detach(), 1e-4, 2e-3)
                close(pa.detach(), pb.detach(), 1e-4, 2e-3)
                sa, sb = oa.state[pa], ob.state[pb]
                assert float(sa["step"]) == float(sb["step"]) == it + 1
                close(sa["exp_avg"], sb["exp_avg"], 1e-4, 2e-3)
                close(sa["exp_avg_sq"], sb["exp_avg_sq"], 1e-4, 2e-3)
        assert used == 5


def test_fused_clip_adam_step_with_weight_decay_groups_matches_torch():
    """lsim_adam_clip_step_ex as HybridPPO uses it (HYBP:86-92, 270-273): ONE Adam over three parameter groups that differ in weight decay,
    gradient clipping over the first group only -- against clip_grad_norm_(first group) + torch.optim.Adam.step()"""
    import copy
    from isaacgymloco_amd.learn.fused_linear import adam_clip_step_hip
    torch.manual_seed(0)
    mk = lambda: torch.nn.ModuleList([torch.nn.Linear(64, 128), torch.nn.Linear(60, 256), torch.nn.Linear(256, 1)]).to("cuda:0")
    net = mk()
    ref = copy.deepcopy(net)
    groups = lambda m: [{"params": m[0].parameters()}, {"params": m[1].parameters(), "weight_decay": 10e-4}, {"params": m[2].parameters(), "weight_decay": 10e-2}]
    lr_a, lr_b = torch.tensor(1e-3, device="cuda:0"), torch.tensor(1e-3, device="cuda:0")
    oa = torch.optim.Adam(groups(net), lr=lr_a, fused=True)
    ob = torch.optim.Adam(groups(ref), lr=lr_b, fused=True)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    used = 0
    for it in range(5):
        x1, x2 = torch.randn(4096, 64, device="cuda:0", generator=g), torch.randn(4096, 60, device="cuda:0", generator=g)
        for m, o in ((net, oa), (ref, ob)):
            o.zero_grad()
            (m[0](x1).square().mean() * 30.0 + m[2](torch.relu(m[1](x2))).square().mean()).backward()
        if adam_clip_step_hip(oa, 1.0, clip_params=list(net[0].parameters())):
            used += 1
        else:
            torch.nn.utils.clip_grad_norm_(net[0].parameters(), 1.0)
            oa.step()
        torch.nn.utils.clip_grad_norm_(ref[0].parameters(), 1.0)
        ob.step()
        for pa, pb in zip(net.parameters(), ref.parameters()):
            scale = float(pb.detach().abs().max())
            assert float((pa.detach() - pb.detach()).abs().max()) < 2e-3 * scale
            assert float((pa.detach() - pb.detach()).abs().gt(1e-4 * scale).float().mean()) < 0.02
            gs = float(pb.grad.abs().max())                                        # clipped in place for group 0, untouched (no decay written back) for the others;
            assert float((pa.grad - pb.grad).abs().max()) < 2e-3 * gs              # the two copies drift apart by rounding, amplified by Adam's normalised step
            assert float(oa.state[pa]["step"]) == float(ob.state[pb]["step"]) == it + 1
    assert used == 4
    # the decay really acts: the head's weights shrink against a run without it
    assert float(net[2].weight.norm()) < float(mk()[2].weight.norm()) * 1.5


def test_fused_clip_adam_step_follows_a_replaced_optimiser_state():
    """the cached pointer tables of adam_clip_step_hip must not outlive the state tensors they point to.  Two steps
    with a gradient arena (every gradient pointer repeats), then optimizer.load_state_dict() with DIFFERENT moments (what runner.load() and
    enable_device_lr's rebuild do): the next fused step must read and update the loaded moments, exactly like torch's Adam on a twin."""
    import copy
    from isaacgymloco_amd.learn.fused_linear import adam_clip_step_hip
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.ELU(), torch.nn.Linear(128, 19)).to("cuda:0")
    ref = copy.deepcopy(net)
    oa = torch.optim.Adam(net.parameters(), lr=torch.tensor(1e-3, device="cuda:0"), fused=True)
    ob = torch.optim.Adam(ref.parameters(), lr=torch.tensor(1e-3, device="cuda:0"), fused=True)
    grads = [torch.zeros_like(p) for p in net.parameters()]           # persistent gradient storage: the pointers repeat from step to step
    g = torch.Generator(device="cuda:0").manual_seed(1)

    def backward_both():
        x = torch.randn(4096, 64, device="cuda:0", generator=g)
        for n_, o_ in ((net, oa), (ref, ob)):
            o_.zero_grad()
            (n_(x).square().mean() * 20.0).backward()
        for p, buf in zip(net.parameters(), grads):
            buf.copy_(p.grad); p.grad = buf

    def step_both():
        backward_both()
        if not adam_clip_step_hip(oa, 1.0):
            torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0); oa.step()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0); ob.step()
    for _ in range(3):
        step_both()
    # a checkpoint whose moments differ: scaled copies, step counters moved on
    sd = copy.deepcopy(ob.state_dict())
    for st in sd["state"].values():
        st["exp_avg"].mul_(-3.0); st["exp_avg_sq"].mul_(7.0); st["step"].add_(10.0)
    old_ptrs = [oa.state[p]["exp_avg"].data_ptr() for p in net.parameters()]
    oa.load_state_dict(copy.deepcopy(sd)); ob.load_state_dict(copy.deepcopy(sd))
    for o_ in (oa, ob):
        for grp in o_.param_groups:
            grp["lr"] = torch.tensor(1e-3, device="cuda:0")
    assert [oa.state[p]["exp_avg"].data_ptr() for p in net.parameters()] != old_ptrs      # load_state_dict really replaced the tensors
    backward_both()
    assert adam_clip_step_hip(oa, 1.0)
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0); ob.step()
    for pa, pb in zip(net.parameters(), ref.parameters()):
        sa, sb = oa.state[pa], ob.state[pb]
        assert float(sa["step"]) == float(sb["step"]) == 14.0
        for k in ("exp_avg", "exp_avg_sq"):
            scale = float(sb[k].abs().max())
            assert float((sa[k] - sb[k]).abs().max()) < 2e-3 * scale, k          # the loaded moments were the ones updated
        assert float((pa.detach() - pb.detach()).abs().max()) < 2e-3 * float(pb.detach().abs().max())
    # and an optimizer that goes away takes its table with it
    from isaacgymloco_amd.learn import fused_linear as FL
    oid = id(oa)
    assert oid in FL._adam_tables
    del oa
    import gc; gc.collect()
    assert oid not in FL._adam_tables


def test_amp_checkpoint_load_keeps_the_fused_optimiser_path(tmp_path):
    """after load() the param groups must carry the optimiser's own backend flags again (Adam.__setstate__ fills a
    portable checkpoint's missing `fused` with None) and the resumed AMP run must keep training"""
    import numpy as np
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    from isaacgymloco_amd.learn.bench_train import train_cfg_dict
    from isaacgymloco_amd.learn.hybrid import HybridPolicyRunner

    def make(seed):
        cfg = C.TASKS["aliengo_amp"][0]()
        cfg.env.num_envs = 256
        env = LeggedRobot(cfg, sim_device="cuda:0", seed=seed, using_amp=True)
        tc = train_cfg_dict("aliengo_amp")
        tc["runner"]["num_steps_per_env"] = 8
        torch.manual_seed(0); np.random.seed(0)
        return HybridPolicyRunner(env, tc, log_dir=None, device="cuda:0")
    run = make(2)
    assert run.enable_graphs()
    run.learn(2, init_at_random_ep_len=True)
    path = str(tmp_path / "model_2.pt")
    run.save(path)
    run2 = make(5)
    assert run2.enable_graphs()
    run2.load(path)
    for opt in (run2.alg.optimizer, run2.alg.actor_critic.estimator.optimizer):
        for g in opt.param_groups:
            assert g["fused"] is True and torch.is_tensor(g["lr"]) and g["lr"].is_cuda
    assert [g.get("weight_decay") for g in run2.alg.optimizer.param_groups] == [0, 10e-4, 10e-2]
    for k, v in run.alg.discriminator.state_dict().items():
        assert torch.equal(v, run2.alg.discriminator.state_dict()[k])
    run2.learn(1)
    assert all(torch.isfinite(v).all() for v in run2.alg.actor_critic.state_dict().values())


def test_fused_actor_input_matches_torch():
    """lsim_actor_input against the torch statement of HAC:136-141 (slice, F.normalize, cat), contiguous and strided histories"""
    from isaacgymloco_amd.learn import modules as M
    torch.manual_seed(2)
    ac = M.HIMActorCritic(270, 238, 45, 12).to("cuda:0")
    g = torch.Generator(device="cuda:0").manual_seed(4)
    for B in (102400, 4096, 37):
        wide = torch.randn(B, 300, device="cuda:0", generator=g)
        for hist in (wide[:, :270].contiguous(), wide[:, :270]):
            got = ac._actor_input(hist)
            with torch.no_grad():
                vel, latent = ac.estimator(hist)
                ref = torch.cat((hist[:, :45], vel, latent), dim=-1)
            assert got.shape == ref.shape == (B, 64)
            torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)


def test_play_style_loop_runs_on_the_env_surface():
    """the evaluation loop of the reference's play.py (play.py:70-158) against this env: its config overrides (no termination bodies, no
    heading command, no command curriculum, one resampling per 10 000 s), commands written by the host every step, the inference policy of a
    runner, the attributes its logger reads -- and a by-hand reset_idx of a few robots in the middle"""
    import numpy as np
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    from isaacgymloco_amd.learn.bench_train import train_cfg_dict
    from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner
    cfg = C.TASKS["aliengo"][0]()
    cfg.env.num_envs = 50                                   # play.py:64
    cfg.terrain.num_rows, cfg.terrain.num_cols = 5, 5       # play.py:65-66
    cfg.terrain.curriculum = False
    cfg.noise.add_noise = False
    cfg.domain_rand.randomize_friction = False
    cfg.domain_rand.push_robots = False
    cfg.domain_rand.disturbance = False
    cfg.domain_rand.randomize_payload_mass = False
    cfg.asset.terminate_after_contacts_on = []              # play.py:81
    cfg.commands.heading_command = False
    cfg.commands.curriculum = False
    cfg.commands.resampling_time = 10000.0
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)
    assert env.viewer is None and len(env.termination_contact_indices) == 0
    env.commands[:, 0], env.commands[:, 1], env.commands[:, 2] = 1.0, 0.0, 0.5
    torch.manual_seed(0)
    runner = HIMOnPolicyRunner(env, train_cfg_dict("aliengo"), log_dir=None, device="cuda:0")
    policy = runner.get_inference_policy(device=env.device)
    obs = env.get_observations()
    env.set_camera(np.array([3.0, 3.0, 3.0]), np.array([10.0, 10.0, 0.0]))
    x_vel = 2.0 * torch.rand(env.num_envs, device=env.device) - 1.0
    for i in range(40):
        actions = policy(obs.detach())
        env.commands[:, 0] = x_vel
        env.commands[:, 1] = 0.0
        env.commands[:, 2] = 0.5
        obs, _, rews, dones, infos, *_ = env.step(actions.detach())
        assert obs.shape == (50, 270) and bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rews).all())
        survivors = ~dones
        assert torch.equal(env.commands[survivors, 0], x_vel[survivors])       # host-written commands survive the step (no resampling, no heading rule)
        assert torch.allclose(env.commands[survivors, 2], torch.full_like(x_vel[survivors], 0.5))
        for v in (env.dof_pos[0, 1], env.dof_vel[0, 1], env.torques[0, 1], env.base_lin_vel[0, 0], env.base_ang_vel[0, 2]):
            assert np.isfinite(v.item())
        assert env.contact_forces[0, env.feet_indices, 2].shape == (4,)
        assert isinstance(infos["episode"], dict)
        _ = actions[0, 1].item() * env.cfg.control.action_scale + env.default_dof_pos[0, 1].item()
        if i == 20:
            ep = env.episode_length_buf.clone()
            env.reset_idx(torch.tensor([3, 17, 42], device=env.device))
            assert bool((env.episode_length_buf[[3, 17, 42]] == 0).all())
            keep = torch.ones(50, dtype=torch.bool, device=env.device); keep[[3, 17, 42]] = False
            assert torch.equal(env.episode_length_buf[keep], ep[keep])
    assert not bool(env.reset_buf.all())


def test_gradient_arena_changes_nothing_but_where_the_gradients_live(monkeypatch):
    """fused_linear.GradArena (persistent flat gradient buckets: the weight-gradient kernels write dW / db into the parameters' slices, autograd
    adopts the slices as .grad) against the same two iterations with LSIM_GRAD_ARENA=0: identical weights and optimiser state bit for bit,
    and with the arena every gradient of the actor-critic lives inside one of its flat buffers after an update."""
    from isaacgymloco_amd.learn import fused_linear as FL

    def run(arena):
        monkeypatch.setenv("LSIM_GRAD_ARENA", "1" if arena else "0")
        FL.set_grad_arena(None)
        env, r = _make(seed=5)
        r.enable_graphs()
        r.learn(2, init_at_random_ep_len=False)
        sd = {k: v.clone() for k, v in r.alg.actor_critic.state_dict().items()}
        return r, sd
    ra, a = run(True)
    arena = FL._arena
    assert arena is not None and {"estimator", "ppo"} <= set(arena.buckets)
    spans = [(b.flat.data_ptr(), b.flat.data_ptr() + 4 * b.flat.numel()) for b in arena.buckets.values()]
    grads = [p.grad for p in ra.alg.actor_critic.parameters() if p.grad is not None]
    assert grads and all(any(lo <= g.data_ptr() < hi for lo, hi in spans) for g in grads)
    rb, b = run(False)
    assert FL._arena is None
    for k in a:
        assert torch.equal(a[k], b[k]), k
    FL.set_grad_arena(None)


def test_deferred_weight_gradient_sums_change_nothing(monkeypatch):
    """fused_linear.deferred_wgrad_reduce (the partial results of all layers of a backward pass summed by ONE launch,
    lsim_wgrad_reduce_batch) against the same two iterations with every layer summing at once (LSIM_DEFER_WGRAD_REDUCE=0): identical weights
    bit for bit; and on one layer directly: both forms of the call, with and without the ELU backward, the same dW / db."""
    import ctypes
    from isaacgymloco_amd import abi, lib
    from isaacgymloco_amd.learn import fused_linear as FL

    def run(defer):
        monkeypatch.setenv("LSIM_DEFER_WGRAD_REDUCE", "1" if defer else "0")
        FL.set_grad_arena(None)
        env, r = _make(seed=7)
        r.enable_graphs()
        r.learn(2, init_at_random_ep_len=False)
        return {k: v.clone() for k, v in r.alg.actor_critic.state_dict().items()}
    a, b = run(True), run(False)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    FL.set_grad_arena(None)
    monkeypatch.setenv("LSIM_DEFER_WGRAD_REDUCE", "1")
    monkeypatch.setattr(FL, "_defer_without_arena", True)         # plain output tensors, no autograd behind these calls
    g = torch.Generator(device="cuda:0").manual_seed(3)
    for k_in, n_out in ((64, 512), (512, 256), (270, 128), (128, 12)):
        x = torch.randn(8192, k_in, device="cuda:0", generator=g)
        go = torch.randn(8192, n_out, device="cuda:0", generator=g)
        dw0, db0 = FL.linear_wgrad(x, go)
        with FL.deferred_wgrad_reduce():
            dw1, db1 = FL.linear_wgrad(x, go, weight_ptr=1234)
            dw2, db2 = FL.linear_wgrad(x, 2.0 * go, weight_ptr=5678)       # a second layer in the same block: its own partial-result buffer
        torch.cuda.synchronize()
        assert torch.equal(dw0, dw1) and torch.equal(db0, db1), (k_in, n_out)
        assert torch.equal(2.0 * dw0, dw2) and torch.equal(2.0 * db0, db2), (k_in, n_out)


def test_module_applied_twice_inside_a_deferred_block_keeps_both_contributions():
    """one HimMLP applied to two inputs inside one deferred block with a gradient arena -- the second contribution to
    a weight cannot take the arena slice again, autograd adds it in place; the first one's pending sum must be in the slice BEFORE that
    (it used to be written afterwards, over the accumulated value).  Against plain torch autograd on an nn.Sequential twin."""
    import copy
    from isaacgymloco_amd.learn import fused_linear as FL
    torch.manual_seed(0)
    net = FL.HimMLP(torch.nn.Linear(64, 512), torch.nn.ELU(), torch.nn.Linear(512, 256), torch.nn.ELU(), FL.SkinnyLinear(256, 12)).to("cuda:0")
    ref = torch.nn.Sequential(*[copy.deepcopy(m) if not isinstance(m, FL.SkinnyLinear) else torch.nn.Linear(256, 12).to("cuda:0") for m in net])
    ref[4].load_state_dict(net[4].state_dict())
    g = torch.Generator(device="cuda:0").manual_seed(5)
    xa, xb = torch.randn(8192, 64, device="cuda:0", generator=g), torch.randn(8192, 64, device="cuda:0", generator=g)
    arena = FL.GradArena()
    FL.set_grad_arena(arena)
    try:
        arena.bucket("all", list(net.parameters()), 0)
        FL.grad_cycle()
        with FL.deferred_wgrad_reduce():
            (net(xa).square().mean() + 3.0 * net(xb).square().mean()).backward()
        arena.bucket("all", list(net.parameters()), 0).adopt()
        (ref(xa).square().mean() + 3.0 * ref(xb).square().mean()).backward()
        torch.cuda.synchronize()
        for (k, pa), pb in zip(net.named_parameters(), ref.parameters()):
            scale = float(pb.grad.abs().max())
            assert float((pa.grad - pb.grad).abs().max()) < 2e-4 * scale, (k, float((pa.grad - pb.grad).abs().max()), scale)
    finally:
        FL.set_grad_arena(None)


def test_ppo_loss_with_the_std_vector_equals_the_broadcast_form():
    """lsim_ppo_loss_std (the policy's std [A] as it is) against lsim_ppo_loss on its broadcast [B, A]: the five statistics, grad_mu and
    grad_value bit for bit (the same arithmetic in the same order), grad_std = the column sums of grad_sigma to summation order"""
    from isaacgymloco_amd.learn.fused_linear import ppo_loss_hip
    g = torch.Generator(device="cuda:0").manual_seed(11)
    B, A = 102400, 12
    r = lambda *s: torch.randn(*s, device="cuda:0", generator=g)
    mu, value = r(B, A).requires_grad_(True), r(B, 1).requires_grad_(True)
    std = (0.5 + torch.rand(A, device="cuda:0", generator=g)).requires_grad_(True)
    actions, old_mu, old_sigma = r(B, A), r(B, A), 0.5 + torch.rand(B, A, device="cuda:0", generator=g)
    old_logp, adv, returns, tv = r(B, 1), r(B, 1), r(B, 1), r(B, 1)
    res = []
    for form in ("std", "broadcast"):
        for t in (mu, value, std):
            t.grad = None
        sigma = std if form == "std" else mu.detach() * 0.0 + std
        loss, st = ppo_loss_hip(mu, sigma, value, actions, old_logp, adv, returns, tv, old_mu, old_sigma, 0.2, 1.0, 0.01, True)
        loss.backward()
        res.append((loss.detach().clone(), st.clone(), mu.grad.clone(), value.grad.clone(), std.grad.clone()))
    (l0, s0, gm0, gv0, gs0), (l1, s1, gm1, gv1, gs1) = res
    assert torch.equal(l0, l1) and torch.equal(s0, s1) and torch.equal(gm0, gm1) and torch.equal(gv0, gv1)
    torch.testing.assert_close(gs0, gs1, rtol=2e-5, atol=1e-7)
    assert float(gs1.abs().max()) > 0


def test_backward_from_the_loss_kernels_gradients_changes_nothing(monkeypatch):
    """fused_linear.backward_losses (torch.autograd.backward(inputs, the gradients the fused loss kernels already produced)) against the ordinary
    loss.backward() (a root gradient of ones times the same gradients): two iterations, identical weights bit for bit"""
    from isaacgymloco_amd.learn import fused_linear as FL

    def run(direct):
        monkeypatch.setenv("LSIM_DIRECT_LOSS_BACKWARD", "1" if direct else "0")
        FL.set_grad_arena(None)
        env, r = _make(seed=9)
        r.enable_graphs()
        r.learn(2, init_at_random_ep_len=False)
        return {k: v.clone() for k, v in r.alg.actor_critic.state_dict().items()}
    a, b = run(True), run(False)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    FL.set_grad_arena(None)


def test_gather_rows_equals_advanced_indexing():
    from isaacgymloco_amd.learn.storage import _gather_rows
    g = torch.Generator(device="cuda:0").manual_seed(1)
    for shape in ((4096, 270), (5000, 238), (4096, 12), (4097, 1), (3000,), (1000, 3, 5), (409600, 270), (8, 238), (9, 270), (10, 64), (12, 384), (50, 386),
                  (40, 271)):       # wide even rows take the wave-per-row kernel (four rows in flight: ragged tails of 1-3 rows included), the rest the block form
        f = torch.randn(*shape, device="cuda:0", generator=g)
        perm = torch.randperm(shape[0] - 7, device="cuda:0")
        assert torch.equal(_gather_rows(f, perm), f[perm]), shape
    b = torch.randint(0, 2, (500, 1), device="cuda:0", dtype=torch.uint8)          # not 4-byte elements: the torch statement
    perm = torch.randperm(500, device="cuda:0")
    assert torch.equal(_gather_rows(b, perm), b[perm])
    # destination rows further apart than their width (lsim_gather_rows_ld): the columns in between are left alone
    for rows, cols, ld in ((4099, 238, 240), (5000, 270, 272), (1000, 45, 48), (77, 12, 16)):
        f = torch.randn(rows, cols, device="cuda:0", generator=g)
        perm = torch.randperm(rows - 3, device="cuda:0")
        buf = torch.full((rows - 3, ld), 7.0, device="cuda:0")
        out = _gather_rows(f, perm, out=buf[:, :cols])
        assert out.data_ptr() == buf.data_ptr() and torch.equal(out, f[perm]) and bool((buf[:, cols:] == 7.0).all()), (rows, cols, ld)


@pytest.mark.parametrize("k_in,n_out,B", [(512, 256, 102400), (256, 128, 102400), (128, 128, 20480 + 12), (512, 256, 4099)])
def test_weight_gradient_on_the_bf16_pipe_is_an_fp32_result(k_in, n_out, B):
    """lsim_wgrad_split_bf16(1): the weight / bias gradient and the ELU backward of the 128-multiple layers with every fp32 operand split
    exactly into three bf16 terms and six products per pair (csrc/ls_learn.h: lsim_k_linear_wgrad_split).  Against fp64 sums its error must
    not exceed the default fp32-pipe kernel's (up to 1.25 x: the two sum in different orders); grad_pre is the same elementwise product, bit
    for bit; ragged batches (rows past the last whole step of 32) and gradient-like operands (wide magnitude range, half the ELU units
    saturated) included."""
    import ctypes
    from isaacgymloco_amd import lib
    L = lib.load()
    gen = torch.Generator(device="cuda:0").manual_seed(k_in + n_out + B)
    x = torch.randn(B, k_in, device="cuda:0", generator=gen)
    g = torch.randn(B, n_out, device="cuda:0", generator=gen) * torch.exp(2.0 * torch.randn(B, 1, device="cuda:0", generator=gen)) * 1e-3
    z = torch.nn.functional.elu(torch.randn(B, n_out, device="cuda:0", generator=gen))
    gp64 = g.double() * torch.where(z > 0, torch.ones_like(z), z + 1.0).double()
    ref_w, ref_b = gp64.t() @ x.double(), gp64.sum(0)
    s = torch.cuda.current_stream().cuda_stream

    def run(on):
        was = L.lsim_wgrad_split_bf16(on)
        try:
            need, parts = ctypes.c_size_t(), ctypes.c_int()
            lib.check(L.lsim_linear_wgrad_workspace(B, k_in, n_out, ctypes.byref(need), ctypes.byref(parts)))
            ws = torch.empty(need.value // 4, device="cuda:0")
            dw, db, gy = torch.empty(n_out, k_in, device="cuda:0"), torch.empty(n_out, device="cuda:0"), torch.empty(B, n_out, device="cuda:0")
            lib.check(L.lsim_linear_elu_wgrad(x.data_ptr(), k_in, g.data_ptr(), n_out, z.data_ptr(), n_out, B, k_in, n_out, dw.data_ptr(), db.data_ptr(),
                                              gy.data_ptr(), ws.data_ptr(), need.value, s), what="lsim_linear_elu_wgrad")
            dw2 = torch.empty_like(dw)
            lib.check(L.lsim_linear_wgrad(x.data_ptr(), k_in, gy.data_ptr(), n_out, B, k_in, n_out, dw2.data_ptr(), None, ws.data_ptr(), need.value, s))
            torch.cuda.synchronize()
            return dw, db, gy, dw2, parts.value
        finally:
            L.lsim_wgrad_split_bf16(was)
    dw0, db0, gy0, dwp0, parts0 = run(0)
    dw1, db1, gy1, dwp1, parts1 = run(1)
    assert torch.equal(gy0, gy1) and torch.equal(gy1, gp64.float())
    mag = gp64.abs().t() @ x.double().abs()                                   # sum |a||b| per output: what a summation error scales with
    e0, e1 = float(((dw0.double() - ref_w).abs() / mag).max()), float(((dw1.double() - ref_w).abs() / mag).max())
    p0, p1 = float(((dwp0.double() - ref_w).abs() / mag).max()), float(((dwp1.double() - ref_w).abs() / mag).max())
    assert e1 <= 1.25 * e0 + 1e-9 and p1 <= 1.25 * p0 + 1e-9, (e0, e1, p0, p1)
    assert e1 < 2e-6 and p1 < 2e-6, (e1, p1)
    bmag = gp64.abs().sum(0)
    assert float(((db1.double() - ref_b).abs() / bmag).max()) < 2e-6
    assert not torch.equal(dw0, dw1) or parts0 != parts1                       # the switch did select another kernel


def test_padded_shuffle_rows_and_padded_first_layers_change_the_update_only_by_rounding(monkeypatch):
    """Round 6: the shuffled 270- / 238-wide observation fields live in rows 272 / 240 floats apart (zero padding) and the networks' first layers run
    the library's fused Linear + ELU forward against zero-padded weight copies (learn/storage.py: _shuffle, learn/fused_linear.py: linear_elu_forward).
    Against the same two iterations with contiguous rows (LSIM_PAD_SHUFFLED=0: those layers on BLAS + torch ELU): the minibatch views hold the same
    values, the fused first layer equals F.elu(F.linear) to fp32 rounding, and the trained weights agree far inside one optimiser step."""
    import torch.nn.functional as F
    from isaacgymloco_amd.learn import fused_linear as FL

    def run(pad):
        monkeypatch.setenv("LSIM_PAD_SHUFFLED", "1" if pad else "0")
        FL.set_grad_arena(None)
        env, r = _make(seed=9)
        r.enable_graphs()
        r.learn(2, init_at_random_ep_len=False)
        st = r.alg.storage
        gen = st.mini_batch_generator(4, 1)
        mb = next(gen)
        gen.close()
        return r, {k: v.clone() for k, v in r.alg.actor_critic.state_dict().items()}, mb
    ra, a, mba = run(True)
    obs, crit = mba[0], mba[1]
    assert obs.shape[1] == 270 and obs.stride(0) == 272 and crit.shape[1] == 238 and crit.stride(0) == 240 and obs.data_ptr() % 16 == 0
    ac = ra.alg.actor_critic
    lin = ac.critic[0]
    with torch.no_grad():
        z = FL.linear_elu_forward(crit, lin.weight, lin.bias)
        ref = F.elu(F.linear(crit.contiguous().double(), lin.weight.double(), lin.bias.double())).float()
    assert FL._padded_weights, "the padded-weight form of the fused forward did not run"
    torch.testing.assert_close(z, ref, rtol=1e-5, atol=2e-5)
    rb, b, mbb = run(False)
    assert mbb[0].is_contiguous() and mbb[1].is_contiguous()
    for k in a:        # 2 iterations x 20 Adam steps at lr 1e-3 from identical states and identical rollouts of iteration 1
        torch.testing.assert_close(a[k], b[k], rtol=0, atol=2e-3, msg=lambda m, k=k: f"{k}: {m}")
    FL.set_grad_arena(None)
