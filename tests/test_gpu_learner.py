"""GPU: the hipGraph-captured rollout step writes the same storage as the eager runner step (same ops); full PPO iteration runs."""
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
    assert not run.enable_graphs()        # graph capture is only wired for the plain HIM runner
    before = {k: v.clone() for k, v in run.alg.discriminator.state_dict().items()}
    run.learn(2, init_at_random_ep_len=True)
    after = run.alg.discriminator.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before)
    assert all(torch.isfinite(v).all() for v in after.values())
    assert all(torch.isfinite(v).all() for v in run.alg.actor_critic.state_dict().values())
