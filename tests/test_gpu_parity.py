"""Parity tests proper: the HIP library, called through the C-ABI, against (1) the golden vectors captured from the
reference's LeggedRobot.step() and (2) the CPU oracle for the dynamics.  Need a real MI355X (-m gpu)."""
import numpy as np
import pytest

import golden_replay as GR
from helpers import C, T, make_oracle, quiet_cfg, abi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", GR.SCENARIOS)
def test_hip_matches_reference_step(name):
    from hip_backend import HipBackend
    fx = GR.load(name)
    cfg = GR.scenario_cfg(name)
    be = HipBackend(cfg, int(fx["num_envs"]), GR.FixtureTerrain(fx), seed=int(fx["seed"]))
    n = 0
    for t, ref in GR.replay(fx, be, be.get, be.put):
        GR.compare_step(t, ref, be.get, be.stats_row)
        n += 1
    assert n == fx["in_actions"].shape[0]


@pytest.mark.parametrize("quiet", [True, False])
def test_hip_physics_matches_oracle(quiet):
    """Dynamics: structured fp32 wave solver (HIP) vs dense fp64 oracle on the same seeds/actions.  fp32 tolerance:
    states agree to 2e-3 abs over 12 steps (48 sub-steps) of contact-rich motion; contact forces to 0.5 N (~0.2 %)."""
    from hip_backend import HipBackend
    N = 16
    cfg = quiet_cfg("aliengo") if quiet else C.TASKS["aliengo"][0]()
    cfg.terrain.terrain_proportions = [1.0, 0.0, 0.0, 0.0]
    orc, lc, model, ter = make_oracle(cfg, N, seed=5)
    be = HipBackend(cfg, N, ter, seed=5)
    orc.reset_all(); be.reset_all()
    rs = np.random.RandomState(0)
    for t in range(12):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        orc.step(a); be.step(a)
        np.testing.assert_array_equal(be.get("reset"), orc.buf["reset"], err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("root_states"), orc.buf["root_states"], atol=2e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("dof_state"), orc.buf["dof_state"], atol=5e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("contact_forces"), orc.buf["contact_forces"], atol=0.5, rtol=5e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("rew"), orc.buf["rew"], atol=1e-3, rtol=1e-3, err_msg=f"step {t}")
        np.testing.assert_allclose(be.get("obs"), orc.buf["obs"], atol=5e-3, rtol=1e-3, err_msg=f"step {t}")


def test_hip_full_size_invariants():
    """BASELINE size (N=4096): size-independent properties -- finite state, unit quaternions, torque and joint-velocity
    limits respected, standing robots carry their weight, reset bookkeeping consistent."""
    import torch
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    cfg = C.aliengo_cfg()
    cfg.env.num_envs = 4096
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(60):
        a = torch.randn(4096, 12, device="cuda", generator=g) * 0.3
        env.step_device(a)
    torch.cuda.synchronize()
    root = env.root_states.cpu().numpy()
    assert np.isfinite(root).all() and np.isfinite(env.obs_buf.cpu().numpy()).all() and np.isfinite(env.rew_buf.cpu().numpy()).all()
    np.testing.assert_allclose(np.linalg.norm(root[:, 3:7], axis=1), 1.0, atol=1e-4)
    tau = env.torques.cpu().numpy()
    assert (np.abs(tau) <= np.array([44, 44, 55] * 4) + 1e-4).all()
    qd = env.dof_vel.cpu().numpy()
    assert (np.abs(qd) <= np.array([20, 20, 15.89] * 4) + 1e-3).all()
    ep = env.episode_length_buf.cpu().numpy()
    assert ep.min() >= 0 and ep.max() <= 61
    assert np.abs(env.obs_buf.cpu().numpy()).max() <= 100.0
    # weight support: mean vertical contact force of the robots that are standing ~ m g (24.94 kg + payload 0..3)
    fz = env.contact_forces[:, :, 2].sum(1).cpu().numpy()
    standing = (root[:, 2] - env.env_origins[:, 2].cpu().numpy() > 0.25) & (fz > 50)
    assert standing.sum() > 500
    assert 180.0 < np.median(fz[standing]) < 340.0
