"""E21: the init-time domain randomisation (motor strength, Kp / Kd factors, payload, COM shift, friction buckets, initial terrain level /
type / origin) against the reference's own `_init_buffers` / `_process_rigid_shape_props` / `_get_env_origins` (LR:999-1028, LR:506-513,
LR:1221-1240) run with the matching Philox uniforms injected (tools/gen_golden_init.py)."""
import os

import numpy as np
import pytest

from helpers import C, ROOT, make_oracle

KEYS = ["motor_strength", "kp_factors", "kd_factors", "motor_strength_factors", "payload", "com_displacement", "friction", "env_origins"]
EXACT = ["terrain_levels", "terrain_types"]


def _fx(task):
    return np.load(os.path.join(ROOT, "tests", "golden", f"init_{task}.npz"))


@pytest.mark.parametrize("task", ["aliengo", "aliengo_stairs"])
def test_oracle_init_draws_match_reference(task):
    fx = _fx(task)
    orc, lc, model, ter = make_oracle(C.TASKS[task][0](), int(fx["num_envs"]), seed=int(fx["seed"]))
    for k in KEYS:
        np.testing.assert_allclose(orc.buf[k].reshape(fx[k].shape), fx[k], rtol=1e-6, atol=1e-6, err_msg=k)
    for k in EXACT:
        np.testing.assert_array_equal(orc.buf[k], fx[k], err_msg=k)
    # the ranges of the configuration (AGC:151-166): a sanity check on the fixture itself
    assert 0.0 <= fx["payload"].min() and fx["payload"].max() <= 3.0 and np.abs(fx["com_displacement"]).max() <= 0.05
    assert 0.2 <= fx["friction"].min() and fx["friction"].max() <= 1.25 and len(np.unique(fx["friction"])) <= 64


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["aliengo", "aliengo_stairs"])
def test_hip_init_draws_match_reference(task):
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    fx = _fx(task)
    cfg = C.TASKS[task][0]()
    cfg.env.num_envs = int(fx["num_envs"])
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=int(fx["seed"]))
    for k in KEYS:
        np.testing.assert_allclose(env.buf[k].cpu().numpy().reshape(fx[k].shape), fx[k], rtol=1e-6, atol=1e-6, err_msg=k)
    for k in EXACT:
        np.testing.assert_array_equal(env.buf[k].cpu().numpy(), fx[k], err_msg=k)
