"""Replay a tests/golden/step_*.npz fixture through a simulator backend (oracle or HIP) -- test infrastructure.

The fixture holds, per step, the injected simulator state + actions (inputs) and everything the
reference's LeggedRobot.step() produced on them (outputs).  `replay()` drives a backend that exposes
`buf[name]` arrays, `.step(actions, flags)`, `.reset_all()`, `.step_counter`, and yields
(step_index, outputs_dict_from_fixture) after each step so the caller can compare.
"""
import os

import numpy as np

from helpers import C, abi

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

def _flat(cfg):
    cfg.terrain.terrain_proportions = [1.0, 0.0, 0.0, 0.0]


def _all_terms(cfg):
    _flat(cfg)
    for k, nm in enumerate(abi.REWARD_NAMES):
        setattr(cfg.rewards.scales, nm, (0.5 + 0.01 * k) * (-1.0 if k % 3 else 1.0))
    cfg.rewards.only_positive_rewards = True


TWEAKS = {
    "aliengo_flat": ("aliengo", _flat),
    "aliengo_stairs": ("aliengo_stairs", lambda cfg: setattr(cfg.terrain, "terrain_proportions", [0.0, 0.0, 0.0, 0.0, 0.5, 0.5, 0.0, 0.0, 0.0, 0.0])),
    "aliengo_allterms": ("aliengo", _all_terms),
    "aliengo_amp": ("aliengo_amp", _flat),
}
SCENARIOS = sorted(TWEAKS)


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, f"step_{name}.npz"))


def scenario_cfg(name):
    task, tweak = TWEAKS[name]
    cfg = C.TASKS[task][0]()
    tweak(cfg)
    return cfg


class FixtureTerrain:
    """Terrain stand-in carrying the fixture's own grid (robust to later generator changes)."""

    def __init__(self, fx):
        self.heightsamples = fx["height_grid"]
        self.env_origins = fx["terrain_origins"]
        self.tot_rows, self.tot_cols = self.heightsamples.shape


# (fixture key, buffer name, atol) -- fp32 tolerance: 2e-5 absolute on O(1) quantities (the restated torch ops
# differ from torch only by summation order / fused multiply-add); exact for integer and boolean outputs
FLOAT_CHECKS = [
    ("torques", "torques", 2e-4), ("commands", "commands", 1e-5), ("base_lin_vel", "base_lin_vel", 1e-5),
    ("base_ang_vel", "base_ang_vel", 1e-5), ("projected_gravity", "projected_gravity", 1e-6),
    ("measured_heights", "measured_heights", 1e-6), ("rew", "rew", 2e-5), ("obs", "obs", 2e-5), ("priv_obs", "priv_obs", 2e-5),
    ("amp_obs", "amp_obs", 1e-5), ("feet_air_time", "feet_air_time", 1e-6), ("root_states", "root_states", 1e-5),
    ("dof_state", "dof_state", 1e-5), ("env_origins", "env_origins", 1e-6), ("kp_factors", "kp_factors", 1e-6),
    ("kd_factors", "kd_factors", 1e-6), ("friction", "friction", 1e-6), ("last_actions", "last_actions", 0.0),
    ("last_last_actions", "last_last_actions", 0.0), ("last_dof_vel", "last_dof_vel", 1e-6),
]
EXACT_CHECKS = [("reset", "reset"), ("time_out", "time_out"), ("extras_time_outs", "extras_time_outs"),
                ("last_contacts", "last_contacts"), ("contact_filt", "contact_filt"), ("terrain_levels", "terrain_levels"),
                ("episode_length", "episode_length")]


def replay(fx, backend, get, put):
    """get(name) -> numpy copy of a backend buffer; put(name, array) writes one."""
    N = int(fx["num_envs"])
    feet = [4, 8, 12, 16]
    backend.reset_all()
    T = fx["in_actions"].shape[0]
    for t in range(T):
        backend.step_counter = int(fx["in_counter_before"][t])
        put("episode_length", fx["in_ep_before"][t])
        put("root_states", fx["in_root"][t])
        put("dof_state", fx["in_dof"][t])
        body = np.zeros((N, 17, 13), np.float32)
        body[:, feet, :] = fx["in_body_feet"][t]
        put("rigid_body_states", body)
        put("contact_forces", fx["in_contact"][t])
        if not np.isnan(fx["in_track_override"][t]):
            es = get("episode_sums")
            es[:, abi.REWARD_IDS["tracking_lin_vel"]] = fx["in_track_override"][t]
            put("episode_sums", es)
        backend.step(fx["in_actions"][t], flags=abi.STEP_SKIP_PHYSICS)
        yield t, {k[4:]: fx[k][t] for k in fx.files if k.startswith("out_")}


def compare_step(t, ref, get, stats_row, dt=0.02):
    """Assert one replayed step against the reference outputs; returns max abs errors for reporting."""
    errs = {}
    for key, name in EXACT_CHECKS:
        got = get(name)
        np.testing.assert_array_equal(got.astype(np.int64), ref[key].astype(np.int64), err_msg=f"step {t}: {key}")
    for key, name, atol in FLOAT_CHECKS:
        got = get(name)
        np.testing.assert_allclose(got, ref[key], rtol=2e-5, atol=atol, err_msg=f"step {t}: {key}")
        errs[key] = float(np.max(np.abs(got - ref[key]))) if got.size else 0.0
    mask = ref["term_mask"].astype(bool)
    np.testing.assert_array_equal(get("reset").astype(bool), mask, err_msg=f"step {t}: termination ids")
    if mask.any():
        np.testing.assert_allclose(get("term_priv_obs")[mask], ref["term_priv_obs"][mask], rtol=2e-5, atol=2e-5, err_msg=f"step {t}: term_priv_obs")
        if np.abs(ref["term_amp"]).max() > 0:   # captured only when the reference ran with USING_AMP (LR:173)
            np.testing.assert_allclose(get("term_amp_obs")[mask], ref["term_amp"][mask], rtol=2e-5, atol=1e-5, err_msg=f"step {t}: terminal AMP states")
    es = get("episode_sums")
    np.testing.assert_allclose(es, ref["episode_sums"], rtol=2e-5, atol=2e-5, err_msg=f"step {t}: episode_sums")
    st = get("stats")[stats_row]
    S = abi.STATS
    np.testing.assert_allclose(st[S["cmd_ranges"]:S["cmd_ranges"] + 8].reshape(4, 2), ref["command_ranges"], rtol=1e-6, atol=1e-6, err_msg=f"step {t}: command_ranges")
    if mask.any():   # extras["episode"] (LR:346-353): mean over reset envs of sum / len / dt
        n = st[S["reset_count"]]
        assert int(n) == int(mask.sum())
        mine = st[S["episode_sums"]:S["episode_sums"] + abi.NUM_REWARD_TERMS] / n / dt
        valid = ~np.isnan(ref["ep_stats"])
        np.testing.assert_allclose(mine[valid], ref["ep_stats"][valid], rtol=1e-4, atol=1e-5, err_msg=f"step {t}: extras[episode]")
        if not np.isnan(ref["level_mean"]):
            np.testing.assert_allclose(get("terrain_levels").astype(np.float32).mean(), ref["level_mean"], rtol=1e-6)
    return errs
