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
    "aliengo_reset_subset": ("aliengo", _flat),      # + the reference's reset_idx(env_ids) called by hand after the last step (fin_* keys)
}
SCENARIOS = sorted(TWEAKS)


def _all_terms_10(cfg):
    _all_terms(cfg)
    cfg.terrain.terrain_proportions = [0.5, 0.0, 0.0, 0.0, 0.2, 0.1, 0.0, 0.0, 0.1, 0.1]


# BASELINE-size fixtures (N = 4096, tools/gen_golden.py: run_scenario_big): file step4096_<name>.npz
BIG_TWEAKS = {
    "aliengo": ("aliengo", _all_terms_10),
    "aliengo_stairs": TWEAKS["aliengo_stairs"],
    "aliengo_amp": ("aliengo_amp", _flat),
}
BIG_SCENARIOS = sorted(BIG_TWEAKS)


def load_big(name):
    return np.load(os.path.join(GOLDEN_DIR, f"step4096_{name}.npz"))


def big_scenario_cfg(name):
    task, tweak = BIG_TWEAKS[name]
    cfg = C.TASKS[task][0]()
    tweak(cfg)
    return cfg


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


def replay(fx, backend, get, put, extra_flags=0):
    """get(name) -> numpy copy of a backend buffer; put(name, array) writes one.  extra_flags: or-ed into the step flags (LSIM_STEP_TWO_KERNELS)."""
    N = int(fx["num_envs"])
    feet = [4, 8, 12, 16]
    big = "big" in fx.files
    if big:   # the init-time draws that place the robots (LR:1221-1244) at BASELINE size, before anything has stepped
        for k in ("terrain_levels", "terrain_types", "env_origins"):
            np.testing.assert_array_equal(get(k), fx["init_" + k], err_msg=f"init {k} at N = {N}")
    backend.reset_all()
    T = len(fx["in_counter_before"])
    for t in range(T):
        backend.step_counter = int(fx["in_counter_before"][t])
        if big:   # injected state rebuilt from tests/big_inputs.py (the generator fed the same arrays to the reference; CRC in the fixture)
            import big_inputs
            inp = big_inputs.synth_step_inputs(N, t, get("env_origins"), int(fx["seed"]))
            assert big_inputs.crc_of_inputs(inp) == fx["in_crc"][t], f"step {t}: synthetic inputs differ from the ones the fixture was generated with"
            if int(fx["in_counter_before"][t]) + 1 == 1000:
                inp["ep_before"][3] = 1000
        else:
            inp = {k: fx["in_" + k][t] for k in ("actions", "root", "dof", "body_feet", "contact", "ep_before")}
        if big and t > 0:
            put("terrain_levels", inp["terrain_levels"])
        put("episode_length", inp["ep_before"])
        put("root_states", inp["root"])
        put("dof_state", inp["dof"])
        body = np.zeros((N, 17, 13), np.float32)
        body[:, feet, :] = inp["body_feet"]
        put("rigid_body_states", body)
        put("contact_forces", inp["contact"])
        if not np.isnan(fx["in_track_override"][t]):
            es = get("episode_sums")
            es[:, abi.REWARD_IDS["tracking_lin_vel"]] = fx["in_track_override"][t]
            put("episode_sums", es)
        last_before = get("last_actions")
        backend.step(inp["actions"], flags=abi.STEP_SKIP_PHYSICS | abi.STEP_RECORD_SUBSTEPS | extra_flags)
        out = {k[4:]: fx[k][t] for k in fx.files if k.startswith("out_")}
        out["last_actions_before"] = last_before
        if big:
            out["sel"] = fx["sel"]
            out["red"] = {k[4:]: fx[k][t] for k in fx.files if k.startswith("red_")}
        yield t, out


def compare_step(t, ref, get_full, stats_row, dt=0.02):
    """Assert one replayed step against the reference outputs; returns max abs errors for reporting.
    BASELINE-size fixtures store the fat outputs for the rows ref["sel"] only, plus fp64 column sums over ALL envs (ref["red"])."""
    errs = {}
    sel, red = ref.get("sel"), ref.get("red", {})
    N = get_full("reset").shape[0]

    def get(name, key=None):
        """backend buffer, cut to the rows the fixture holds for `key`"""
        got = get_full(name)
        if sel is not None and key is not None and ref[key].shape[0] != N:
            if "sum_" + key in red:   # every env contributes: |sum error| <= rtol * sum|x| + N * atol
                atol = dict((k, a) for k, _, a in FLOAT_CHECKS).get(key, 2e-5)
                s = got.astype(np.float64).sum(0)
                bound = 2e-5 * red["abs_" + key] + N * max(atol, 1e-7)
                assert np.all(np.abs(s - red["sum_" + key]) <= bound), f"step {t}: column sums of {key} over all {N} envs"
            got = got[sel]
        return got
    for key, name in EXACT_CHECKS:
        got = get(name, key)
        np.testing.assert_array_equal(got.astype(np.int64), ref[key].astype(np.int64), err_msg=f"step {t}: {key}")
    for key, name, atol in FLOAT_CHECKS:
        got = get(name, key)
        # `rew` over a whole BASELINE-size batch with all 51 terms on: terms of +-1e3 (dof_acc, torques, ...) cancel to O(10), so the worst of
        # 4096 envs shows the fp32 summation-order / contraction difference amplified ~30x (measured 2.8e-5 relative on MI355X)
        rtol = 1e-4 if (key == "rew" and sel is not None) else 2e-5
        np.testing.assert_allclose(got, ref[key], rtol=rtol, atol=atol, err_msg=f"step {t}: {key}")
        errs[key] = float(np.max(np.abs(got - ref[key]))) if got.size else 0.0
    # E2, the action-delay model (LR:133-138) on EVERY sub-step: the drawn delay, the torques _compute_torques returned for each of the four
    # delayed actions (only the last sub-step's action equals `actions` whatever the delay), and the delayed actions rebuilt from the
    # backend's own buffers
    delay = get("delay_steps")
    np.testing.assert_array_equal(delay, ref["delay_steps"], err_msg=f"step {t}: delay_steps")
    np.testing.assert_allclose(get("substep_torques", "substep_torques"), ref["substep_torques"], rtol=2e-5, atol=2e-4, err_msg=f"step {t}: per-sub-step torques")
    act, last = get("actions"), ref["last_actions_before"]
    if sel is not None:
        act, last, delay = act[sel], last[sel], delay[sel]
    sub = np.arange(ref["delayed_actions"].shape[1])
    rebuilt = last[:, None, :] + (act - last)[:, None, :] * (sub[None, :, None] >= delay[:, None, None]).astype(np.float32)
    np.testing.assert_allclose(rebuilt, ref["delayed_actions"], rtol=0, atol=1e-6, err_msg=f"step {t}: delayed_actions")
    mask = ref["term_mask"].astype(bool)
    np.testing.assert_array_equal(get("reset").astype(bool), mask, err_msg=f"step {t}: termination ids")
    if mask.any():
        msel = mask if sel is None else mask[sel]
        gtp, gta = get_full("term_priv_obs"), get_full("term_amp_obs")
        if sel is not None:
            gtp, gta = gtp[sel], gta[sel]
        np.testing.assert_allclose(gtp[msel], ref["term_priv_obs"][msel], rtol=2e-5, atol=2e-5, err_msg=f"step {t}: term_priv_obs")
        if np.abs(ref["term_amp"]).max() > 0:   # captured only when the reference ran with USING_AMP (LR:173)
            np.testing.assert_allclose(gta[msel], ref["term_amp"][msel], rtol=2e-5, atol=1e-5, err_msg=f"step {t}: terminal AMP states")
    es = get("episode_sums", "episode_sums")
    np.testing.assert_allclose(es, ref["episode_sums"], rtol=2e-5, atol=2e-5, err_msg=f"step {t}: episode_sums")
    if "stumble_sums" in ref:   # BASELINE-size fixtures: the two terms with index slices (LR:1597-1607), every env
        ids = [abi.REWARD_IDS["feet_stumble"], abi.REWARD_IDS["feet_stumble_up"]]
        np.testing.assert_allclose(get_full("episode_sums")[:, ids], ref["stumble_sums"], rtol=2e-5, atol=1e-6, err_msg=f"step {t}: stumble slices")
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


def replay_final_reset(fx, backend, get, put):
    """after replay(): the by-hand reset_idx(env_ids) of the fixture (LR:290 called outside step()) through backend.reset_envs(mask)"""
    N = int(fx["num_envs"])
    assert backend.step_counter == int(fx["fin_counter"])
    es = get("episode_sums")
    es[:, abi.REWARD_IDS["tracking_lin_vel"]] = fx["fin_track"]
    put("episode_sums", es)
    mask = np.zeros(N, np.uint8)
    mask[fx["fin_ids"]] = 1
    backend.reset_envs(mask)
    return mask


def compare_final_reset(fx, mask, get, stats_row, dt=0.02):
    """what the reference's by-hand reset_idx(env_ids) left behind, same tolerances as compare_step"""
    atols = dict((k, a) for k, _, a in FLOAT_CHECKS)
    for key in ("reset", "time_out", "extras_time_outs", "terrain_levels", "episode_length"):
        np.testing.assert_array_equal(get(key).astype(np.int64), fx["fin_" + key].astype(np.int64), err_msg=f"by-hand reset: {key}")
    for key in ("commands", "root_states", "dof_state", "env_origins", "kp_factors", "kd_factors", "friction", "last_actions", "last_last_actions",
                "last_dof_vel", "feet_air_time", "measured_heights"):
        np.testing.assert_allclose(get(key), fx["fin_" + key], rtol=2e-5, atol=atols[key], err_msg=f"by-hand reset: {key}")
    np.testing.assert_allclose(get("episode_sums"), fx["fin_episode_sums"], rtol=2e-5, atol=2e-5, err_msg="by-hand reset: episode_sums")
    st = get("stats")[stats_row]
    S = abi.STATS
    np.testing.assert_allclose(st[S["cmd_ranges"]:S["cmd_ranges"] + 8].reshape(4, 2), fx["fin_command_ranges"], rtol=1e-6, atol=1e-6, err_msg="by-hand reset: command_ranges")
    assert int(st[S["reset_count"]]) == int(mask.sum())
    mine = st[S["episode_sums"]:S["episode_sums"] + abi.NUM_REWARD_TERMS] / st[S["reset_count"]] / dt
    valid = ~np.isnan(fx["fin_ep_stats"])
    assert valid.any()
    np.testing.assert_allclose(mine[valid], fx["fin_ep_stats"][valid], rtol=1e-4, atol=1e-5, err_msg="by-hand reset: extras[episode]")
    if not np.isnan(fx["fin_level_mean"]):
        np.testing.assert_allclose(get("terrain_levels").astype(np.float32).mean(), fx["fin_level_mean"], rtol=1e-6)
    # the fixture is not vacuous: the chosen envs moved, the others did not, and the command curriculum fired on the SET's mean
    moved = np.abs(fx["fin_root_states"] - fx["finb_root_states"]).max(1) > 0
    np.testing.assert_array_equal(moved, mask.astype(bool))
    assert np.all(fx["fin_episode_length"][mask == 1] == 0) and np.all(fx["fin_episode_length"][mask == 0] == fx["finb_episode_length"][mask == 0])
    assert fx["fin_command_ranges"][0, 1] > fx["fin_ranges_before"][0, 1] or fx["fin_command_ranges"][0, 0] < fx["fin_ranges_before"][0, 0]
