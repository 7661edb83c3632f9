"""LeggedRobot.reset_idx(env_ids) called from outside a step (LR:290-361) on a subset of the robots: lsim_reset_envs against the oracle's
reset_idx -- the same function the reference's step() calls on the terminated envs (LR:229), pinned there by the golden fixtures, and pinned
for the by-hand call itself by tests/golden/step_aliengo_reset_subset.npz (the reference's own reset_idx(env_ids) after the last replayed step;
test_oracle_golden / test_emu_golden / test_gpu_parity).  Here: more cases (ragged subset, empty set, mask of ones, curriculum over the set).
CPU leg: the lane-emulated kernel sources; GPU leg: the HIP library through the C-ABI and through LeggedRobot.reset_idx."""
import numpy as np
import pytest

from helpers import C, make_oracle, abi

SYNC = ("root_states", "dof_state", "commands", "last_actions", "last_last_actions", "last_dof_pos", "last_dof_vel", "last_torques", "last_root_vel",
        "episode_length", "terrain_levels", "env_origins", "kp_factors", "kd_factors", "motor_strength_factors", "friction", "restitution",
        "feet_air_time", "last_contacts", "episode_sums", "obs", "time_out", "reset", "measured_heights", "extras_time_outs")
# what reset_idx writes; draws are pure functions of (seed, env, step, tag) and every value is a few fp32 operations: exact for the
# lane-emulated sources (compiled without contraction, like the oracle), within a few fp32 ulps for the HIP build (lo + u * (hi - lo) is one FMA there)
EXACT = ("root_states", "dof_state", "commands", "last_actions", "last_last_actions", "last_dof_pos", "last_dof_vel", "last_torques",
         "episode_length", "terrain_levels", "env_origins", "kp_factors", "kd_factors", "motor_strength_factors", "friction", "restitution",
         "feet_air_time", "episode_sums", "reset", "extras_time_outs", "obs")
S = abi.STATS


def _cfg():
    cfg = C.TASKS["aliengo"][0]()
    cfg.commands.curriculum = True
    cfg.domain_rand.randomize_restitution = True
    return cfg


def _walk(orc, be, get, put, steps, seed):
    rs = np.random.RandomState(seed)
    N = orc.cfg.num_envs
    for t in range(steps):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        orc.step(a); be.step(a)
    for k in SYNC:
        put(k, orc.buf[k])


ULP = dict(rtol=2e-6, atol=2e-7)      # a few fp32 ulps: the yaw command goes through atan2 (LR:461-463)


def _same(a, b, exact, msg):
    if exact or a.dtype.kind != "f":
        np.testing.assert_array_equal(a, b, err_msg=msg)
    else:
        np.testing.assert_allclose(a, b, err_msg=msg, **ULP)


def _compare(orc, get, stats_row, mask, before, tag, exact=True):
    for k in EXACT:
        _same(get(k), orc.buf[k], exact, f"{tag}: {k}")
    np.testing.assert_allclose(get("measured_heights"), orc.buf["measured_heights"], atol=1e-6, err_msg=tag)
    mine, ref = get("stats")[stats_row()], orc.buf["stats"][orc.stats_row]
    assert mine[S["reset_count"]] == ref[S["reset_count"]] == mask.sum(), tag
    np.testing.assert_array_equal(mine[S["cmd_ranges"]:S["cmd_ranges"] + 8], ref[S["cmd_ranges"]:S["cmd_ranges"] + 8], err_msg=tag)
    n = abi.NUM_REWARD_TERMS
    np.testing.assert_allclose(mine[S["episode_sums"]:S["episode_sums"] + n], ref[S["episode_sums"]:S["episode_sums"] + n], rtol=1e-5, atol=1e-6, err_msg=tag)
    keep = mask == 0
    for k in ("root_states", "dof_state", "commands", "episode_length", "episode_sums", "feet_air_time", "kp_factors", "terrain_levels", "last_actions"):
        np.testing.assert_array_equal(get(k)[keep], before[k][keep], err_msg=f"{tag}: {k} of an env that was not reset changed")


def _check(orc, be, get, put, stats_row, set_counter, exact=True):
    N = orc.cfg.num_envs
    c = orc.cfg

    def snapshot():
        """both sides start each case from the oracle's state, to the bit"""
        for k in SYNC:
            put(k, orc.buf[k])
        return {k: orc.buf[k].copy() for k in SYNC}
    orc.reset_all(); be.reset_all()
    _walk(orc, be, get, put, 5, seed=0)
    # 1: a ragged subset, first and last env included
    mask = np.zeros(N, np.uint8); mask[[0, 3, 4, 11, N - 1]] = 1
    before = snapshot()
    orc.reset_envs(mask); be.reset_envs(mask)
    _compare(orc, get, stats_row, mask, before, "subset", exact)
    assert np.all(orc.buf["episode_length"][mask == 1] == 0) and np.all(orc.buf["episode_length"][mask == 0] == 5)
    assert np.all(orc.buf["reset"][mask == 1] == 1)
    moved = np.abs(orc.buf["root_states"] - before["root_states"]).max(1) > 0
    np.testing.assert_array_equal(moved, mask == 1)
    # 2: the simulation goes on from the partially reset state (same tolerance as the dynamics parity tests)
    rs = np.random.RandomState(5)
    for t in range(3):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        orc.step(a); be.step(a)
        np.testing.assert_array_equal(get("reset"), orc.buf["reset"], err_msg=f"step {t} after the subset reset")
        np.testing.assert_allclose(get("root_states"), orc.buf["root_states"], atol=2e-3, rtol=1e-3)
        np.testing.assert_allclose(get("obs"), orc.buf["obs"], atol=5e-3, rtol=1e-3)
    for k in SYNC:
        put(k, orc.buf[k])
    # 3: command curriculum over the reset SET (LR:307-308, LR:868-880): on a multiple of max_episode_length the mean tracking reward of the
    #    two chosen envs is above the bar while the mean over all envs is not -- the ranges must widen
    orc.step_counter = c.max_episode_length; set_counter(c.max_episode_length)
    es = orc.buf["episode_sums"].copy()
    es[:, abi.REWARD_IDS["tracking_lin_vel"]] = 0.0
    bar = 0.8 * c.reward_scales[abi.REWARD_IDS["tracking_lin_vel"]] * c.max_episode_length
    es[[2, 7], abi.REWARD_IDS["tracking_lin_vel"]] = 1.25 * bar
    orc.buf["episode_sums"][...] = es; put("episode_sums", es)
    mask = np.zeros(N, np.uint8); mask[[2, 7]] = 1
    before = snapshot()
    r0 = orc.buf["stats"][orc.stats_row][S["cmd_ranges"]:S["cmd_ranges"] + 8].copy()
    orc.reset_envs(mask); be.reset_envs(mask)
    _compare(orc, get, stats_row, mask, before, "curriculum", exact)
    r1 = get("stats")[stats_row()][S["cmd_ranges"]:S["cmd_ranges"] + 8]
    assert r1[1] == pytest.approx(min(r0[1] + 0.1, c.max_forward_curriculum)) and r1[1] > r0[1], (r0, r1)
    # 4: an empty id set is the reference's early return (LR:298)
    before = snapshot()
    mask = np.zeros(N, np.uint8)
    orc.reset_envs(mask); be.reset_envs(mask)
    _compare(orc, get, stats_row, mask, before, "empty", exact)
    for k in SYNC:
        np.testing.assert_array_equal(get(k), before[k], err_msg=f"empty: {k}")
    # 4b: every by-hand call draws fresh values (include/lsim.h: the draws are salted with the handle's call count): the same robots
    #     reset twice between two steps get two different states
    mask = np.zeros(N, np.uint8); mask[[1, 2, 7]] = 1
    before = snapshot()
    orc.reset_envs(mask); be.reset_envs(mask)
    _compare(orc, get, stats_row, mask, before, "by hand, first", exact)
    first = {k: orc.buf[k][mask == 1].copy() for k in ("dof_state", "root_states", "commands", "kp_factors")}
    before = snapshot()
    orc.reset_envs(mask); be.reset_envs(mask)
    _compare(orc, get, stats_row, mask, before, "by hand, again", exact)
    for k, v in first.items():
        assert not np.array_equal(orc.buf[k][mask == 1], v), f"{k}: the second by-hand reset repeated the first"
    # 5: every env through the mask form equals reset_idx(all) in everything that is not drawn (the by-hand call salts its draws)
    orc.step_counter = c.max_episode_length + 1; set_counter(c.max_episode_length + 1)
    mask = np.ones(N, np.uint8)
    before = snapshot()
    orc.reset_envs(mask); be.reset_envs(mask)
    _compare(orc, get, stats_row, mask, before, "all through the mask", exact)
    snap = {k: get(k).copy() for k in EXACT}
    for k in SYNC:
        orc.buf[k][...] = before[k]; put(k, before[k])
    orc.reset_all(); be.reset_all()
    for k in ("last_actions", "last_last_actions", "last_dof_pos", "last_dof_vel", "last_torques", "episode_length", "feet_air_time", "episode_sums",
              "reset", "extras_time_outs", "obs"):
        np.testing.assert_array_equal(get(k), snap[k], err_msg=f"reset_all vs mask of ones: {k}")
    assert not np.array_equal(get("dof_state"), snap["dof_state"])


def test_emu_reset_envs_matches_oracle():
    import emu_binding
    N = 24
    orc, lc, model, ter = make_oracle(_cfg(), N, seed=9)
    emu = emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins)

    def put(k, v):
        emu.buf[k][...] = v

    def setc(v):
        emu.step_counter = v
    _check(orc, emu, lambda k: emu.buf[k], put, lambda: emu.stats_row, setc)


@pytest.mark.gpu
def test_hip_reset_envs_matches_oracle():
    from hip_backend import HipBackend
    N = 24
    cfg = _cfg()
    orc, lc, model, ter = make_oracle(cfg, N, seed=9)
    be = HipBackend(cfg, N, ter, seed=9)

    def setc(v):
        be.step_counter = v
    _check(orc, be, be.get, be.put, lambda: be.stats_row, setc, exact=False)


@pytest.mark.gpu
def test_legged_robot_reset_idx_accepts_a_subset():
    """the host-side mirror: env.reset_idx(ids) = lsim_reset_envs on the mask of ids; extras["episode"] refreshed (LR:346-356); an empty
    list returns at once (LR:298); the graph rollout's pending work is flushed first"""
    import torch
    from hip_backend import HipBackend
    N = 16
    cfg = _cfg()
    orc, lc, model, ter = make_oracle(cfg, N, seed=4)
    be = HipBackend(cfg, N, ter, seed=4)
    env = be.env
    orc.reset_all(); be.reset_all()
    _walk(orc, be, be.get, be.put, 4, seed=1)
    ids = torch.tensor([9, 1, 14], device="cuda:0")
    mask = np.zeros(N, np.uint8); mask[[1, 9, 14]] = 1
    env.extras.pop("episode", None)
    calls = []
    env.before_external_step = lambda: calls.append(1)
    env.reset_idx(ids)
    env.before_external_step = None
    torch.cuda.synchronize()
    assert calls == [1]
    orc.reset_envs(mask)
    for k in EXACT:
        _same(be.get(k), orc.buf[k], False, k)
    ref = orc.buf["stats"][orc.stats_row]
    for name, idx in abi.REWARD_IDS.items():
        if "rew_" + name in env.extras["episode"]:
            want = ref[S["episode_sums"] + idx] / 3.0 / env.dt
            assert float(env.extras["episode"]["rew_" + name]) == pytest.approx(want, rel=1e-4, abs=1e-6), name
    row = be.stats_row
    env.reset_idx([])
    assert be.stats_row == row


def test_state_dict_carries_the_by_hand_reset_salt():
    """the per-call salt of lsim_reset_envs is part of LeggedRobot.state_dict(): a resumed run redraws the SAME by-hand
    reset states as the uninterrupted one.  CPU leg (lane emulator behind the product's LeggedRobot class): env A resets by hand twice, a
    fresh env B takes A's state_dict; the third by-hand reset of both draws the same joint state, and differs from a B without the salt."""
    import torch
    from emu_env import EmuLeggedRobot
    cfg = _cfg()
    cfg.env.num_envs = 8

    def make():
        return EmuLeggedRobot(cfg, seed=6)
    a = make()
    a.reset()
    a.reset_idx(torch.tensor([1, 2])); a.reset_idx(torch.tensor([2]))
    sd = a.state_dict()
    assert sd["reset_calls"] == 2
    b, c = make(), make()
    b.reset(); c.reset()
    b.load_state_dict(sd)
    assert b.state_dict()["reset_calls"] == 2
    sd0 = dict(sd); sd0.pop("reset_calls")
    c.load_state_dict(sd0)                                    # a checkpoint from before round 5: salt 0
    for e in (a, b, c):
        e.reset_idx(torch.tensor([5]))
    assert torch.equal(a.dof_pos[5], b.dof_pos[5]) and torch.equal(a.root_states[5], b.root_states[5])
    assert not torch.equal(a.dof_pos[5], c.dof_pos[5])
    for e in (a, b, c):
        e.close()
