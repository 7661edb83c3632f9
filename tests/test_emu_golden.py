"""CPU lane-emulation of the HIP kernel sources (isaacgymloco_amd/csrc/ls_*.h) against the reference golden
vectors and against the oracle's physics.  Development aid for the GPU kernels' lane orchestration; the
authoritative parity tests are the -m gpu ones that call the HIP library through the C-ABI."""
import numpy as np
import pytest

import golden_replay as GR
from helpers import LC, aliengo, make_oracle, quiet_cfg, abi


def make_emu_from_fixture(name, big=False):
    import emu_binding
    fx = GR.load_big(name) if big else GR.load(name)
    cfg = GR.big_scenario_cfg(name) if big else GR.scenario_cfg(name)
    N = int(fx["num_envs"])
    model = aliengo.build_model()
    ter = GR.FixtureTerrain(fx)
    lc = LC.make_lsim_config(cfg, num_envs=N, terrain=ter, model=model, seed=int(fx["seed"]))
    return fx, emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins)


FORMS = {"fused": 0, "two_kernels": abi.STEP_TWO_KERNELS}      # kernel A with the fused tail + the finish kernel (default) / kernels A + B on every step


@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("name", GR.SCENARIOS)
def test_emu_matches_reference_step(name, form):
    fx, sim = make_emu_from_fixture(name)

    def get(n):
        return np.array(sim.buf[n])

    def put(n, a):
        sim.buf[n][...] = a
    for t, ref in GR.replay(fx, sim, get, put, extra_flags=FORMS[form]):
        GR.compare_step(t, ref, get, sim.stats_row)
    if "fin_ids" in fx.files:     # the reference's by-hand reset_idx(env_ids): kernel B's masked mode (lsim_reset_envs)
        mask = GR.replay_final_reset(fx, sim, get, put)
        GR.compare_final_reset(fx, mask, get, sim.stats_row)


@pytest.mark.parametrize("name", GR.BIG_SCENARIOS)
def test_emu_matches_reference_step_at_baseline_size(name):
    fx, sim = make_emu_from_fixture(name, big=True)

    def get(n):
        return np.array(sim.buf[n])

    def put(n, a):
        sim.buf[n][...] = a
    for t, ref in GR.replay(fx, sim, get, put):
        GR.compare_step(t, ref, get, sim.stats_row)


def _recover_task_check(make_backend, get, put, steps=12, N=16):
    """task "aliengo_recover" (aliengo_recover_config.py): robots reset in ANY orientation (LR:786-794 with +-3.14 ranges), no termination on
    contact, the `_up` reward variants active -- the trunk and the upper legs carry the robot, which the walking tasks never exercise.
    A robot thrashing on its back is chaotic, so (as in the stairs tests) the states are re-synchronised every step and the bar is the share
    of env-steps within the fp32 tolerance."""
    from helpers import C
    cfg = C.TASKS["aliengo_recover"][0]()
    orc, lc, model, ter = make_oracle(cfg, N, seed=3)
    assert model.termination_body_mask == 0 and lc.heading_command == 0 and lc.only_positive_rewards == 1
    active = {abi.REWARD_NAMES[i] for i in range(abi.NUM_REWARD_TERMS) if lc.reward_scales[i] != 0}
    assert {"upward", "stand_nice", "orientation_up", "base_height_up", "has_contact"} <= active and "orientation" not in active
    be = make_backend(cfg, lc, model, ter, N)
    orc.reset_all(); be.reset_all()
    rs = np.random.RandomState(0)
    trunk_load = 0.0
    ok = tot = 0
    for t in range(steps):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        for k in ("root_states", "dof_state", "commands", "last_actions", "last_last_actions", "episode_length", "terrain_levels", "env_origins",
                  "kp_factors", "kd_factors", "friction", "pending_force", "feet_air_time", "last_contacts", "episode_sums", "obs", "last_dof_vel",
                  "last_dof_pos", "last_torques", "last_root_vel"):
            put(be, k, orc.buf[k])
        orc.step(a); be.step(a)
        same = get(be, "reset") == orc.buf["reset"]
        e_root = np.abs(get(be, "root_states") - orc.buf["root_states"]).max(1)
        e_rew = np.abs(get(be, "rew") - orc.buf["rew"])
        e_obs = np.abs(get(be, "obs") - orc.buf["obs"]).max(1)
        ok += int((same & (e_root < 2e-3) & (e_rew < 1e-3 + 1e-3 * np.abs(orc.buf["rew"])) & (e_obs < 5e-3)).sum()); tot += N
        trunk_load = max(trunk_load, float(orc.buf["contact_forces"][:, 0, 2].max()))
    print(f"aliengo_recover: {ok} of {tot} env-steps within tolerance")
    assert ok >= 0.97 * tot, (ok, tot)
    assert (orc.buf["projected_gravity"][:, 2] > 0.3).any(), "some robot must be on its back"
    assert trunk_load > 20.0, "a robot on its back rests on the trunk"


def test_emu_recover_task_matches_oracle():
    import emu_binding
    def put(be, k, v):
        be.buf[k][...] = v
    _recover_task_check(lambda cfg, lc, model, ter, N: emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins), lambda be, k: be.buf[k], put)


@pytest.mark.parametrize("task", ["aliengo", "aliengo_stairs"])
def test_emu_fused_tail_equals_the_two_kernel_form(task):
    """the per-env work of kernel B run by kernel A's wave (default) against kernels A + B (LSIM_STEP_TWO_KERNELS; the form command-curriculum
    steps always take): full physics, resets, a command-curriculum step in the window -- EVERY buffer bit for bit after every step"""
    import emu_binding
    from helpers import C
    N = 24
    sims = []
    for _ in range(2):
        cfg = C.TASKS[task][0]()
        cfg.env.episode_length_s = 0.4          # time-out resets inside the window
        orc, lc, model, ter = make_oracle(cfg, N, seed=9)
        orc.close()
        sims.append(emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins))
    a_, b_ = sims
    a_.reset_all(); b_.reset_all()
    rs = np.random.RandomState(3)
    resets = 0
    for t in range(30):
        if t == 12:      # the next step evaluates the command curriculum (LR:307): the default form falls back to two kernels there
            a_.step_counter = b_.step_counter = int(lc.max_episode_length) * 2 - 1
        act = rs.normal(0, 1, (N, 12)).astype(np.float32)
        a_.step(act); b_.step(act, flags=abi.STEP_TWO_KERNELS)
        assert a_.stats_row == b_.stats_row
        for name in abi.BUFFER_IDS:
            if name == "stats":      # the ticket word is the two-kernel form's own bookkeeping
                ra, rb = np.array(a_.buf[name]), np.array(b_.buf[name])
                np.testing.assert_array_equal(ra[:, :abi.STATS["fix"]], rb[:, :abi.STATS["fix"]], err_msg=f"step {t} stats")
                continue
            np.testing.assert_array_equal(np.array(a_.buf[name]), np.array(b_.buf[name]), err_msg=f"step {t} buffer {name}")
        resets += int(np.array(a_.buf["reset"]).sum())
    assert resets > N // 2
