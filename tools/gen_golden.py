"""Container-only: capture golden vectors of the reference's environment step.

The reference's `LeggedRobot` (legged_gym/envs/base/legged_robot.py) is instantiated WITHOUT Isaac Gym:
its gym handle is a stub whose `acquire_*_tensor` calls hand back tensors owned by this script, every
other gym call is a no-op.  `LeggedRobot.step(actions)` then executes exactly the reference's own
torch code (delay model, 4x `_compute_torques`, the whole `post_physics_step()` with callback,
termination, rewards, termination observations, `reset_idx`, observations) on the simulator state we
inject -- i.e. the reference run with physics frozen, which is what `LSIM_STEP_SKIP_PHYSICS` does
on our side.

Random numbers: the reference draws from torch's global generator, our kernels from counter-based
Philox streams (include/lsim.h).  To compare value-for-value the script patches the reference's draw
functions (`torch_rand_float`, `torch.rand_like`, `torch.randint`, `torch.randint_like`) so that each
call site receives the Philox uniforms of the matching (env, step, tag, idx) -- "inject the random
tensors" (SURVEY.md 8a quirk 12).

Injected simulator states are produced by the build's own CPU physics (oracle) so that contacts,
velocities and falls are realistic; they are inputs, stored in the fixture.

Output: tests/golden/step_<scenario>.npz  (inputs + every reference output per step).
Only data is written -- no reference source text.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import refenv  # noqa: E402

refenv.install()
import legged_gym.envs  # noqa: E402,F401
from legged_gym.envs.base import legged_robot as LRmod  # noqa: E402
from legged_gym.envs.base.legged_robot import LeggedRobot  # noqa: E402
from legged_gym.utils.terrain import Terrain as RefTerrain  # noqa: E402
from legged_gym.envs.aliengo import aliengo_config, aliengo_stairs_config, aliengo_amp_config  # noqa: E402
from isaacgym import gymapi  # noqa: E402  (stub)

import philox_np  # noqa: E402
from helpers import make_oracle  # noqa: E402
from isaacgymloco_amd import abi  # noqa: E402
from isaacgymloco_amd.envs import config as C  # noqa: E402
from isaacgymloco_amd.robots import aliengo  # noqa: E402

TAG = abi.RNG_TAGS
GOLDEN = os.path.join(ROOT, "tests", "golden")


# ----------------------------------------------------------------------------- RNG injection
class RngCtx:
    def __init__(self, seed, rank, num_envs):
        self.seed, self.rank, self.N = seed, rank, num_envs
        self.stepw = 0
        self.plan = []          # list of (tag, idx_base, env_ids or None)
        self.in_reset = False
        self.active = True      # False: draws fall through to torch's own generator (init-time, values overwritten later)

    def uniforms(self, shape):
        tag, base, env_ids = self.plan.pop(0)
        envs = np.arange(self.N) if env_ids is None else np.asarray(env_ids, dtype=np.int64)
        shape = tuple(shape)
        rows = shape[0]
        cols = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        assert rows == len(envs), (tag, shape, len(envs))
        u = philox_np.u01(self.seed, self.rank, envs[:, None], self.stepw, tag, base + np.arange(cols)[None, :])
        return torch.from_numpy(u.reshape(shape).astype(np.float32))


CTX = None
_ORIG = dict(rand_like=torch.rand_like, randint=torch.randint, randint_like=torch.randint_like, rand=torch.rand)


def _planned():
    return CTX is not None and CTX.active


def patched_torch_rand_float(lower, upper, shape, device):
    if not _planned():
        return (upper - lower) * _ORIG["rand"](*shape) + lower
    return (upper - lower) * CTX.uniforms(shape) + lower


def patched_rand_like(t, **kw):
    if not _planned():
        return _ORIG["rand_like"](t, **kw)
    return CTX.uniforms(t.shape)


def patched_randint(low, high, size, device=None, **kw):
    if not _planned():
        return _ORIG["randint"](low, high, size, **kw)
    tag = CTX.plan[0][0]
    u = CTX.uniforms(size)
    r = (u * float(high - low)).to(torch.long) + low
    if tag == TAG["delay"]:
        CTX.last_delay = r.clone()     # LR:134 keeps delay_steps in a local: recorded here for the fixture
    return r


def patched_randint_like(t, high, **kw):
    if not _planned():
        return _ORIG["randint_like"](t, high, **kw)
    u = CTX.uniforms(t.shape)
    return (u * float(high)).to(t.dtype)


def wrap(cls, name, before=None, after=None):
    orig = getattr(cls, name)

    def f(self, *a, **k):
        if before:
            before(self, *a, **k)
        r = orig(self, *a, **k)
        if after:
            after(self, *a, **k)
        return r
    setattr(cls, name, f)
    return orig


def install_rng_patches():
    LRmod.torch_rand_float = patched_torch_rand_float
    torch.rand_like = patched_rand_like
    torch.randint = patched_randint
    torch.randint_like = patched_randint_like

    def np_ids(env_ids):
        return env_ids.cpu().numpy()

    def before_step(self, actions):
        CTX.stepw = self.common_step_counter + 1
        CTX.plan = [(TAG["delay"], 0, None)]
        CTX.sub_torques = []
    wrap(LeggedRobot, "step", before=before_step)

    orig_ct = LeggedRobot._compute_torques

    def recording_compute_torques(self, actions):      # LR:146: one call per sub-step; the reference keeps only the last result
        t = orig_ct(self, actions)
        if CTX is not None and hasattr(CTX, "sub_torques"):
            CTX.sub_torques.append(t.detach().clone().view(self.num_envs, -1))
        return t
    LeggedRobot._compute_torques = recording_compute_torques

    def before_resample(self, env_ids):
        tag = TAG["reset_cmd"] if CTX.in_reset else TAG["cmd"]
        ids = np_ids(env_ids)
        CTX.plan = [(tag, 0, ids), (tag, 1, ids), (tag, 2, ids), (tag, 3, ids[ids < self.num_envs * 0.2])]

    def after_resample(self, env_ids):
        if CTX.in_reset:  # raw draws in reset_idx body (LR:336-341)
            ids = np_ids(env_ids)
            plan = []
            if self.cfg.domain_rand.randomize_kp:
                plan.append((TAG["reset_dr"], 0, ids))
            if self.cfg.domain_rand.randomize_kd:
                plan.append((TAG["reset_dr"], 1, ids))
            if self.cfg.domain_rand.randomize_motor_strength:
                plan.append((TAG["reset_dr"], 2, ids))
            CTX.plan = plan
    wrap(LeggedRobot, "_resample_commands", before=before_resample, after=after_resample)

    wrap(LeggedRobot, "_push_robots", before=lambda self: setattr(CTX, "plan", [(TAG["push"], 0, None)]))
    wrap(LeggedRobot, "_disturbance_robots", before=lambda self: setattr(CTX, "plan", [(TAG["disturb"], 0, None)]))
    wrap(LeggedRobot, "compute_termination_observations",
         before=lambda self, env_ids: setattr(CTX, "plan", ([(TAG["term_noise"], 0, None)] if self.add_noise else []) + [(TAG["term_noise"], 45, None)]))
    wrap(LeggedRobot, "compute_observations",
         before=lambda self: setattr(CTX, "plan", ([(TAG["obs_noise"], 0, None)] if self.add_noise else []) + [(TAG["obs_noise"], 45, None)]))
    wrap(LeggedRobot, "_update_terrain_curriculum",
         before=lambda self, env_ids: setattr(CTX, "plan", [(TAG["reset_level"], 0, np_ids(env_ids))]))

    def before_reset_dofs(self, env_ids):
        ids = np_ids(env_ids)
        CTX.plan = [(TAG["reset_dof"], 0, ids), (TAG["reset_dof"], 12, ids)]
    wrap(LeggedRobot, "_reset_dofs", before=before_reset_dofs)

    def before_reset_root(self, env_ids):
        ids = np_ids(env_ids)
        CTX.plan = [(TAG["reset_root"], k, ids) for k in range(12)]
    wrap(LeggedRobot, "_reset_root_states", before=before_reset_root)

    def before_shape_props(self, env_ids):
        ids = np_ids(env_ids)
        plan = []
        if self.cfg.domain_rand.randomize_friction:
            plan.append((TAG["reset_dr"], 3, ids))
        if self.cfg.domain_rand.randomize_restitution:
            plan.append((TAG["reset_dr"], 4, ids))
        CTX.plan = plan
    wrap(LeggedRobot, "refresh_actor_rigid_shape_props", before=before_shape_props)

    def before_reset_idx(self, env_ids):
        CTX.in_reset = True

    def after_reset_idx(self, env_ids):
        CTX.in_reset = False
    wrap(LeggedRobot, "reset_idx", before=before_reset_idx, after=after_reset_idx)


# ----------------------------------------------------------------------------- reference env without Isaac Gym
class GenGym:
    """gym handle whose acquire_* calls return tensors owned by the generator; the rest are no-ops."""

    def __init__(self, tensors):
        self._t = tensors

    def acquire_actor_root_state_tensor(self, sim):
        return self._t["root"]

    def acquire_dof_state_tensor(self, sim):
        return self._t["dof"]

    def acquire_net_contact_force_tensor(self, sim):
        return self._t["contact"]

    def acquire_rigid_body_state_tensor(self, sim):
        return self._t["body"]

    def get_actor_rigid_shape_properties(self, env, actor):
        return []

    def __getattr__(self, name):
        return lambda *a, **k: None


def build_reference_env(ref_cfg, terrain, model, N):
    env = object.__new__(LeggedRobot)
    env.cfg = ref_cfg
    ref_cfg.env.num_envs = N
    props = ref_cfg.terrain.terrain_proportions
    import math
    names = ["flat", "rough", "smoothslope", "roughslope", "stairsup", "stairsdown", "discreteobstacles", "steppingstones", "pit", "gap"]
    start = 0
    for k, nm in enumerate(names):  # LR:70-90 (index bookkeeping, data only)
        end = math.ceil(N * sum(props[:k + 1])) if k < 9 else N
        setattr(env, nm + "_start_idx", start)
        setattr(env, nm + "_end_idx", end)
        start = end
    env.sim_params = gymapi.SimParams()
    env.sim_params.dt = ref_cfg.sim.dt
    env.height_samples = None
    env.debug_viz = False
    env.init_done = False
    env._parse_cfg(ref_cfg)
    # BaseTask.__init__ buffer allocation (BT:57-79) without gym
    env.device = "cpu"
    env.headless = True
    env.num_envs, env.num_obs, env.num_privileged_obs, env.num_actions = N, ref_cfg.env.num_observations, ref_cfg.env.num_privileged_obs, ref_cfg.env.num_actions
    env.obs_buf = torch.zeros(N, env.num_obs)
    env.rew_buf = torch.zeros(N)
    env.reset_buf = torch.ones(N, dtype=torch.long)
    env.episode_length_buf = torch.zeros(N, dtype=torch.long)
    env.time_out_buf = torch.zeros(N, dtype=torch.bool)
    env.privileged_obs_buf = torch.zeros(N, env.num_privileged_obs)
    env.extras = {}
    env.viewer = None
    env.enable_viewer_sync = True
    env.sim = None
    env.num_one_step_obs = ref_cfg.env.num_one_step_observations
    env.num_one_step_privileged_obs = ref_cfg.env.num_one_step_privileged_obs
    env.history_length = int(env.num_obs / env.num_one_step_obs)
    # create_sim / _create_envs products (LR:463-482, LR:1107-1219) supplied as data
    env.up_axis_idx = 2
    rt = object.__new__(RefTerrain)  # the reference's Terrain class, filled with our grid (its generators need isaacgym)
    rt.cfg = ref_cfg.terrain
    rt.env_length, rt.env_width = ref_cfg.terrain.terrain_length, ref_cfg.terrain.terrain_width
    rt.xSize = ref_cfg.terrain.terrain_length * ref_cfg.terrain.num_rows
    rt.ySize = ref_cfg.terrain.terrain_width * ref_cfg.terrain.num_cols
    rt.heightsamples = terrain.heightsamples
    rt.env_origins = terrain.env_origins
    rt.tot_rows, rt.tot_cols = terrain.tot_rows, terrain.tot_cols
    env.terrain = rt
    env.height_samples = torch.tensor(terrain.heightsamples).view(terrain.tot_rows, terrain.tot_cols)
    env.num_dof = env.num_dofs = 12
    env.num_bodies = 17
    env.dof_names = list(aliengo.DOF_NAMES)
    env.envs = [None] * N
    env.actor_handles = [None] * N
    env.feet_indices = torch.tensor(list(model.feet_bodies), dtype=torch.long)
    pen = [i for i in range(17) if (model.penalised_body_mask >> i) & 1]
    # LR:1151-1153 order: all "thigh", then "calf", then "base"
    order = [i for i in pen if "thigh" in aliengo.BODY_NAMES[i]] + [i for i in pen if "calf" in aliengo.BODY_NAMES[i]] + [i for i in pen if "base" in aliengo.BODY_NAMES[i]]
    env.penalised_contact_indices = torch.tensor(order, dtype=torch.long)
    env.termination_contact_indices = torch.tensor([i for i in range(17) if (model.termination_body_mask >> i) & 1], dtype=torch.long)
    lim = torch.zeros(12, 2)
    env.dof_vel_limits = torch.zeros(12)
    env.torque_limits = torch.zeros(12)
    for j in range(12):  # _process_dof_props LR:560-578 (soft limits)
        lo, hi = model.dof_pos_lower[j], model.dof_pos_upper[j]
        lim[j, 0], lim[j, 1] = lo, hi
        env.dof_vel_limits[j] = model.dof_vel_limit[j]
        env.torque_limits[j] = model.dof_effort_limit[j]
        m = (lim[j, 0] + lim[j, 1]) / 2
        r = lim[j, 1] - lim[j, 0]
        lim[j, 0] = m - 0.5 * r * ref_cfg.rewards.soft_dof_pos_limit
        lim[j, 1] = m + 0.5 * r * ref_cfg.rewards.soft_dof_pos_limit
    env.dof_pos_limits = lim
    base_init = ref_cfg.init_state.pos + ref_cfg.init_state.rot + ref_cfg.init_state.lin_vel + ref_cfg.init_state.ang_vel
    env.base_init_state = torch.tensor(base_init, dtype=torch.float)
    env.custom_origins = True
    env.max_terrain_level = ref_cfg.terrain.num_rows
    env.terrain_origins = torch.from_numpy(terrain.env_origins).to(torch.float)
    env.default_rigid_body_mass = torch.tensor([b.mass for b in model.bodies])
    tensors = dict(root=torch.zeros(N, 13), dof=torch.zeros(N * 12, 2), contact=torch.zeros(N * 17, 3), body=torch.zeros(N * 17, 13))
    tensors["root"][:, 6] = 1.0
    env.gym = GenGym(tensors)
    # terrain levels/types/origins placeholders needed by _init_buffers -> _get_heights
    env.terrain_levels = torch.zeros(N, dtype=torch.long)
    env.terrain_types = torch.zeros(N, dtype=torch.long)
    env.env_origins = torch.zeros(N, 3)
    env._init_buffers()              # the reference's own buffer set-up (LR:913-1032)
    env._prepare_reward_function()   # LR:1035-1059
    env.init_done = True
    return env, tensors


def sync_initial_state(env, tensors, orc):
    """copy the oracle's init-time draws into the reference so both start from identical state"""
    b = orc.buf
    t = lambda a: torch.from_numpy(np.array(a))  # noqa: E731
    env.motor_strength = t(b["motor_strength"])
    env.Kp_factors = t(b["kp_factors"]).unsqueeze(1)
    env.Kd_factors = t(b["kd_factors"]).unsqueeze(1)
    env.motor_strength_factors = t(b["motor_strength_factors"]).unsqueeze(1)
    env.payload = t(b["payload"]).unsqueeze(1)
    env.com_displacement = t(b["com_displacement"])
    env.friction_coeffs = t(b["friction"]).unsqueeze(1)
    env.restitution_coeffs = t(b["restitution"]).unsqueeze(1)
    env.terrain_levels = t(b["terrain_levels"])
    env.terrain_types = t(b["terrain_types"])
    env.env_origins = t(b["env_origins"])
    tensors["root"][:] = t(b["root_states"])
    tensors["dof"][:] = t(b["dof_state"]).view(-1, 2)


def inject_state(tensors, root, dof, body, contact):
    tensors["root"][:] = torch.from_numpy(root)
    tensors["dof"][:] = torch.from_numpy(dof).view(-1, 2)
    tensors["body"][:] = torch.from_numpy(body).view(-1, 13)
    tensors["contact"][:] = torch.from_numpy(contact).view(-1, 3)


OUT_KEYS = ["obs", "priv_obs", "rew", "reset", "time_out", "extras_time_outs", "term_ids", "term_priv_obs", "term_amp", "amp_obs",
            "commands", "torques", "base_lin_vel", "base_ang_vel", "projected_gravity", "feet_air_time", "last_contacts",
            "contact_filt", "measured_heights", "root_states", "dof_state", "terrain_levels", "env_origins", "episode_length",
            "kp_factors", "kd_factors", "friction", "last_actions", "last_last_actions", "last_dof_vel", "episode_sums",
            "command_ranges", "ep_stats", "level_mean", "delayed_actions", "delay_steps", "substep_torques"]


def capture(env, tensors, N):
    names = abi.REWARD_NAMES
    es = np.zeros((N, len(names)), np.float32)
    for k, nm in enumerate(names):
        if nm in env.episode_sums:
            es[:, k] = env.episode_sums[nm].numpy()
    ep = np.full(len(names), np.nan, np.float32)
    level_mean = np.float32(np.nan)
    if "episode" in env.extras:
        for k, nm in enumerate(names):
            if "rew_" + nm in env.extras["episode"]:
                ep[k] = float(env.extras["episode"]["rew_" + nm])
        if "terrain_level" in env.extras["episode"]:
            level_mean = np.float32(float(env.extras["episode"]["terrain_level"]))
    cr = env.command_ranges
    return dict(
        obs=env.obs_buf.numpy().copy(), priv_obs=env.privileged_obs_buf.numpy().copy(), rew=env.rew_buf.numpy().copy(),
        reset=env.reset_buf.numpy().astype(np.uint8), time_out=env.time_out_buf.numpy().astype(np.uint8),
        extras_time_outs=(env.extras["time_outs"].numpy().astype(np.uint8) if "time_outs" in env.extras else np.zeros(N, np.uint8)),
        commands=env.commands.numpy().copy(), torques=env.torques.numpy().copy(),
        base_lin_vel=env.base_lin_vel.numpy().copy(), base_ang_vel=env.base_ang_vel.numpy().copy(),
        projected_gravity=env.projected_gravity.numpy().copy(), feet_air_time=env.feet_air_time.numpy().copy(),
        last_contacts=env.last_contacts.numpy().astype(np.uint8), contact_filt=env.contact_filt.numpy().astype(np.uint8),
        measured_heights=env.measured_heights.numpy().copy(), root_states=tensors["root"].numpy().copy(),
        dof_state=tensors["dof"].numpy().reshape(N, 12, 2).copy(), terrain_levels=env.terrain_levels.numpy().copy(),
        env_origins=env.env_origins.numpy().copy(), episode_length=env.episode_length_buf.numpy().copy(),
        kp_factors=env.Kp_factors.numpy()[:, 0].copy(), kd_factors=env.Kd_factors.numpy()[:, 0].copy(),
        friction=env.friction_coeffs.numpy()[:, 0].copy(), last_actions=env.last_actions.numpy().copy(),
        last_last_actions=env.last_last_actions.numpy().copy(), last_dof_vel=env.last_dof_vel.numpy().copy(),
        episode_sums=es, ep_stats=ep, level_mean=level_mean, amp_obs=env.get_amp_observations().numpy().copy(),
        command_ranges=np.array([cr["lin_vel_x"], cr["lin_vel_y"], cr["ang_vel_yaw"], cr["heading"]], dtype=np.float64),
        delayed_actions=env.delayed_actions.numpy().copy(),                                   # (N, 4, 12), LR:133-138
        delay_steps=CTX.last_delay.numpy().reshape(N).astype(np.int32),                        # the draw of LR:134
        substep_torques=torch.stack(CTX.sub_torques[-env.cfg.control.decimation:], dim=1).numpy().copy(),   # (N, 4, 12), LR:146 per sub-step
    )


def run_scenario(name, task, ref_cfg_cls, N, segments, seed=1, tweak=None, using_amp=False, final_reset_ids=None):
    """segments: list of (start_counter, num_steps).  final_reset_ids: after the last step the reference's reset_idx(ids) is called BY HAND
    (LR:290, outside step()) and what it left behind is stored under fin_* (the reference-side pin of lsim_reset_envs)."""
    global CTX
    LRmod.USING_AMP = using_amp   # LR:173: step() returns the 8-tuple with terminal AMP states
    cfg = C.TASKS[task][0]()
    if tweak:
        tweak(cfg)
    gen, lc, model, terrain = make_oracle(cfg, N, seed=seed)          # state generator with real (oracle) physics
    orc, _, _, _ = make_oracle(cfg, N, seed=seed)                     # only used for its init-time draws here
    ref_cfg = ref_cfg_cls()
    if tweak:
        tweak(ref_cfg)
    CTX = None
    env, tensors = build_reference_env(ref_cfg, terrain, model, N)
    sync_initial_state(env, tensors, orc)
    CTX = RngCtx(seed, 0, N)
    rs = np.random.RandomState(1234)

    # runner start-up (HIMR:84, BT:111-115): reset_idx(all) + one zero-action step
    CTX.stepw = 0
    env.reset_idx(torch.arange(N))
    gen.reset_all()
    steps = []
    first = True
    for seg, (start_counter, nsteps) in enumerate(segments):
        if not first:
            env.common_step_counter = start_counter
            gen.step_counter = start_counter
            # spread episode lengths so that time-outs / resampling boundaries are crossed (HIMR:90-91 does the same)
            ep = rs.randint(0, 1003, size=N).astype(np.int64)
            ep[:4] = [499, 999, 1000, 1001][:min(4, N)]
            env.episode_length_buf = torch.from_numpy(ep.copy())
            gen.buf["episode_length"][:] = ep
        for i in range(nsteps):
            if first:
                actions = np.zeros((N, 12), np.float32)
            else:
                actions = rs.normal(0, 1.0, size=(N, 12)).astype(np.float32)
                if i % 5 == 0:
                    actions[0] *= 300.0   # exercises clip_actions
            counter_before = env.common_step_counter
            if counter_before + 1 == 1000 and N > 3:            # guarantee a reset on the curriculum step (LR:307)
                env.episode_length_buf[3] = 1000
            ep_before = env.episode_length_buf.numpy().copy()
            gen.step(actions)                                   # evolves a realistic simulator state
            root = gen.buf["root_states"].copy(); dof = gen.buf["dof_state"].copy()
            body = gen.buf["rigid_body_states"].copy(); contact = gen.buf["contact_forces"].copy()
            if not first and i == 2 and N > 6:                  # edge cases: out of border, fall-down, vel violation
                root[5, 0] = -3.0
                root[6, 9] = -6.0
                root[2, 7] = 4.0
                root[4, 7] = -4.0
            if not first and i == 3:                            # feet_stumble: lateral >> vertical foot force
                contact[:8, 4, :] = np.array([30.0, 0.0, 2.0], np.float32)
            track_override = np.float32(np.nan)
            if counter_before + 1 == 1000:                      # make the command curriculum fire (LR:875)
                track_override = np.float32(30.0)
                if "tracking_lin_vel" in env.episode_sums:
                    env.episode_sums["tracking_lin_vel"][:] = float(track_override)
            inject_state(tensors, root, dof, body, contact)
            ret = env.step(torch.from_numpy(actions))
            out = capture(env, tensors, N)
            term_ids = ret[5].numpy().copy()
            out["term_ids"] = term_ids
            tp = np.zeros((N, 238), np.float32); tp[term_ids] = ret[6].numpy()
            out["term_priv_obs"] = tp
            ta = np.zeros((N, 30), np.float32)
            if len(ret) > 7:
                ta[term_ids] = ret[7].numpy()
            else:
                ta[term_ids] = 0
            out["term_amp"] = ta
            feet = np.array(list(model.feet_bodies))
            steps.append(dict(inp=dict(actions=actions, root=root, dof=dof, body_feet=body[:, feet, :], contact=contact,
                                       ep_before=ep_before, counter_before=np.int64(counter_before),
                                       track_override=track_override), out=out))
            first = False
            # keep the generator's carried state aligned with what the reference did to the injected state (resets)
            gen.buf["root_states"][:] = tensors["root"].numpy()
            gen.buf["dof_state"][:] = tensors["dof"].numpy().reshape(N, 12, 2)
            gen.buf["episode_length"][:] = env.episode_length_buf.numpy()
            gen.buf["terrain_levels"][:] = env.terrain_levels.numpy()
            gen.buf["env_origins"][:] = env.env_origins.numpy()
    pack = {"num_envs": np.int64(N), "seed": np.int64(seed), "task": np.array(task),
            "height_grid": terrain.heightsamples, "terrain_origins": terrain.env_origins.astype(np.float32),
            "segments": np.array(segments, dtype=np.int64)}
    if final_reset_ids is not None:
        ids = np.asarray(final_reset_ids, dtype=np.int64)
        # the command curriculum averages over the reset SET (LR:307-308, LR:875): only the chosen envs carry a tracking sum above the bar
        track = np.zeros(N, np.float32)
        track[ids] = 1.25 * 0.8 * float(env.reward_scales["tracking_lin_vel"]) * float(env.max_episode_length)
        env.episode_sums["tracking_lin_vel"][:] = torch.from_numpy(track)
        env.extras.pop("episode", None)
        pack["fin_ids"] = ids
        pack["fin_track"] = track
        pack["fin_counter"] = np.int64(env.common_step_counter)
        pack["fin_ranges_before"] = np.array([env.command_ranges[k] for k in ("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading")], dtype=np.float64)
        before = capture(env, tensors, N)
        # draws keyed as lsim_reset_envs keys them: the step word of the last step, salted with the handle's by-hand-reset count
        # (the first such call here: 1 x 0x9E3779B9 -- include/lsim.h)
        CTX.stepw = (int(CTX.stepw) ^ 0x9E3779B9) & 0xFFFFFFFF
        env.reset_idx(torch.from_numpy(ids))
        after = capture(env, tensors, N)
        for k in ("commands", "root_states", "dof_state", "terrain_levels", "env_origins", "episode_length", "kp_factors", "kd_factors", "friction",
                  "last_actions", "last_last_actions", "last_dof_vel", "feet_air_time", "reset", "extras_time_outs", "time_out", "episode_sums",
                  "ep_stats", "level_mean", "command_ranges", "measured_heights"):
            pack["fin_" + k] = after[k]
        for k in ("root_states", "commands", "episode_length"):
            pack["finb_" + k] = before[k]
    for k in steps[0]["inp"]:
        pack["in_" + k] = np.stack([s["inp"][k] for s in steps])
    for k in steps[0]["out"]:
        if k == "term_ids":
            m = np.zeros((len(steps), N), np.uint8)
            for si, s in enumerate(steps):
                m[si, s["out"]["term_ids"]] = 1
            pack["out_term_mask"] = m
        else:
            pack["out_" + k] = np.stack([s["out"][k] for s in steps])
    path = os.path.join(GOLDEN, f"step_{name}.npz")
    np.savez_compressed(path, **pack)
    nres = int(pack["out_term_mask"].sum())
    print(f"wrote {path}: {len(steps)} steps, N={N}, resets={nres}, size={os.path.getsize(path) / 1e6:.2f} MB")


# keys stored for ALL envs of a BASELINE-size fixture (cheap, and they carry the index-dependent logic: high-velocity commands of the first 20 % of the
# envs LR:649, terrain columns LR:1234, the stumble slices LR:1597-1607 through `rew`); every other key is stored for the rows `sel` only
BIG_FULL_KEYS = ["rew", "reset", "time_out", "extras_time_outs", "commands", "terrain_levels", "env_origins", "episode_length", "delay_steps",
                 "kp_factors", "kd_factors", "friction", "term_mask", "command_ranges", "ep_stats", "level_mean", "stumble_sums"]


def big_selection(env, N):
    """every 37th env + a band of +-2 around every index boundary the reference's code has"""
    cols = env.cfg.terrain.num_cols
    bounds = {0, N - 1, int(N * 0.2)}                                        # LR:649: env_ids < num_envs * 0.2
    bounds |= {int(k * N / cols) for k in range(1, cols)}                     # LR:1234: terrain_types = floor(i / (N / cols))
    for nm in ("flat", "rough", "smoothslope", "roughslope", "stairsup", "stairsdown", "discreteobstacles", "steppingstones", "pit", "gap"):
        bounds |= {int(getattr(env, nm + "_start_idx")), int(getattr(env, nm + "_end_idx"))}   # LR:72-90 -> LR:1597-1607
    sel = set(range(0, N, 37))
    for b in bounds:
        sel |= {i for i in range(b - 2, b + 3) if 0 <= i < N}
    return np.array(sorted(sel), dtype=np.int64)


def run_scenario_big(name, task, ref_cfg_cls, N=4096, seed=1, tweak=None, using_amp=False, counters=(0, 999, 1000)):
    """BASELINE-size fixture: reset + zero-action step, then steps at the given counters (999 -> 1000 fires the command
    curriculum, LR:307).  Injected states come from tests/big_inputs.py (not stored); outputs are stored for all envs where cheap, else for `sel`."""
    global CTX
    import big_inputs
    LRmod.USING_AMP = using_amp
    cfg = C.TASKS[task][0]()
    if tweak:
        tweak(cfg)
    orc, lc, model, terrain = make_oracle(cfg, N, seed=seed)          # init-time draws only
    ref_cfg = ref_cfg_cls()
    if tweak:
        tweak(ref_cfg)
    CTX = None
    env, tensors = build_reference_env(ref_cfg, terrain, model, N)
    sync_initial_state(env, tensors, orc)
    init = dict(terrain_levels=env.terrain_levels.numpy().copy(), terrain_types=env.terrain_types.numpy().copy(), env_origins=env.env_origins.numpy().copy())
    CTX = RngCtx(seed, 0, N)
    CTX.stepw = 0
    env.reset_idx(torch.arange(N))
    sel = big_selection(env, N)
    steps, crcs = [], []
    for t, counter_before in enumerate(counters):
        env.common_step_counter = counter_before
        inp = big_inputs.synth_step_inputs(N, t, env.env_origins.numpy(), seed)
        crcs.append(big_inputs.crc_of_inputs(inp))
        if counter_before + 1 == 1000:
            inp["ep_before"][3] = 1000                              # a reset on the curriculum step (LR:307)
        env.episode_length_buf = torch.from_numpy(inp["ep_before"].copy())
        if t > 0:
            env.terrain_levels = torch.from_numpy(inp["terrain_levels"].copy())
        body = np.zeros((N, 17, 13), np.float32)
        body[:, list(model.feet_bodies), :] = inp["body_feet"]
        track_override = np.float32(np.nan)
        if counter_before + 1 == 1000:
            track_override = np.float32(30.0)
            if "tracking_lin_vel" in env.episode_sums:
                env.episode_sums["tracking_lin_vel"][:] = float(track_override)
        inject_state(tensors, inp["root"], inp["dof"], body, inp["contact"])
        ret = env.step(torch.from_numpy(inp["actions"]))
        out = capture(env, tensors, N)
        out["stumble_sums"] = out["episode_sums"][:, [abi.REWARD_IDS["feet_stumble"], abi.REWARD_IDS["feet_stumble_up"]]].copy()
        term_ids = ret[5].numpy().copy()
        m = np.zeros(N, np.uint8); m[term_ids] = 1
        out["term_mask"] = m
        tp = np.zeros((N, 238), np.float32); tp[term_ids] = ret[6].numpy()
        out["term_priv_obs"] = tp
        ta = np.zeros((N, 30), np.float32)
        if len(ret) > 7:
            ta[term_ids] = ret[7].numpy()
        out["term_amp"] = ta
        steps.append(dict(counter_before=np.int64(counter_before), track_override=track_override, out=out))
    pack = {"num_envs": np.int64(N), "seed": np.int64(seed), "task": np.array(task), "big": np.int64(1), "sel": sel,
            "height_grid": terrain.heightsamples, "terrain_origins": terrain.env_origins.astype(np.float32),
            "in_counter_before": np.array([s["counter_before"] for s in steps]), "in_track_override": np.array([s["track_override"] for s in steps]),
            "in_crc": np.array(crcs, dtype=np.uint32)}
    for k, v in init.items():
        pack["init_" + k] = v
    for k in steps[0]["out"]:
        arrs = [s["out"][k] for s in steps]
        if k not in BIG_FULL_KEYS:
            # fat keys: the selected rows + full-batch reductions (fp64 sum and sum of magnitudes per column)
            pack["red_sum_" + k] = np.stack([a.astype(np.float64).sum(0) for a in arrs])
            pack["red_abs_" + k] = np.stack([np.abs(a.astype(np.float64)).sum(0) for a in arrs])
            arrs = [a[sel] for a in arrs]
        pack["out_" + k] = np.stack(arrs)
    path = os.path.join(GOLDEN, f"step4096_{name}.npz")
    np.savez_compressed(path, **pack)
    print(f"wrote {path}: {len(steps)} steps, N={N}, |sel|={len(sel)}, resets={[int(s['out']['term_mask'].sum()) for s in steps]}, "
          f"size={os.path.getsize(path) / 1e6:.2f} MB")


def main():
    install_rng_patches()

    def flat_only(cfg):
        cfg.terrain.terrain_proportions = [1.0, 0.0, 0.0, 0.0]

    if sys.argv[1:] == ["reset_subset"]:     # only the by-hand reset_idx(env_ids) fixture
        run_scenario("aliengo_reset_subset", "aliengo", aliengo_config.AlienGoRoughCfg, 16, [(0, 3), (997, 3)], tweak=flat_only,
                     final_reset_ids=[1, 3, 4, 9, 15])
        return

    def stairs_only(cfg):   # only generators that exist in-tree (TER:229-294): stairs up/down
        cfg.terrain.terrain_proportions = [0.0, 0.0, 0.0, 0.0, 0.5, 0.5, 0.0, 0.0, 0.0, 0.0]

    run_scenario("aliengo_flat", "aliengo", aliengo_config.AlienGoRoughCfg, 16, [(0, 6), (795, 10), (996, 8)], tweak=flat_only)
    run_scenario("aliengo_stairs", "aliengo_stairs", aliengo_stairs_config.AlienGoStairsCfg, 16, [(0, 6), (795, 8), (996, 8)], tweak=stairs_only)

    def all_terms(cfg):     # every _reward_* function of LR:1444-1770 active (the 30 terms no shipped config enables)
        flat_only(cfg)
        for k, nm in enumerate(abi.REWARD_NAMES):
            setattr(cfg.rewards.scales, nm, (0.5 + 0.01 * k) * (-1.0 if k % 3 else 1.0))
        cfg.rewards.only_positive_rewards = True
    run_scenario("aliengo_allterms", "aliengo", aliengo_config.AlienGoRoughCfg, 16, [(0, 4), (795, 8)], tweak=all_terms)
    run_scenario("aliengo_amp", "aliengo_amp", aliengo_amp_config.AlienGoRoughCfg, 16, [(0, 4), (795, 8)], tweak=flat_only, using_amp=True)

    # BASELINE size (cfg 2-4): the same three tasks at N = 4096.  `aliengo` runs with every reward term on so that the stumble slices are live.
    def all_terms_10(cfg):   # 10-entry proportions of in-tree generators only (flat, stairs, pit, gap): the stairs-up / pit / gap slices of LR:1597-1607 are non-empty
        all_terms(cfg)
        cfg.terrain.terrain_proportions = [0.5, 0.0, 0.0, 0.0, 0.2, 0.1, 0.0, 0.0, 0.1, 0.1]
    run_scenario_big("aliengo", "aliengo", aliengo_config.AlienGoRoughCfg, tweak=all_terms_10)
    run_scenario_big("aliengo_stairs", "aliengo_stairs", aliengo_stairs_config.AlienGoStairsCfg, tweak=stairs_only)
    run_scenario_big("aliengo_amp", "aliengo_amp", aliengo_amp_config.AlienGoRoughCfg, tweak=flat_only, using_amp=True)


if __name__ == "__main__":
    main()
