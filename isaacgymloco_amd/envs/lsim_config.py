"""Flatten a task config (envs/config.py ConfigNode, same fields as the reference's LeggedRobotCfg)
into the C-ABI `lsim_config` struct, performing the same derived-value arithmetic the reference does
at construction time (LeggedRobot._parse_cfg LR:1252-1263, _init_buffers LR:913-1032,
_prepare_reward_function LR:1035-1059, __init__ LR:70-90) in Python double precision."""
import math
import os

import numpy as np

from .. import abi
from ..robots import aliengo

MESH_TYPES = {"plane": 0, "heightfield": 1, "trimesh": 2}
CONTROL_TYPES = {"P": 0, "V": 1, "T": 2}

# Solver parameters with no counterpart in the reference's config.  The solver itself follows cfg.sim.physx
# (LRC:245-247): solver_type 1 = TGS with num_position_iterations sub-iterations (every reference config), solver_type 0 = PGS -- there
# `solver_iterations` velocity-level sweeps (PhysX's own PGS counts are not comparable: its rows are relaxed per body pair).
# LSIM_SOLVER=pgs / tgs overrides the config (A/B measurements, tools/).
SOLVER_DEFAULTS = dict(solver_iterations=8, erp=0.2, contact_slop=0.001)
SOLVER_TYPES = {"pgs": 0, "tgs": 1}


def _get(node, name, default=None):
    return getattr(node, name, default) if node is not None else default


def make_lsim_config(cfg, num_envs=None, terrain=None, model=None, seed=1, rank=0, using_amp=False, dof_names=None):
    """cfg: ConfigNode (or any object with the reference's attribute tree).  terrain: envs.terrain.Terrain."""
    c = abi.LsimConfig()
    model = model or aliengo.build_model()
    dof_names = dof_names or aliengo.DOF_NAMES
    N = int(num_envs if num_envs is not None else cfg.env.num_envs)
    c.abi_version = abi.ABI_VERSION
    c.num_envs = N
    c.seed = seed
    c.rank = rank

    # --- control
    sim_dt = float(cfg.sim.dt)
    dt = cfg.control.decimation * sim_dt                                 # LR:1253
    c.sim_dt = sim_dt
    c.decimation = int(cfg.control.decimation)
    c.control_type = CONTROL_TYPES[cfg.control.control_type]
    c.action_scale = cfg.control.action_scale
    c.hip_reduction = cfg.control.hip_reduction
    for i, name in enumerate(dof_names):                                   # LR:980-995
        c.default_dof_pos[i] = cfg.init_state.default_joint_angles[name]
        kp = kd = 0.0
        for key in cfg.control.stiffness.keys():
            if key in name:
                kp, kd = cfg.control.stiffness[key], cfg.control.damping[key]
        c.p_gains[i], c.d_gains[i] = kp, kd
        c.torque_limits[i] = model.dof_effort_limit[i]                    # LR:571
    c.clip_actions = cfg.normalization.clip_actions
    c.clip_observations = cfg.normalization.clip_observations

    # --- domain randomisation
    dr = cfg.domain_rand
    c.delay = int(bool(_get(dr, "delay", False)))
    for flag, rng in (("randomize_kp", "kp_range"), ("randomize_kd", "kd_range"),
                      ("randomize_motor_strength", "motor_strength_range"), ("randomize_friction", "friction_range"),
                      ("randomize_restitution", "restitution_range"), ("randomize_payload_mass", "payload_mass_range"),
                      ("randomize_com_displacement", "com_displacement_range")):
        setattr(c, flag, int(bool(_get(dr, flag, False))))
        r = _get(dr, rng, [0.0, 0.0])
        getattr(c, rng)[0], getattr(c, rng)[1] = r[0], r[1]
    c.push_robots = int(bool(dr.push_robots))
    c.push_interval = int(np.ceil(dr.push_interval_s / dt))                # LR:1263
    c.max_push_vel_xy = dr.max_push_vel_xy
    c.disturbance = int(bool(dr.disturbance))
    c.disturbance_interval = int(dr.disturbance_interval)
    c.disturbance_range[0], c.disturbance_range[1] = dr.disturbance_range

    # --- reset
    ratio = _get(dr, "dof_init_pos_ratio_range", None)                     # LR:698
    c.has_dof_init_pos_ratio = int(ratio is not None)
    if ratio is not None:
        c.dof_init_pos_ratio_range[0], c.dof_init_pos_ratio_range[1] = ratio
    c.randomize_dof_vel = int(bool(_get(dr, "randomize_dof_vel", False)))
    vr = _get(dr, "init_dof_vel_range", [-1.0, 1.0])                       # LR:708 reads this (absent) key: default range
    c.dof_init_vel_range[0], c.dof_init_vel_range[1] = vr
    pos_r = _get(dr, "base_init_pos_range", None)
    c.has_base_init_pos_range = int(pos_r is not None)
    if pos_r is not None:
        for k, ax in enumerate("xyz"):
            c.base_init_pos_range[k][0], c.base_init_pos_range[k][1] = pos_r[ax]
    rot_r = _get(dr, "base_init_rot_range", None)
    c.has_base_init_rot_range = int(rot_r is not None)
    if rot_r is not None:
        for k, ax in enumerate(("roll", "pitch", "yaw")):
            r = rot_r.get(ax, [-math.pi, math.pi])
            c.base_init_rot_range[k][0], c.base_init_rot_range[k][1] = r
    vel_r = _get(dr, "base_init_vel_range", None) or (-0.5, 0.5)           # LR:773-776
    for k, ax in enumerate(("x", "y", "z", "roll", "pitch", "yaw")):
        r = vel_r[ax] if isinstance(vel_r, dict) else vel_r
        c.base_init_vel_range[k][0], c.base_init_vel_range[k][1] = r
    init = list(cfg.init_state.pos) + list(cfg.init_state.rot) + list(cfg.init_state.lin_vel) + list(cfg.init_state.ang_vel)
    for k in range(13):
        c.base_init_state[k] = init[k]

    # --- commands
    for k, name in enumerate(("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading")):
        r = getattr(cfg.commands.ranges, name)
        c.command_ranges[k][0], c.command_ranges[k][1] = r
    c.heading_command = int(bool(cfg.commands.heading_command))
    c.resampling_steps = int(cfg.commands.resampling_time / dt)            # LR:612
    c.commands_curriculum = int(bool(cfg.commands.curriculum))
    c.max_forward_curriculum = _get(cfg.commands, "max_forward_curriculum", 1.0)
    c.max_backward_curriculum = _get(cfg.commands, "max_backward_curriculum", 1.0)
    c.max_lat_curriculum = _get(cfg.commands, "max_lat_curriculum", 1.0)

    # --- terrain
    t = cfg.terrain
    c.mesh_type = MESH_TYPES[t.mesh_type]
    c.horizontal_scale, c.vertical_scale, c.border_size = t.horizontal_scale, t.vertical_scale, t.border_size
    if c.mesh_type != 0:
        if terrain is None:
            raise ValueError("a Terrain is required for heightfield/trimesh configs")
        c.grid_rows, c.grid_cols = terrain.tot_rows, terrain.tot_cols
    c.terrain_num_rows, c.terrain_num_cols = t.num_rows, t.num_cols
    c.terrain_length, c.terrain_width = t.terrain_length, t.terrain_width
    c.terrain_curriculum = int(bool(t.curriculum) and c.mesh_type != 0)  # LR:1258-1259
    c.max_init_terrain_level = t.max_init_terrain_level
    c.measure_heights = int(bool(t.measure_heights))
    c.num_points_x, c.num_points_y = len(t.measured_points_x), len(t.measured_points_y)
    for i, v in enumerate(t.measured_points_x):
        c.measured_points_x[i] = v
    for i, v in enumerate(t.measured_points_y):
        c.measured_points_y[i] = v
    c.slope_threshold = t.slope_treshold
    c.terrain_friction = t.static_friction
    c.terrain_restitution = t.restitution

    # --- termination (LR:266-282: every check is gated on hasattr(cfg, "termination"))
    term = _get(cfg, "termination", None)
    c.term_base_vel_violate_commands = int(bool(_get(term, "base_vel_violate_commands", False)))
    c.term_out_of_border = int(bool(_get(term, "out_of_border", False)))
    c.term_fall_down = int(bool(_get(term, "fall_down", False)))
    c.max_episode_length = int(np.ceil(cfg.env.episode_length_s / dt))     # LR:1261
    c.episode_length_s = cfg.env.episode_length_s
    c.send_timeouts = int(bool(cfg.env.send_timeouts))

    # --- rewards
    scales = cfg.rewards.scales.to_dict() if hasattr(cfg.rewards.scales, "to_dict") else dict(vars(cfg.rewards.scales))
    for name, scale in scales.items():
        if scale == 0:
            continue                                                        # LR:1043-1044
        if name not in abi.REWARD_IDS:
            raise AttributeError(f"'LeggedRobot' object has no attribute '_reward_{name}'")  # what LR:1055 raises
        c.reward_scales[abi.REWARD_IDS[name]] = scale * dt                 # LR:1046
    r = cfg.rewards
    c.only_positive_rewards = int(bool(r.only_positive_rewards))
    c.tracking_sigma = r.tracking_sigma
    c.soft_dof_pos_limit, c.soft_dof_vel_limit, c.soft_torque_limit = r.soft_dof_pos_limit, r.soft_dof_vel_limit, r.soft_torque_limit
    c.base_height_target = r.base_height_target
    c.max_contact_force = r.max_contact_force
    c.foot_height_target_base = _get(r, "foot_height_target_base", 0.0)
    c.foot_height_target_terrain = _get(r, "foot_height_target_terrain", 0.0)
    props = list(t.terrain_proportions)

    def idx(k):
        return math.ceil(N * sum(props[:k]))                                # LR:72-90
    c.stairsup_start_idx, c.stairsup_end_idx = idx(4), idx(5)
    c.pit_start_idx, c.gap_end_idx = idx(8), N

    # --- observations
    os_ = cfg.normalization.obs_scales
    c.obs_scale_lin_vel, c.obs_scale_ang_vel = os_.lin_vel, os_.ang_vel
    c.obs_scale_dof_pos, c.obs_scale_dof_vel, c.obs_scale_height = os_.dof_pos, os_.dof_vel, os_.height_measurements
    c.add_noise = int(bool(cfg.noise.add_noise))
    ns, nl = cfg.noise.noise_scales, cfg.noise.noise_level                 # LR:898-908
    c.noise_vec_ang_vel = ns.ang_vel * nl * os_.ang_vel
    c.noise_vec_gravity = ns.gravity * nl
    c.noise_vec_dof_pos = ns.dof_pos * nl * os_.dof_pos
    c.noise_vec_dof_vel = ns.dof_vel * nl * os_.dof_vel
    c.noise_vec_height = ns.height_measurements * nl * os_.height_measurements

    # --- simulator
    for k in range(3):
        c.gravity[k] = cfg.sim.gravity[k]
    c.solver_iterations = SOLVER_DEFAULTS["solver_iterations"]
    physx = cfg.sim.physx
    c.solver_type = int(_get(physx, "solver_type", 1))                     # LRC:245 (0: pgs, 1: tgs)
    c.num_position_iterations = int(_get(physx, "num_position_iterations", 4))   # LRC:246
    override = os.environ.get("LSIM_SOLVER", "").lower()
    if override:
        c.solver_type = SOLVER_TYPES[override]
    if c.solver_type not in (0, 1):
        raise ValueError(f"cfg.sim.physx.solver_type must be 0 (pgs) or 1 (tgs), got {c.solver_type}")
    if c.solver_type == 1 and not 1 <= c.num_position_iterations <= abi.DEFINES["LSIM_MAX_POSITION_ITERATIONS"]:
        raise ValueError(f"cfg.sim.physx.num_position_iterations must be in 1..{abi.DEFINES['LSIM_MAX_POSITION_ITERATIONS']}, "
                         f"got {c.num_position_iterations}")
    # solver_type 1: one velocity-level pass over the joint-limit rows alone after the last position iteration (include/lsim.h); 0 = round 4
    c.tgs_limit_passes = int(os.environ.get("LSIM_TGS_LIMIT_PASSES", _get(physx, "tgs_limit_passes", 1)))
    if not 0 <= c.tgs_limit_passes <= abi.DEFINES["LSIM_MAX_POSITION_ITERATIONS"]:
        raise ValueError(f"tgs_limit_passes must be in 0..{abi.DEFINES['LSIM_MAX_POSITION_ITERATIONS']}, got {c.tgs_limit_passes}")
    if int(_get(physx, "num_velocity_iterations", 0)) != 0:
        raise ValueError("cfg.sim.physx.num_velocity_iterations != 0 is not modelled (every reference config sets 0, LRC:247)")
    c.contact_offset = cfg.sim.physx.contact_offset
    c.max_depenetration_velocity = cfg.sim.physx.max_depenetration_velocity
    c.erp = SOLVER_DEFAULTS["erp"]
    c.contact_slop = SOLVER_DEFAULTS["contact_slop"]
    c.using_amp = int(bool(using_amp))
    c.max_linear_velocity = cfg.asset.max_linear_velocity
    c.max_angular_velocity = cfg.asset.max_angular_velocity
    # linear velocities of the state tensors: the body's centre of mass, as PhysX reports them (include/lsim.h lin_vel_at_com); the link-origin
    # convention of rounds 1-4 stays reachable for A/B runs and old checkpoints' evaluation (cfg.sim.physx.lin_vel_at_com = 0 / LSIM_LIN_VEL=origin)
    c.lin_vel_at_com = int(bool(_get(physx, "lin_vel_at_com", 1)))
    lv = os.environ.get("LSIM_LIN_VEL", "").lower()
    if lv:
        if lv not in ("com", "origin"):
            raise ValueError(f"LSIM_LIN_VEL must be 'com' or 'origin', got {lv!r}")
        c.lin_vel_at_com = int(lv == "com")
    return c
