"""Environment / training configuration for the Aliengo tasks.

Mirrors the *interface* of the reference's nested config classes (attribute access such as
`cfg.env.num_envs`, `cfg.rewards.scales.tracking_lin_vel`, `class_to_dict`-style conversion) so
that code written against legged_gym configs reads the same, but is built from plain nested
dictionaries: a task config is `defaults` deep-merged with the task's overrides.

Values restate: LeggedRobotCfg / LeggedRobotCfgPPO (envs/base/legged_robot_config.py:48-298),
AlienGoRoughCfg(PPO) (envs/aliengo/aliengo_config.py:33-337), AlienGoStairsCfg(PPO)
(envs/aliengo/aliengo_stairs_config.py:40-238) and the AMP variant
(envs/aliengo/aliengo_amp_config.py:41-347).  tests/test_config.py checks every leaf against
tests/golden/ref_cfg_*.json (captured from the reference with helpers.class_to_dict, HLP:45).
"""
import copy
import math


class ConfigNode:
    """Attribute-style view of a nested dict (leaf lists/dicts are kept as they are)."""

    def __init__(self, d=None, _raw_dict_keys=()):
        for k, v in (d or {}).items():
            if isinstance(v, dict) and k not in _raw_dict_keys and not v.get("__leaf__", False):
                v = ConfigNode(v, _raw_dict_keys)
            elif isinstance(v, dict) and v.get("__leaf__", False):
                v = {kk: vv for kk, vv in v.items() if kk != "__leaf__"}
            setattr(self, k, v)

    def to_dict(self):
        out = {}
        for k in sorted(vars(self)):  # alphabetical, like dir() in helpers.class_to_dict (HLP:49)
            v = getattr(self, k)
            out[k] = v.to_dict() if isinstance(v, ConfigNode) else copy.deepcopy(v)
        return out

    def __contains__(self, k):
        return hasattr(self, k)

    def __repr__(self):
        return f"ConfigNode({self.to_dict()})"


def class_to_dict(obj):
    """Same contract as legged_gym.utils.helpers.class_to_dict (HLP:45-60) for ConfigNode trees."""
    return obj.to_dict() if isinstance(obj, ConfigNode) else obj


def deep_merge(base, over, replace=()):
    """Recursive dict merge; keys named in `replace` are replaced wholesale (classes the reference
    re-declares without inheriting, e.g. `class domain_rand:` in AGC:148)."""
    out = copy.deepcopy(base)
    for k, v in over.items():
        if v is None and isinstance(out.get(k), dict):
            del out[k]   # a class the derived config does not have
            continue
        if k in replace or not (isinstance(v, dict) and isinstance(out.get(k), dict)) or v.get("__leaf__") or out[k].get("__leaf__"):
            out[k] = copy.deepcopy(v)
        else:
            out[k] = deep_merge(out[k], v, replace)
    return out


def _leaf(**kw):
    kw["__leaf__"] = True
    return kw


# ----------------------------------------------------------------------------- base defaults (LRC:48-255)
LEGGED_ROBOT_DEFAULTS = {
    "env": dict(num_envs=4096, num_one_step_observations=45, num_observations=270, num_one_step_privileged_obs=238,
                num_privileged_obs=238, num_actions=12, env_spacing=3.0, send_timeouts=True, episode_length_s=20,
                reference_state_initialization=False),
    "terrain": dict(mesh_type="trimesh", horizontal_scale=0.1, vertical_scale=0.005, border_size=25, curriculum=True,
                    static_friction=1.0, dynamic_friction=1.0, restitution=0.0, measure_heights=True,
                    measured_points_x=[-0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2, -0.1, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8],
                    measured_points_y=[-0.5, -0.4, -0.3, -0.2, -0.1, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5],
                    selected=False, terrain_kwargs=None, max_init_terrain_level=5, terrain_length=8.0, terrain_width=8.0,
                    num_rows=10, num_cols=20, terrain_proportions=[0.1, 0.2, 0.3, 0.3, 0.1], slope_treshold=0.75),
    "commands": dict(curriculum=True, max_forward_curriculum=2.0, num_commands=4, resampling_time=10.0, heading_command=True,
                     ranges=dict(lin_vel_x=[-1.0, 1.0], lin_vel_y=[-1.0, 1.0], ang_vel_yaw=[-3.14, 3.14], heading=[-3.14, 3.14])),
    "init_state": dict(pos=[0.0, 0.0, 1.0], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0],
                       default_joint_angles=_leaf(joint_a=0.0, joint_b=0.0)),
    "control": dict(control_type="P", stiffness=_leaf(joint_a=10.0, joint_b=15.0), damping=_leaf(joint_a=1.0, joint_b=1.5),
                    action_scale=0.5, decimation=4, hip_reduction=1.0),
    "asset": dict(file="", name="legged_robot", foot_name="None", penalize_contacts_on=[], terminate_after_contacts_on=[],
                  disable_gravity=False, collapse_fixed_joints=True, fix_base_link=False, default_dof_drive_mode=3,
                  self_collisions=0, replace_cylinder_with_capsule=True, flip_visual_attachments=True, density=0.001,
                  angular_damping=0.0, linear_damping=0.0, max_angular_velocity=1000.0, max_linear_velocity=1000.0,
                  armature=0.0, thickness=0.01),
    "domain_rand": dict(randomize_payload_mass=True, payload_mass_range=[-1, 2], randomize_com_displacement=True,
                        com_displacement_range=[-0.05, 0.05], randomize_link_mass=False, link_mass_range=[0.9, 1.1],
                        randomize_friction=True, friction_range=[0.2, 1.25], randomize_restitution=False,
                        restitution_range=[0.0, 1.0], randomize_motor_strength=True, motor_strength_range=[0.9, 1.1],
                        randomize_kp=True, kp_range=[0.9, 1.1], randomize_kd=True, kd_range=[0.9, 1.1],
                        randomize_initial_joint_pos=True, initial_joint_pos_range=[0.5, 1.5], disturbance=True,
                        disturbance_range=[-30.0, 30.0], disturbance_interval=8, push_robots=True, push_interval_s=16,
                        max_push_vel_xy=1.0, delay=True),
    "rewards": dict(reward_curriculum=False, reward_curriculum_term=["lin_vel_z"], reward_curriculum_schedule=[0, 1000, 1, 0],
                    scales=dict(termination=-0.0, tracking_lin_vel=1.0, tracking_ang_vel=0.5, lin_vel_z=-2.0, ang_vel_xy=-0.05,
                                orientation=-0.0, torques=-0.00001, dof_vel=-0.0, dof_acc=-2.5e-7, base_height=-0.0,
                                feet_air_time=1.0, collision=-1.0, feet_stumble=-0.0, action_rate=-0.01, stand_still=-0.0),
                    only_positive_rewards=True, tracking_sigma=0.25, soft_dof_pos_limit=1.0, soft_dof_vel_limit=1.0,
                    soft_torque_limit=1.0, base_height_target=1.0, max_contact_force=100.0, clearance_height_target=0.09),
    "normalization": dict(obs_scales=dict(lin_vel=2.0, ang_vel=0.25, dof_pos=1.0, dof_vel=0.05, height_measurements=5.0),
                          clip_observations=100.0, clip_actions=100.0),
    "noise": dict(add_noise=True, noise_level=1.0,
                  noise_scales=dict(dof_pos=0.01, dof_vel=1.5, lin_vel=0.1, ang_vel=0.2, gravity=0.05, height_measurements=0.1)),
    "viewer": dict(ref_env=0, pos=[10, 15, 6], lookat=[11.0, 25, 3.0]),
    "sim": dict(dt=0.005, substeps=1, gravity=[0.0, 0.0, -9.81], up_axis=1,
                physx=dict(num_threads=10, solver_type=1, num_position_iterations=4, num_velocity_iterations=0,
                           contact_offset=0.01, rest_offset=0.0, bounce_threshold_velocity=0.5,
                           max_depenetration_velocity=1.0, max_gpu_contact_pairs=2 ** 23,
                           default_buffer_size_multiplier=5, contact_collection=2)),
}

LEGGED_ROBOT_PPO_DEFAULTS = {
    "seed": 1,
    "runner_class_name": "HIMOnPolicyRunner",
    "policy": dict(init_noise_std=1.0, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu"),
    "algorithm": dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01,
                      num_learning_epochs=5, num_mini_batches=4, learning_rate=1.0e-3, schedule="adaptive", gamma=0.99,
                      lam=0.95, desired_kl=0.01, max_grad_norm=1.0),
    "runner": dict(policy_class_name="HIMActorCritic", algorithm_class_name="HIMPPO", num_steps_per_env=100,
                   max_iterations=200000, save_interval=20, experiment_name="test", run_name="", resume=False,
                   load_run=-1, checkpoint=-1, resume_path=None),
}

# ----------------------------------------------------------------------------- Aliengo "flat" (AGC:33-291)
_ALIENGO_DOMAIN_RAND = dict(
    randomize_payload_mass=True, payload_mass_range=[0.0, 3.0], randomize_com_displacement=True,
    com_displacement_range=[-0.05, 0.05], randomize_link_mass=False, link_mass_range=[0.9, 1.1],
    randomize_friction=True, friction_range=[0.2, 1.25], randomize_restitution=False, restitution_range=[0.0, 1.0],
    randomize_motor_strength=True, motor_strength_range=[0.9, 1.1], randomize_kp=True, kp_range=[0.9, 1.1],
    randomize_kd=True, kd_range=[0.9, 1.1],
    base_init_pos_range=_leaf(x=[-1.0, 1.0], y=[-1.0, 1.0], z=[0.0, 0.05]),
    base_init_rot_range=_leaf(roll=[-0.2, 0.2], pitch=[-0.2, 0.2], yaw=[-0.0, 0.0]),
    base_init_vel_range=_leaf(x=[-0.5, 0.5], y=[-0.5, 0.5], z=[-0.5, 0.5], roll=[-0.5, 0.5], pitch=[-0.5, 0.5], yaw=[-0.5, 0.5]),
    dof_init_pos_ratio_range=[0.5, 1.5], randomize_dof_vel=True, dof_init_vel_range=[-0.1, 0.1],
    disturbance=True, disturbance_range=[-30.0, 30.0], disturbance_interval=8,
    push_robots=True, push_interval_s=16, max_push_vel_xy=1.0, delay=True, recover_mode=False)

_ALIENGO_SCALES = dict(
    termination=-0.0, tracking_lin_vel=1.5, tracking_ang_vel=1.5, lin_vel_z=-2.0, ang_vel_xy=-0.05, orientation=-2.0,
    base_height=-8.0, torques=-0.0002, torque_limits=-0.0, dof_vel=-0.0, dof_acc=-2.5e-7, stand_still=-0.1, hip_pos=-0.2,
    thigh_pose=-0.05, calf_pose=-0.05, dof_pos_limits=-0.0, dof_vel_limits=-0.0, joint_power=-2e-5, feet_mirror=-0.05,
    action_rate=-0.02, smoothness=-0.01, hip_action_magnitude=-0.0, collision=-0.0, feet_contact_forces=-0.00015,
    feet_air_time=0.25, has_contact=0.0, feet_stumble=-0.0, feet_slide=-0.01, foot_clearance_base=-0.1,
    foot_clearance_base_terrain=-0.0, stuck=-0.01, upward=0.0)

ALIENGO_OVERRIDES = {
    "init_state": dict(pos=[0.0, 0.0, 0.50], default_joint_angles=_leaf(
        FL_hip_joint=0.0, RL_hip_joint=0.0, FR_hip_joint=-0.0, RR_hip_joint=-0.0,
        FL_thigh_joint=0.8, RL_thigh_joint=0.8, FR_thigh_joint=0.8, RR_thigh_joint=0.8,
        FL_calf_joint=-1.5, RL_calf_joint=-1.5, FR_calf_joint=-1.5, RR_calf_joint=-1.5)),
    "terrain": dict(border_size=15, terrain_proportions=[0.3, 0.3, 0.2, 0.2]),
    "control": dict(stiffness=_leaf(joint=40.0), damping=_leaf(joint=2.0)),
    "commands": dict(max_forward_curriculum=1.5, max_backward_curriculum=1.0, max_lat_curriculum=1.0,
                     ranges=dict(lin_vel_x=[-1.0, 1.0], lin_vel_y=[-0.5, 0.5], ang_vel_yaw=[-1.0, 1.0], heading=[-math.pi, math.pi])),
    "asset": dict(file="{LEGGED_GYM_ROOT_DIR}/resources/robots/aliengo/urdf/aliengo.urdf", name="aliengo", foot_name="foot",
                  penalize_contacts_on=["thigh", "calf", "base"], terminate_after_contacts_on=["base"],
                  privileged_contacts_on=["base", "thigh", "calf"], self_collisions=1),
    "termination": dict(base_vel_violate_commands=False, out_of_border=True, fall_down=True),
    "domain_rand": _ALIENGO_DOMAIN_RAND,
    "rewards": dict(scales=_ALIENGO_SCALES, reward_curriculum=False, reward_curriculum_term=["feet_edge"],
                    reward_curriculum_schedule=[[4000, 10000, 0.1, 1.0]], only_positive_rewards=False, tracking_sigma=0.25,
                    soft_dof_pos_limit=0.95, soft_dof_vel_limit=0.95, soft_torque_limit=0.95, base_height_target=0.43,
                    foot_height_target_base=-0.27, foot_height_target_terrain=0.15, max_contact_force=100.0),
}
_REPLACED = ("domain_rand", "scales", "normalization", "noise", "termination")  # re-declared without inheritance (AGC:141,148,217,272,282)

ALIENGO_PPO_OVERRIDES = {
    "runner": dict(max_iterations=1000, save_interval=100, experiment_name="flat_aliengo"),
}

# ----------------------------------------------------------------------------- Aliengo stairs (AGS:40-238)
_STAIRS_SCALES = dict(
    termination=-50.0, tracking_lin_vel=1.5, tracking_ang_vel=0.75, lin_vel_z=-2.0, ang_vel_xy=-0.05, orientation=-0.2,
    base_height=-5.0, torques=-0.0002, torque_limits=-0.0, dof_vel=-0.0, dof_acc=-2.5e-7, stand_still=-0.01, hip_pos=-0.2,
    thigh_pose=-0.1, calf_pose=-0.1, dof_pos_limits=-0.0, dof_vel_limits=-0.0, joint_power=-6e-5, feet_mirror=-0.0,
    action_rate=-0.01, smoothness=-0.0, hip_action_magnitude=-0.0, collision=-3.0, feet_contact_forces=-0.00015,
    feet_air_time=0.1, has_contact=0.0, feet_stumble=-1.0, feet_slide=-0.01, foot_clearance_base=-0.0,
    foot_clearance_base_terrain=-0.0, stuck=-1.0, upward=0.0)

ALIENGO_STAIRS_OVERRIDES = {
    "terrain": dict(terrain_length=10.0, terrain_width=10.0, terrain_proportions=[0.0, 0.0, 0.1, 0.1, 0.3, 0.3, 0.2, 0.0, 0.0, 0.0]),
    "termination": dict(base_vel_violate_commands=True, out_of_border=True, fall_down=True),
    "rewards": dict(scales=_STAIRS_SCALES),
}
ALIENGO_STAIRS_PPO_OVERRIDES = {
    "runner": dict(max_iterations=4000, save_interval=200, experiment_name="stairs_aliengo", resume=True),
}

# ----------------------------------------------------------------------------- Aliengo AMP (AGA:41-347): differences only
AMP_MOTION_CLIPS = ("trot0", "trot1", "trot2", "left_turn0", "left_turn1", "right_turn0", "right_turn1")  # AGA:34-36 globs
ALIENGO_AMP_OVERRIDES = {
    "commands": dict(max_forward_curriculum=2.0),
    "domain_rand": dict(_ALIENGO_DOMAIN_RAND, dof_init_vel_range=[-1.0, 1.0]),
    "rewards": dict(scales=dict(_ALIENGO_SCALES, base_height=-10.0)),
    "termination": None,   # the AMP config declares no `termination` class: LR:266-282 then skip those checks
}
ALIENGO_AMP_PPO_OVERRIDES = {
    "runner_class_name": "HybridPolicyRunner",
    "algorithm": dict(amp_replay_buffer_size=1000000),
    "runner": dict(algorithm_class_name="HybridPPO", amp_reward_coef=0.01, amp_num_preload_transitions=2000000,
                   amp_task_reward_lerp=0.3, amp_discr_hidden_dims=[1024, 512], min_normalized_std=[0.05, 0.02, 0.05] * 4,
                   amp_motion_files=[f"mocap_motions_aliengo/{c}.txt" for c in AMP_MOTION_CLIPS]),
}


# ----------------------------------------------------------------------------- Aliengo recovery (aliengo_recover_config.py:38-190): differences only
# robots start in any orientation (LR:786-794 draws roll / pitch / yaw from +-pi), nothing terminates on contact, the `_up` reward variants
# (LR:1459-1770: the term times the upright-ness of the base) replace the plain ones.  The reference also switches PhysX self-collisions
# ON for this task (`self_collisions = 0`); this build has no robot-robot or link-link contact: noted, not modelled.
_RECOVER_SCALES = dict(
    termination=-0.0, tracking_lin_vel=2.0, tracking_ang_vel=1.0, lin_vel_z_up=-2.0, ang_vel_xy_up=-0.05, orientation_up=-2.0,
    base_height_up=-5.0, torques=-0.0002, torque_limits=-0.0, dof_vel=-0.0, dof_acc=-2.5e-7, stand_nice=-0.1, hip_pos_up=-0.3,
    thigh_pose_up=-0.05, calf_pose_up=-0.05, dof_pos_limits=-0.0, dof_vel_limits=-0.0, joint_power=-2e-5, feet_mirror_up=-0.05,
    action_rate=-0.02, smoothness=-0.01, hip_action_magnitude=-0.01, collision_up=-0.0, feet_contact_forces=-0.00015,
    feet_air_time=0.25, has_contact=0.3, feet_stumble_up=-0.0, feet_slide_up=-0.01, foot_clearance_base_up=-0.1,
    foot_clearance_base_terrain=-0.0, stuck=-0.05, upward=1.0)
ALIENGO_RECOVER_OVERRIDES = {
    "commands": dict(max_forward_curriculum=2.0, heading_command=False),
    "asset": dict(terminate_after_contacts_on=[], self_collisions=0),
    "terrain": dict(terrain_proportions=[0.5, 0.5]),
    "domain_rand": dict(_ALIENGO_DOMAIN_RAND, recover_mode=True,
                        base_init_rot_range=_leaf(roll=[-3.14, 3.14], pitch=[-3.14, 3.14], yaw=[-3.14, 3.14])),
    "rewards": dict(only_positive_rewards=True, scales=_RECOVER_SCALES),
}
ALIENGO_RECOVER_PPO_OVERRIDES = {
    "runner": dict(max_iterations=2000, experiment_name="recover_aliengo", resume=True),
}


def _build(*layers):
    d = {}
    for layer in layers:
        d = deep_merge(d, layer, replace=_REPLACED)
    return d


def aliengo_cfg():
    """AlienGoRoughCfg (task "aliengo", envs/__init__.py:53)."""
    return ConfigNode(_build(LEGGED_ROBOT_DEFAULTS, ALIENGO_OVERRIDES))


def aliengo_cfg_ppo():
    return ConfigNode(_build(LEGGED_ROBOT_PPO_DEFAULTS, ALIENGO_PPO_OVERRIDES))


def aliengo_stairs_cfg():
    """AlienGoStairsCfg (task "aliengo_stairs")."""
    return ConfigNode(_build(LEGGED_ROBOT_DEFAULTS, ALIENGO_OVERRIDES, ALIENGO_STAIRS_OVERRIDES))


def aliengo_stairs_cfg_ppo():
    return ConfigNode(_build(LEGGED_ROBOT_PPO_DEFAULTS, ALIENGO_PPO_OVERRIDES, ALIENGO_STAIRS_PPO_OVERRIDES))


def aliengo_amp_cfg():
    """aliengo_amp_config.AlienGoRoughCfg (selected when USING_AMP, envs/__init__.py:39-40)."""
    return ConfigNode(_build(LEGGED_ROBOT_DEFAULTS, ALIENGO_OVERRIDES, ALIENGO_AMP_OVERRIDES))


def aliengo_amp_cfg_ppo():
    return ConfigNode(_build(LEGGED_ROBOT_PPO_DEFAULTS, ALIENGO_PPO_OVERRIDES, ALIENGO_AMP_PPO_OVERRIDES))


def aliengo_recover_cfg():
    """AlienGoRoughRecoverCfg (task "aliengo_recover", envs/__init__.py:55)."""
    return ConfigNode(_build(LEGGED_ROBOT_DEFAULTS, ALIENGO_OVERRIDES, ALIENGO_RECOVER_OVERRIDES))


def aliengo_recover_cfg_ppo():
    return ConfigNode(_build(LEGGED_ROBOT_PPO_DEFAULTS, ALIENGO_PPO_OVERRIDES, ALIENGO_RECOVER_PPO_OVERRIDES))


# ----------------------------------------------------------------------------- Go1 on the Aliengo task (BASELINE config 5's second robot)
# The reference registers a "go1" task (envs/__init__.py:52) whose config predates its own LeggedRobot (it lacks the
# dof_init_pos_ratio_range / base_init_*_range / termination keys LR:699-812 read) and does not run.  This task is therefore the
# Aliengo task -- same observations, actions, reward set, curricula -- with the robot-specific values of the reference's
# Go1RoughCfg (go1_config.py: init_state, control, asset, base_height_target) and the Go1 model table (robots/tables/go1.json).
# NOT reference-comparable.
GO1_OVERRIDES = {
    "init_state": dict(pos=[0.0, 0.0, 0.42], default_joint_angles=_leaf(
        FL_hip_joint=0.1, RL_hip_joint=0.1, FR_hip_joint=-0.1, RR_hip_joint=-0.1,
        FL_thigh_joint=0.8, RL_thigh_joint=1.0, FR_thigh_joint=0.8, RR_thigh_joint=1.0,
        FL_calf_joint=-1.5, RL_calf_joint=-1.5, FR_calf_joint=-1.5, RR_calf_joint=-1.5)),
    "control": dict(stiffness=_leaf(joint=40.0), damping=_leaf(joint=1.0), action_scale=0.25),
    "asset": dict(file="{LEGGED_GYM_ROOT_DIR}/resources/robots/go1/urdf/go1.urdf", name="go1", flip_visual_attachments=False),
    "rewards": dict(base_height_target=0.3, foot_height_target_base=-0.2),
}


def go1_cfg():
    return ConfigNode(_build(LEGGED_ROBOT_DEFAULTS, ALIENGO_OVERRIDES, GO1_OVERRIDES))


def go1_cfg_ppo():
    return ConfigNode(_build(LEGGED_ROBOT_PPO_DEFAULTS, ALIENGO_PPO_OVERRIDES, {"runner": dict(experiment_name="flat_go1")}))


# ----------------------------------------------------------------------------- Go2 on the Aliengo task (BASELINE config 5 names Go2)
# The reference has no Go2 asset or task; it does ship Go2 mocap clips (datasets/mocap_motions_go2), to which the kinematics of
# robots/tables/go2.json are fitted (tools/gen_go2_table.py; inertial values and limits: Unitree's published description, nominal).  Task
# values: Unitree's own legged_gym settings for the robot (default pose, Kp 20 / Kd 0.5, action scale 0.25).  NOT reference-comparable.
GO2_OVERRIDES = {
    "init_state": dict(pos=[0.0, 0.0, 0.42], default_joint_angles=_leaf(
        FL_hip_joint=0.1, RL_hip_joint=0.1, FR_hip_joint=-0.1, RR_hip_joint=-0.1,
        FL_thigh_joint=0.8, RL_thigh_joint=1.0, FR_thigh_joint=0.8, RR_thigh_joint=1.0,
        FL_calf_joint=-1.5, RL_calf_joint=-1.5, FR_calf_joint=-1.5, RR_calf_joint=-1.5)),
    "control": dict(stiffness=_leaf(joint=20.0), damping=_leaf(joint=0.5), action_scale=0.25),
    "asset": dict(file="{LEGGED_GYM_ROOT_DIR}/resources/robots/go2/urdf/go2.urdf", name="go2", flip_visual_attachments=False),
    "rewards": dict(base_height_target=0.3, foot_height_target_base=-0.2),
}


def go2_cfg():
    return ConfigNode(_build(LEGGED_ROBOT_DEFAULTS, ALIENGO_OVERRIDES, GO2_OVERRIDES))


def go2_cfg_ppo():
    return ConfigNode(_build(LEGGED_ROBOT_PPO_DEFAULTS, ALIENGO_PPO_OVERRIDES, {"runner": dict(experiment_name="flat_go2")}))


TASKS = {
    "go1": (go1_cfg, go1_cfg_ppo),
    "go2": (go2_cfg, go2_cfg_ppo),
    "aliengo": (aliengo_cfg, aliengo_cfg_ppo),
    "aliengo_stairs": (aliengo_stairs_cfg, aliengo_stairs_cfg_ppo),
    "aliengo_amp": (aliengo_amp_cfg, aliengo_amp_cfg_ppo),
    "aliengo_recover": (aliengo_recover_cfg, aliengo_recover_cfg_ppo),
}
