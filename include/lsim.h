/*
 * lsim.h -- C-ABI of the MI355X-native vectorised legged-robot simulator ("leggedsim").
 *
 * Drop-in boundary for the hot path of xyyandhtl/IsaacgymLoco: everything that
 * LeggedRobot.step() (legged_gym/envs/base/legged_robot.py:122-176) does between
 * receiving `actions` and returning the observation tuple, i.e. the Isaac Gym tensor
 * API calls listed in SURVEY.md 2.3 plus the torch post-physics stack.
 *
 * The reference has no FFI of its own on this path (it calls the closed-source
 * `isaacgym.gymapi` pybind module); each entry point below names the reference
 * call(s) it replaces.  Plain pointers and sizes only -- no torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative LSIM_E_* code; the text of the
 *     last error of a handle is available from lsim_last_error().  No exceptions cross
 *     the ABI.  One caller thread per handle.
 *   - all device work is enqueued on the caller-supplied HIP stream (pass
 *     torch.cuda.current_stream().cuda_stream); no call synchronises the host unless
 *     documented.
 *   - device memory: the caller may pass one pre-allocated arena (lsim_query_arena gives
 *     the size) so that a PyTorch host can alias every buffer zero-copy the way the
 *     reference aliases simulator state with gymtorch.wrap_tensor (LR:930-944); with
 *     arena == NULL the library allocates (hipMalloc) and owns it.
 *   - quaternions are xyzw; root/body velocities are world-frame (LR:929-941).
 */
#ifndef LSIM_H
#define LSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSIM_ABI_VERSION 6   /* 2: LSIM_BUF_CONTACT_COUNT, fixed-point words in LSIM_BUF_STATS (round 2); 3: LSIM_BUF_SUBSTEP_TORQUES (round 3);
                                 4: lsim_config.solver_type / num_position_iterations out of the reserved words (round 4);
                                 5: lsim_config.lin_vel_at_com (centre-of-mass linear velocities, the PhysX convention) and tgs_limit_passes, lsim_get / set_reset_calls (round 5);
                                 6: LSIM_BUF_NONFINITE + LSIM_STATS_NONFINITE (robots whose simulated state is not finite), lsim_amp_step and the discriminator-update kernels (round 6) */

/* ---- fixed sizes of the robot family on this path (12-DoF quadrupeds) ---- */
#define LSIM_NUM_DOF 12
#define LSIM_NUM_BODIES 17      /* base + 4 x (hip, thigh, calf, foot); LR:1143 */
#define LSIM_NUM_LEGS 4
#define LSIM_NUM_ACTIONS 12
#define LSIM_ONE_STEP_OBS 45    /* LRC:51 */
#define LSIM_OBS_HISTORY 6      /* LRC:52 */
#define LSIM_NUM_OBS 270
#define LSIM_MAX_HEIGHT_PTS_X 32
#define LSIM_MAX_HEIGHT_PTS_Y 32
#define LSIM_NUM_HEIGHT_PTS 187 /* 17 x 11, AGC:79-80; the obs layout hard-codes 187 (LR:895) */
#define LSIM_NUM_PRIV_OBS 238   /* 45 + 3 + 3 + 187, LRC:53 */
#define LSIM_NUM_AMP_OBS 30     /* LR:416 */
#define LSIM_NUM_BASE_HEIGHT_PTS 63 /* 7 x 9, LR:1308-1312 */
#ifndef LSIM_MAX_COLLISION_POINTS /* the CPU oracle can be compiled with a larger table: densely sampled TRUE collision shapes (tests/test_shape_variants.py) */
#define LSIM_MAX_COLLISION_POINTS 64
#endif
#ifndef LSIM_MAX_CONTACTS      /* the CPU oracle can be compiled with a larger cap to measure what the cap costs (tests/test_contact_cap.py) */
#define LSIM_MAX_CONTACTS 8
#endif
#define LSIM_MAX_POSITION_ITERATIONS 12   /* solver_type 1: sub-iterations per sim_dt (the reference sets 4, LRC:246) */
#define LSIM_SOLVER_PGS 0
#define LSIM_SOLVER_TGS 1
#define LSIM_TERRAIN_LEVELS_MAX 32
#define LSIM_TERRAIN_TYPES_MAX 32

/* error codes */
#define LSIM_OK 0
#define LSIM_E_INVALID (-1)
#define LSIM_E_NOMEM (-2)
#define LSIM_E_HIP (-3)
#define LSIM_E_UNSUPPORTED (-4)
#define LSIM_E_ABI (-5)

/* ---- reward terms: every `_reward_<name>` defined in LR:1444-1770, in the
 * alphabetical order in which the reference evaluates and accumulates them
 * (class_to_dict iterates dir(), HLP:49 -> LR:1050-1055 -> LR:369-373). ---- */
enum lsim_reward_id {
    LSIM_R_ACTION_RATE = 0,
    LSIM_R_ANG_VEL_XY,
    LSIM_R_ANG_VEL_XY_UP,
    LSIM_R_BASE_HEIGHT,
    LSIM_R_BASE_HEIGHT_UP,
    LSIM_R_CALF_POSE,
    LSIM_R_CALF_POSE_UP,
    LSIM_R_COLLISION,
    LSIM_R_COLLISION_UP,
    LSIM_R_DOF_ACC,
    LSIM_R_DOF_POS_DIF,
    LSIM_R_DOF_POS_LIMITS,
    LSIM_R_DOF_VEL,
    LSIM_R_DOF_VEL_LIMITS,
    LSIM_R_FEET_AIR_TIME,
    LSIM_R_FEET_CONTACT_FORCES,
    LSIM_R_FEET_MIRROR,
    LSIM_R_FEET_MIRROR_UP,
    LSIM_R_FEET_SLIDE,
    LSIM_R_FEET_SLIDE_UP,
    LSIM_R_FEET_STUMBLE,
    LSIM_R_FEET_STUMBLE_UP,
    LSIM_R_FOOT_CLEARANCE_BASE,
    LSIM_R_FOOT_CLEARANCE_BASE_UP,
    LSIM_R_FOOT_CLEARANCE_TERRAIN,
    LSIM_R_FOOT_CLEARANCE_TERRAIN_UP,
    LSIM_R_HAS_CONTACT,
    LSIM_R_HIP_ACTION_MAGNITUDE,
    LSIM_R_HIP_POS,
    LSIM_R_HIP_POS_UP,
    LSIM_R_JOINT_POWER,
    LSIM_R_LIN_VEL_Z,
    LSIM_R_LIN_VEL_Z_UP,
    LSIM_R_ORIENTATION,
    LSIM_R_ORIENTATION_UP,
    LSIM_R_POWER,
    LSIM_R_POWER_DISTRIBUTION,
    LSIM_R_SMOOTHNESS,
    LSIM_R_STAND_NICE,
    LSIM_R_STAND_STILL,
    LSIM_R_STUCK,
    LSIM_R_TERMINATION,
    LSIM_R_THIGH_POSE,
    LSIM_R_THIGH_POSE_UP,
    LSIM_R_TORQUE_LIMITS,
    LSIM_R_TORQUES,
    LSIM_R_TORQUES_DIF,
    LSIM_R_TORQUES_DISTRIBUTION,
    LSIM_R_TRACKING_ANG_VEL,
    LSIM_R_TRACKING_LIN_VEL,
    LSIM_R_UPWARD,
    LSIM_NUM_REWARD_TERMS
};

/* ---- counter-based RNG draw sites (Philox4x32-10, key = (seed, rank)) ----
 * counter = (env, common_step_counter, tag, idx >> 2), lane = idx & 3,
 * u = (x >> 8) * 2^-24 in [0,1).  One tag per torch draw site of the reference
 * (stream order listed in SURVEY.md 8a quirk 12). */
enum lsim_rng_tag {
    LSIM_RNG_DELAY = 1,        /* LR:134            idx 0                                  */
    LSIM_RNG_CMD = 2,          /* LR:641-651        idx 0 vx, 1 vy, 2 heading|yaw, 3 vx_hi */
    LSIM_RNG_PUSH = 3,         /* LR:827            idx 0..1                               */
    LSIM_RNG_DISTURB = 4,      /* LR:842            idx 0..2                               */
    LSIM_RNG_TERM_NOISE = 5,   /* LR:451, LR:457    idx 0..44 obs, 45..231 heights         */
    LSIM_RNG_RESET_LEVEL = 6,  /* LR:864            idx 0                                  */
    LSIM_RNG_RESET_DOF = 7,    /* LR:699, LR:709    idx 0..11 pos ratio, 12..23 vel        */
    LSIM_RNG_RESET_ROOT = 8,   /* LR:730-812        idx 0..2 xyz, 3..5 rpy, 6..11 vel      */
    LSIM_RNG_RESET_CMD = 9,    /* LR:320 -> 641-651 same idx as LSIM_RNG_CMD               */
    LSIM_RNG_RESET_DR = 10,    /* LR:337-341, 535   idx 0 kp, 1 kd, 2 motor factor, 3 friction, 4 restitution */
    LSIM_RNG_OBS_NOISE = 11,   /* LR:394, LR:400    idx 0..44 obs, 45..231 heights         */
    LSIM_RNG_INIT = 12,        /* LR:999-1028, 1232 idx 0..11 motor_strength, 12 kp, 13 kd, 14 motor factor,
                                  15 payload, 16..18 com, 19 friction bucket id, 20 terrain level;
                                  step word = 0xFFFFFFFF                                    */
    LSIM_RNG_INIT_BUCKET = 13, /* LR:511            env word = bucket index, idx 0         */
    LSIM_RNG_POLICY = 14       /* HIMP:94 (actor_critic.act sample), lsim_rollout_act: step word = draw counter,
                                  block p gives the two Box-Muller pairs of actions 2p, 2p+1       */
};

/* ---- robot model: the URDF after Isaac Gym's fixed-joint collapse ----
 * body order: 0 base, then for leg l in (FL, FR, RL, RR): 1+4l hip, 2+4l thigh, 3+4l calf, 4+4l foot.
 * dof order : 3l + (0 hip, 1 thigh, 2 calf)  (LR:1145). */
typedef struct lsim_body {
    float mass;
    float com[3];        /* in the body (link) frame */
    float inertia[6];    /* about the com, body axes: xx, xy, xz, yy, yz, zz */
    float joint_pos[3];  /* origin of this body's frame in the parent frame (rpy is 0 for all joints) */
    float joint_axis[3]; /* revolute axis (unit), zero vector for base / fixed feet */
    int32_t parent;      /* -1 for base */
    int32_t dof;         /* -1 for base and for the fixed foot */
} lsim_body;

typedef struct lsim_collision_point {
    float pos[3];   /* sphere centre in the body frame */
    float radius;   /* 0 for box corners */
    int32_t body;   /* body index the contact force is reported on */
    int32_t pad;
} lsim_collision_point;

typedef struct lsim_robot_model {
    lsim_body bodies[LSIM_NUM_BODIES];
    float dof_pos_lower[LSIM_NUM_DOF];  /* hard URDF limits (rad) */
    float dof_pos_upper[LSIM_NUM_DOF];
    float dof_vel_limit[LSIM_NUM_DOF];
    float dof_effort_limit[LSIM_NUM_DOF];
    int32_t num_collision_points;       /* ordered by priority: overflow beyond LSIM_MAX_CONTACTS is dropped from the end */
    int32_t pad;
    lsim_collision_point points[LSIM_MAX_COLLISION_POINTS];
    int32_t feet_bodies[LSIM_NUM_LEGS];   /* LR:1209-1211 */
    uint32_t penalised_body_mask;         /* LR:1213-1215 */
    uint32_t termination_body_mask;       /* LR:1217-1219 */
} lsim_robot_model;

/* ---- flat configuration: the values of the reference's nested config classes that the
 * path reads (LeggedRobotCfg LRC:48 and the Aliengo subclasses AGC/AGS/AGA). ---- */
typedef struct lsim_config {
    int32_t abi_version;      /* LSIM_ABI_VERSION */
    int32_t num_envs;         /* LRC:50 */
    uint32_t seed;            /* LRC:258 */
    uint32_t rank;            /* second Philox key word: one stream per data-parallel rank */

    /* control (LRC:117-127, AGC:94-100) */
    float sim_dt;             /* 0.005, LRC:239 */
    int32_t decimation;       /* 4 */
    int32_t control_type;     /* 0 'P', 1 'V', 2 'T' (LR:676-687) */
    float action_scale;
    float hip_reduction;
    float p_gains[LSIM_NUM_DOF];
    float d_gains[LSIM_NUM_DOF];
    float torque_limits[LSIM_NUM_DOF];   /* LR:571 */
    float default_dof_pos[LSIM_NUM_DOF]; /* LR:980-996 */
    float clip_actions;       /* LRC:216 */
    float clip_observations;

    /* domain randomisation (AGC:148-214) */
    int32_t delay;
    int32_t randomize_kp;  float kp_range[2];
    int32_t randomize_kd;  float kd_range[2];
    int32_t randomize_motor_strength; float motor_strength_range[2];
    int32_t randomize_friction; float friction_range[2];
    int32_t randomize_restitution; float restitution_range[2];
    int32_t randomize_payload_mass; float payload_mass_range[2];
    int32_t randomize_com_displacement; float com_displacement_range[2];
    int32_t push_robots; int32_t push_interval; float max_push_vel_xy;        /* LR:627, LR:1263 */
    int32_t disturbance; int32_t disturbance_interval; float disturbance_range[2]; /* LR:631 */

    /* reset (LR:690-820) */
    int32_t has_dof_init_pos_ratio; float dof_init_pos_ratio_range[2];
    int32_t randomize_dof_vel; float dof_init_vel_range[2];  /* effective range read at LR:708 */
    int32_t has_base_init_pos_range; float base_init_pos_range[3][2];
    int32_t has_base_init_rot_range; float base_init_rot_range[3][2];
    float base_init_vel_range[6][2];
    float base_init_state[13];      /* LR:1160-1161 */

    /* commands (AGC:102-115) */
    float command_ranges[4][2];     /* lin_vel_x, lin_vel_y, ang_vel_yaw, heading */
    int32_t heading_command;
    int32_t resampling_steps;       /* int(resampling_time / dt), LR:612 */
    int32_t commands_curriculum;
    float max_forward_curriculum, max_backward_curriculum, max_lat_curriculum;

    /* terrain (AGC:67-91) */
    int32_t mesh_type;              /* 0 plane, 1 heightfield, 2 trimesh */
    float horizontal_scale, vertical_scale, border_size;
    int32_t grid_rows, grid_cols;   /* tot_rows, tot_cols, TER:59-60 */
    int32_t terrain_num_rows, terrain_num_cols;  /* levels, types */
    float terrain_length, terrain_width;         /* env_length, env_width */
    int32_t terrain_curriculum;
    int32_t max_init_terrain_level;
    int32_t measure_heights;
    int32_t num_points_x, num_points_y;          /* 17, 11 */
    float measured_points_x[LSIM_MAX_HEIGHT_PTS_X];
    float measured_points_y[LSIM_MAX_HEIGHT_PTS_Y];
    float slope_threshold;          /* trimesh vertical-wall correction, TER:72-75 */
    float terrain_friction, terrain_restitution;

    /* termination (AGC:141-146, LR:249-286) */
    int32_t term_base_vel_violate_commands, term_out_of_border, term_fall_down;
    int32_t max_episode_length;     /* ceil(episode_length_s / dt), LR:1261 */
    int32_t send_timeouts;

    /* rewards (AGC:216-270) */
    float reward_scales[LSIM_NUM_REWARD_TERMS];  /* already multiplied by dt (LR:1046); 0 = inactive */
    int32_t only_positive_rewards;
    float tracking_sigma, soft_dof_pos_limit, soft_dof_vel_limit, soft_torque_limit;
    float base_height_target, max_contact_force, foot_height_target_base, foot_height_target_terrain;
    int32_t stairsup_start_idx, stairsup_end_idx, pit_start_idx, gap_end_idx;   /* LR:79-90 */
    float episode_length_s;

    /* observations (AGC:272-291) */
    float obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_dof_pos, obs_scale_dof_vel, obs_scale_height;
    int32_t add_noise;
    /* entries of noise_scale_vec (LR:883-910) as the host computed them in double precision:
       ang_vel (obs 3:6), gravity (6:9), dof_pos (9:21), dof_vel (21:33), heights (45+6 .. +187); commands/actions are 0 */
    float noise_vec_ang_vel, noise_vec_gravity, noise_vec_dof_pos, noise_vec_dof_vel, noise_vec_height;

    /* simulator (LRC:238-255); the solver is the build's own (DESIGN.md "Physics") */
    float gravity[3];
    int32_t solver_iterations;      /* sweeps of the velocity-level Gauss-Seidel solver (solver_type 0) */
    float contact_offset, max_depenetration_velocity, erp, contact_slop;
    int32_t using_amp;              /* LRC:36: step() also produces terminal AMP states */
    float max_linear_velocity, max_angular_velocity;   /* asset options LRC:229-230 (1000 / 1000): PhysX clamps body velocities there */
    /* cfg.sim.physx.solver_type (LRC:245): 0 = PGS -- `solver_iterations` velocity-level sweeps over the whole sim_dt;
       1 = TGS (Temporal Gauss-Seidel, what every reference config sets) -- the sim_dt is split into `num_position_iterations`
       sub-iterations (LRC:246; 1..LSIM_MAX_POSITION_ITERATIONS), each relaxes every row once against the positional error reached
       so far.  num_velocity_iterations (LRC:247) is 0 in the reference and not modelled. */
    int32_t solver_type;
    int32_t num_position_iterations;
    /* what the LINEAR velocity columns of the state tensors mean (root_states[:, 7:10], rigid_body_states[:, :, 7:10]; LR:929-941):
       1 = the velocity of the body's CENTRE OF MASS, which is what PhysX's getLinearVelocity() / setLinearVelocity() read and write and
       therefore what the reference's base_lin_vel (LR:198-199), feet velocities (LR:941), pushes (LR:822-828) and reset velocities
       (LR:816) are -- with the per-env payload COM displacement of the base included (LR:1025-1028); 0 = the velocity of the link origin
       (rounds 1-4 of this build).  Positions are the link origin's in both.  Injected tensors (LSIM_STEP_SKIP_PHYSICS) are taken as they
       are.  make_lsim_config sets 1. */
    int32_t lin_vel_at_com;
    /* solver_type 1 only: velocity-level Gauss-Seidel passes over the joint-limit rows ALONE after the last position iteration (contact
       impulses frozen, bounds of the configuration reached).  One relaxation per position iteration leaves the limit rows of a stiffly
       coupled leg short of their bounds; one such pass brings the share of joint speeds beyond the URDF limit to that of 8 PGS sweeps
       (DESIGN.md section 4).  0 = none (round 4); make_lsim_config sets 1; at most LSIM_MAX_POSITION_ITERATIONS. */
    int32_t tgs_limit_passes;
    int32_t reserved[2];
} lsim_config;

/* ---- device buffers.  Shapes are per handle (N = num_envs); dtype codes below. ---- */
#define LSIM_DT_F32 0
#define LSIM_DT_I64 1
#define LSIM_DT_U8 2   /* torch.bool compatible */
#define LSIM_DT_I32 3
#define LSIM_DT_I16 4

enum lsim_buffer_id {
    LSIM_BUF_OBS = 0,            /* f32 [N,270]   obs_buf, newest frame first (LR:403) */
    LSIM_BUF_PRIV_OBS,           /* f32 [N,238]   privileged_obs_buf (LR:404) */
    LSIM_BUF_REW,                /* f32 [N]       rew_buf (LR:368-380) */
    LSIM_BUF_RESET,              /* u8  [N]       reset_buf (LR:255, LR:329) */
    LSIM_BUF_TIME_OUT,           /* u8  [N]       time_out_buf (LR:260) */
    LSIM_BUF_EXTRAS_TIME_OUTS,   /* u8  [N]       extras["time_outs"]: only refreshed on steps with >=1 reset (LR:358-359) */
    LSIM_BUF_EPISODE_LENGTH,     /* i64 [N]       episode_length_buf (BT:73), caller-writable (HIMR:90-91) */
    LSIM_BUF_ROOT_STATES,        /* f32 [N,13]    gym root state tensor (LR:930) */
    LSIM_BUF_DOF_STATE,          /* f32 [N,12,2]  gym dof state tensor (LR:932) */
    LSIM_BUF_RIGID_BODY_STATES,  /* f32 [N,17,13] (LR:938) */
    LSIM_BUF_CONTACT_FORCES,     /* f32 [N,17,3]  net contact force per body, last sub-step (LR:944) */
    LSIM_BUF_TORQUES,            /* f32 [N,12]    last sub-step torques (LR:146) */
    LSIM_BUF_ACTIONS,            /* f32 [N,12]    clipped actions (LR:130) */
    LSIM_BUF_LAST_ACTIONS,       /* f32 [N,12] */
    LSIM_BUF_LAST_LAST_ACTIONS,  /* f32 [N,12] */
    LSIM_BUF_LAST_DOF_POS,       /* f32 [N,12] */
    LSIM_BUF_LAST_DOF_VEL,       /* f32 [N,12] */
    LSIM_BUF_LAST_TORQUES,       /* f32 [N,12] */
    LSIM_BUF_LAST_ROOT_VEL,      /* f32 [N,6] */
    LSIM_BUF_COMMANDS,           /* f32 [N,4]     (LR:967) */
    LSIM_BUF_BASE_LIN_VEL,       /* f32 [N,3]     (LR:198) */
    LSIM_BUF_BASE_ANG_VEL,       /* f32 [N,3] */
    LSIM_BUF_PROJECTED_GRAVITY,  /* f32 [N,3] */
    LSIM_BUF_FEET_AIR_TIME,      /* f32 [N,4] */
    LSIM_BUF_LAST_CONTACTS,      /* u8  [N,4] */
    LSIM_BUF_CONTACT_FILT,       /* u8  [N,4] */
    LSIM_BUF_MEASURED_HEIGHTS,   /* f32 [N,187]   (LR:624) */
    LSIM_BUF_PENDING_FORCE,      /* f32 [N,3]     body-local force on the base drawn at LR:842-844, consumed (and cleared) by the
                                                  first sub-step of the next step; the reference's self.disturbance is zero
                                                  again after every step (LR:235), its value survives only inside priv obs */
    LSIM_BUF_TERRAIN_LEVELS,     /* i64 [N] */
    LSIM_BUF_TERRAIN_TYPES,      /* i64 [N] */
    LSIM_BUF_ENV_ORIGINS,        /* f32 [N,3] */
    LSIM_BUF_KP_FACTORS,         /* f32 [N] */
    LSIM_BUF_KD_FACTORS,         /* f32 [N] */
    LSIM_BUF_MOTOR_STRENGTH,     /* f32 [N,12]    drawn once (LR:999-1007) */
    LSIM_BUF_MOTOR_STRENGTH_FACTORS, /* f32 [N]   redrawn at reset, never used (quirk 8) */
    LSIM_BUF_FRICTION,           /* f32 [N] */
    LSIM_BUF_RESTITUTION,        /* f32 [N] */
    LSIM_BUF_PAYLOAD,            /* f32 [N] */
    LSIM_BUF_COM_DISPLACEMENT,   /* f32 [N,3] */
    LSIM_BUF_EPISODE_SUMS,       /* f32 [N,LSIM_NUM_REWARD_TERMS] */
    LSIM_BUF_TERM_PRIV_OBS,      /* f32 [N,238]   rows valid where reset_buf (LR:227) */
    LSIM_BUF_TERM_AMP_OBS,       /* f32 [N,30]    rows valid where reset_buf (LR:228) */
    LSIM_BUF_AMP_OBS,            /* f32 [N,30]    get_amp_observations() of the post-step state (LR:406-416) */
    LSIM_BUF_DELAY_STEPS,        /* i32 [N]       last drawn action delay (LR:134) */
    LSIM_BUF_CONTACT_COUNT,      /* i32 [N,2]     diagnostic: collision points within contact_offset of the terrain BEFORE the cap of
                                                  LSIM_MAX_CONTACTS -- [0] maximum over the sub-steps of this step, [1] last sub-step */
    LSIM_BUF_SUBSTEP_TORQUES,    /* f32 [N,decimation,12] diagnostic, written only under LSIM_STEP_RECORD_SUBSTEPS: the torques of EVERY sub-step
                                                  (LR:146 keeps only the last), i.e. _compute_torques(delayed_actions[:, i]) of LR:138-146 -- what
                                                  makes the action-delay model observable from outside */
    LSIM_BUF_STATS,              /* f32 [2,LSIM_STATS_SIZE] device-side per-step reductions, see below */
    LSIM_BUF_NONFINITE,          /* i64 [2]       [0] env-steps since lsim_create in which a robot's simulated state (root 13, joint angles 12, joint velocities 12, after the
                                                  last sub-step) held a NaN or an infinity -- CUMULATIVE, never cleared by the library (the caller may zero it); [1] the
                                                  step counter of the latest such step.  A robot that went non-finite stays so until its episode times out: it is
                                                  counted in every step.  Must stay 0: anything else is a solver blow-up, and until round 6 it was only visible as a slow kernel A */
    LSIM_BUF_HEIGHT_GRID,        /* i16 [rows,cols] */
    LSIM_BUF_TERRAIN_ORIGINS,    /* f32 [levels,types,3] */
    LSIM_BUF_TERRAIN_MESH,       /* i32 [rows,cols] per grid vertex of the reference's triangle mesh: bits 0-15 = height sample (int16);
                                    bits 16-17 = dx+1, bits 18-19 = dy+1 (horizontal displacement of the vertex, in cells: the
                                    slope_treshold vertical-wall correction, TER:72-75); bit 20 = some vertex of the 4x4 block around
                                    cell (i,j) is displaced (contacts there use the exact triangle query) */
    LSIM_NUM_BUFFERS
};

/* layout of one row of LSIM_BUF_STATS.  Two rows ping-pong: every lsim_step / lsim_reset_all call fills the row
 * lsim_get_stats_row() reports after the call and clears the other row for the next call, so a row stays valid
 * until the next call has run (read it, or copy it on the stream, before stepping again).
 *   [0]                      number of envs reset this step
 *   [1 .. 1+T)               sum over reset envs of episode_sums[k] / clip(ep_len,1)   (LR:349; divide by [0] and dt)
 *   [1+T]                    reserved (the host forms mean(terrain_levels), LR:353, from LSIM_BUF_TERRAIN_LEVELS)
 *   [2+T .. 10+T)            command_ranges[4][2] live values (LR:877-880)
 *   [10+T]                   reserved (the tracking sum of LR:875 is kept in fixed point, below)
 *   [11+T]                   reserved
 *   [12+T]                   number of envs whose simulated state was not finite in THIS step (see LSIM_BUF_NONFINITE for the running total)
 *   [LSIM_STATS_FIX ..)      internal, not for the host: int64 fixed-point (2^-32, each addend clamped to +-2^20) accumulators of [1 .. 1+T) and of the sum over reset envs of
 *                            episode_sums[tracking_lin_vel] (LR:875), and a ticket counter.  Waves add to them with integer atomics, so the sums
 *                            -- extras["episode"] and the command-curriculum decision -- do not depend on the order the waves arrive in; the last
 *                            resetting wave of a step converts [1 .. 1+T) to fp32.
 */
#define LSIM_STATS_RESET_COUNT 0
#define LSIM_STATS_EPISODE_SUMS 1
#define LSIM_STATS_LEVEL_SUM (1 + LSIM_NUM_REWARD_TERMS)
#define LSIM_STATS_CMD_RANGES (2 + LSIM_NUM_REWARD_TERMS)
#define LSIM_STATS_TRACK_SUM (10 + LSIM_NUM_REWARD_TERMS)
#define LSIM_STATS_RESET_STEPS (11 + LSIM_NUM_REWARD_TERMS)
#define LSIM_STATS_NONFINITE (12 + LSIM_NUM_REWARD_TERMS)
#define LSIM_STATS_FIX ((17 + LSIM_NUM_REWARD_TERMS) & ~1)      /* even float index: the int64 words are 8-byte aligned (rows are too) */
#define LSIM_STATS_FIX_TRACK LSIM_NUM_REWARD_TERMS             /* word index of the tracking sum */
#define LSIM_STATS_FIX_TICKET (LSIM_NUM_REWARD_TERMS + 1)      /* word index of the ticket counter */
#define LSIM_STATS_FIX_WORDS (LSIM_NUM_REWARD_TERMS + 2)
#define LSIM_STATS_SIZE (LSIM_STATS_FIX + 2 * LSIM_STATS_FIX_WORDS)

/* flags of lsim_step_ex */
#define LSIM_STEP_DEFAULT 0u
#define LSIM_STEP_SKIP_PHYSICS 1u   /* test hook: use ROOT/DOF/RIGID_BODY/CONTACT buffers as injected by the caller
                                       instead of simulating; still computes torques (E3) for the 4 sub-steps */
#define LSIM_STEP_NO_RESET 2u       /* test hook: compute reset_buf but do not reset_idx */
#define LSIM_STEP_RECORD_SUBSTEPS 4u /* test hook: also write LSIM_BUF_SUBSTEP_TORQUES (one 48-byte store per sub-step and robot) */
#define LSIM_STEP_TWO_KERNELS 8u    /* test / measurement hook: run reset_idx + observations as the separate kernel B on every step (the form the
                                       command-curriculum steps always take) instead of inside kernel A.  Same results, bit for bit */
#define LSIM_STEP_FLAT_PRIORITY 16u /* measurement hook: every wave of kernel A at the default issue priority (by default a robot's wave is raised with
                                       its number of contacts, so that the launch's slowest waves are not also waiting for their turn).  Same results */

typedef struct lsim_sim* lsim_handle;

/* sizeof() of the two structs as compiled into the library -- lets a foreign binding verify its mirror. */
int lsim_sizeof_config(void);
int lsim_sizeof_model(void);
int lsim_abi_version(void);

/* bytes of device memory one simulator instance needs (all buffers, 256-B aligned). */
int lsim_query_arena(const lsim_config* cfg, size_t* bytes_out);

/* replaces gym.create_sim + add_triangle_mesh/add_heightfield/add_ground + load_asset + create_env/create_actor
 * + acquire_*_tensor + prepare_sim (LR:467, LR:1069-1104, LR:1135, LR:1184-1205, LR:917-920, BT:85) and the
 * buffer allocation of BaseTask.__init__/LeggedRobot._init_buffers (BT:70-79, LR:913-1032).
 * height_grid: host int16 [grid_rows*grid_cols] (may be NULL for mesh_type plane);
 * terrain_origins: host float [terrain_num_rows*terrain_num_cols*3] (may be NULL for plane). */
int lsim_create(const lsim_config* cfg, const lsim_robot_model* model,
                const int16_t* height_grid, const float* terrain_origins,
                void* arena_dev, int device_id, lsim_handle* out);

/* replaces gymtorch.wrap_tensor(acquire_*) (LR:930-944): device pointer + shape of one buffer. */
int lsim_get_buffer(lsim_handle h, int buffer_id, void** dev_ptr, int64_t shape[4], int* ndim, int* dtype);

/* LeggedRobot.reset_idx(all envs) as called by BaseTask.reset (BT:113); the caller follows it with one
 * zero-action lsim_step to complete reset() (BT:114). */
int lsim_reset_all(lsim_handle h, void* hip_stream);

/* LeggedRobot.reset_idx(env_ids) (LR:290-361) called from outside a step (a play / evaluation script resetting some robots by hand):
 * terrain curriculum per env, command curriculum over the reset set, dof / root / command / domain-randomisation redraws, buffer
 * clears, extras["episode"] sums into the stats row, episode_length_buf = 0.  reset_mask_dev: device uint8 [N], nonzero = reset this env
 * (the boolean form of env_ids; it is read by the kernels of this call only).  An all-zero mask is the reference's early return
 * (LR:298): only the stats rows swap, with a reset count of 0.  As after lsim_reset_all, observations are not recomputed (the reference
 * does not either): they are those of the next lsim_step.  Random draws are keyed by (env, common_step_counter, number of lsim_reset_envs
 * calls on this handle so far): every call draws fresh values, as every reset_idx of the reference does -- resetting an env that already
 * reset in the adjacent step, or the same env twice between two steps, gives it a new state each time.  Asynchronous on hip_stream. */
int lsim_reset_envs(lsim_handle h, const uint8_t* reset_mask_dev, void* hip_stream);

/* LeggedRobot.step(actions) (LR:122-176): replaces set_dof_actuation_force_tensor/simulate/fetch_results/
 * refresh_* x4 (LR:146-152), refresh_* (LR:187-190), set_*_indexed (LR:714, LR:818), set_actor_root_state_tensor
 * (LR:828), apply_rigid_body_force_tensors (LR:844) and the whole post_physics_step (LR:178-247).
 * actions_dev: device float [N,12].  Asynchronous on hip_stream. */
int lsim_step(lsim_handle h, const float* actions_dev, void* hip_stream);
int lsim_step_ex(lsim_handle h, const float* actions_dev, uint32_t flags, void* hip_stream);

/* host-side scalars (no device sync): common_step_counter (LR:194). */
int lsim_get_step_counter(lsim_handle h, int64_t* counter_out);
int lsim_set_step_counter(lsim_handle h, int64_t counter);
/* number of lsim_reset_envs calls on this handle so far (the salt of their random draws): a resumed run that restores it together with the
 * step counter redraws the same by-hand reset states as an uninterrupted one. */
int lsim_get_reset_calls(lsim_handle h, uint32_t* calls_out);
int lsim_set_reset_calls(lsim_handle h, uint32_t calls);
/* which row (0/1) of LSIM_BUF_STATS the most recent lsim_step / lsim_reset_all filled. */
int lsim_get_stats_row(lsim_handle h, int* row_out);

/* measurement aid (replaces the reference's time.time() bracketing, HIMR:106-147): with capacity > 0 every following
 * lsim_step records HIP events around its kernels on the caller's stream (no host sync); lsim_read_profile waits for
 * the last recorded step and returns per-step durations in milliseconds of kernel A (physics + post-physics) and
 * kernel B (reset + observations) for the most recent min(n_steps, capacity) steps; *n_inout: in = array length,
 * out = number of entries written.  capacity == 0 disables and frees the events. */
int lsim_set_profiling(lsim_handle h, int capacity);
int lsim_read_profile(lsim_handle h, float* ms_kernel_a, float* ms_kernel_b, int* n_inout);

/* name of a reward term / buffer (for bindings and logs); NULL if out of range. */
const char* lsim_reward_name(int reward_id);
const char* lsim_buffer_name(int buffer_id);

const char* lsim_last_error(lsim_handle h);
void lsim_destroy(lsim_handle h);

/* ---- rollout-side fused kernels (SURVEY.md 8f "fused storage"): the elementwise / storage half of the on-policy rollout
 * step.  The networks' GEMMs stay with the caller (PyTorch); these two calls replace the ~35 small torch kernels of
 *   HIMPPO.act                (HIMP:90-103: sample a ~ N(mean, std), log-prob, keep mean/std/values/observations),
 *   HIMPPO.process_env_step   (HIMP:105-118: next critic obs with the termination rows patched in, HIMR:119-121;
 *                              time-out bootstrap  r += gamma * V * time_out),
 *   HIMRolloutStorage.add_transitions (HST:92-106: eleven copies into the [T, N, .] storage at the step index).
 * Stateless: every argument is a device pointer (or scalar) of the caller; `step_idx_dev` and `draw_counter_dev` live
 * in device memory so that both calls can be captured in a HIP graph and replayed.  All pointers are fp32 unless noted. */
typedef struct lsim_rollout_storage {
    float* observations;                 /* [T, N, num_obs]       HST:60 */
    float* privileged_observations;      /* [T, N, num_priv_obs]  HST:62 */
    float* next_privileged_observations; /* [T, N, num_priv_obs]  HST:63 */
    float* actions;                      /* [T, N, num_actions] */
    float* values;                       /* [T, N, 1] */
    float* actions_log_prob;             /* [T, N, 1] */
    float* mu;                           /* [T, N, num_actions] */
    float* sigma;                        /* [T, N, num_actions] */
    float* rewards;                      /* [T, N, 1] */
    uint8_t* dones;                      /* [T, N, 1] u8 */
    int32_t num_steps, num_envs, num_obs, num_priv_obs, num_actions;   /* num_actions <= 32; row sizes even */
} lsim_rollout_storage;

/* actions_out[N, A] = mean + std * z with z ~ N(0,1) from Philox4x32-10 keyed (seed, rank), counter
 * (env, *draw_counter_dev, LSIM_RNG_POLICY, pair index), Box-Muller; log-prob summed over the action dimension;
 * writes storage row *step_idx_dev of observations, privileged_observations, actions, values, actions_log_prob, mu, sigma.
 * mean [N, A], std [A] (the policy's std parameter), values [N, 1], obs [N, num_obs], priv_obs [N, num_priv_obs]. */
int lsim_rollout_act(const lsim_rollout_storage* st, const int64_t* step_idx_dev, const int64_t* draw_counter_dev,
                     const float* mean, const float* std, const float* values, const float* obs, const float* priv_obs,
                     uint32_t seed, uint32_t rank, float* actions_out, void* stream);
/* after lsim_step: writes storage row *step_idx_dev of next_privileged_observations (term_priv_obs rows where dones, else
 * priv_obs), rewards (+ gamma * values * time_outs when time_outs != NULL), dones; then a second tiny kernel advances
 * *step_idx_dev and *draw_counter_dev by one.  dones / time_outs are u8 [N]; values is the [N, 1] critic output kept
 * from lsim_rollout_act's inputs. */
int lsim_rollout_post(const lsim_rollout_storage* st, int64_t* step_idx_dev, int64_t* draw_counter_dev,
                      const uint8_t* dones, const uint8_t* time_outs, const float* rewards, const float* values,
                      const float* priv_obs, const float* term_priv_obs, float gamma, void* stream);
/* lsim_rollout_act / lsim_rollout_post with the storage row and the sampler's counter passed BY VALUE (a host-driven rollout loop knows
 * both): same kernels, same results, and no second launch to advance device-side counters -- the caller advances its own.
 * LSIM_E_INVALID when step_idx is outside [0, num_steps). */
int lsim_rollout_act_at(const lsim_rollout_storage* st, int64_t step_idx, int64_t draw_counter,
                        const float* mean, const float* std, const float* values, const float* obs, const float* priv_obs,
                        uint32_t seed, uint32_t rank, float* actions_out, void* stream);
int lsim_rollout_post_at(const lsim_rollout_storage* st, int64_t step_idx, const uint8_t* dones, const uint8_t* time_outs,
                         const float* rewards, const float* values, const float* priv_obs, const float* term_priv_obs, float gamma,
                         void* stream);
/* GAE(lambda) reverse sweep of HIMRolloutStorage.compute_returns (HST:113-123), one thread per env over the T stored steps:
 *   delta = r_t + (1 - done_t) * gamma * V_{t+1} - V_t;   A_t = delta + (1 - done_t) * gamma * lam * A_{t+1};   returns_t = A_t + V_t
 * with V_T = last_values [N, 1].  Writes returns [T, N, 1] and the raw advantages returns - values [T, N, 1]; the batch
 * normalisation of HST:126-127 (a global mean / std, reduced over ranks when data parallel) stays with the caller. */
int lsim_rollout_gae(const lsim_rollout_storage* st, const float* last_values, float gamma, float lam,
                     float* returns, float* advantages, void* stream);

/* ---- fused rollout-time forward of the HIM policy (HAC:136-163, HES:64-68, HIMP:90-96): estimator encoder
 * (observation history -> 3 velocity + latent), L2-normalised latent, actor on [one-step obs, velocity, latent], critic on the
 * privileged observation -- 11 Linear layers with ELU between them -- in one launch, MFMA fp32, activations in LDS.
 * Weights are passed PADDED and TILED: both dimensions rounded up to multiples of 16 (zero filled beyond [n_out][k_in]) and stored as
 * 16 x 16 blocks, weight[(n / 16) * (k_pad / 16) + k / 16][n % 16][k % 16] -- the 1 KB block of output tile n / 16 and k chunk k / 16 is
 * contiguous, so the wave that owns an output tile reads whole cache lines, each once (ABI v5; v4 took [n_pad][k_pad] row-major: half a
 * line per row and chunk, with the second half evicted from the CU's 32 KB vector cache before its turn when a wave owned two tiles) --
 * bias [n_pad]; 16-byte aligned (the caller packs torch's nn.Linear parameters once per policy update).
 * mean_out [num_envs, num_actions], values_out [num_envs, 1].  LSIM_E_UNSUPPORTED for other topologies / sizes (hidden widths > 512,
 * inputs > 272): run the networks in the host framework then. */
typedef struct lsim_mlp_layer {
    const float* weight;      /* [n_pad / 16][k_pad / 16][16][16]: see above */
    const float* bias;        /* [n_pad] */
    int32_t k_pad, n_pad, k_in, n_out;
} lsim_mlp_layer;
typedef struct lsim_him_policy {
    lsim_mlp_layer encoder[3];   /* HES:36-45  history -> ... -> 3 + latent (no activation after the last layer) */
    lsim_mlp_layer actor[4];     /* HAC:66-80  one_step_obs + 3 + latent -> ... -> num_actions */
    lsim_mlp_layer critic[4];    /* HAC:82-95  privileged obs -> ... -> 1 */
    int32_t num_obs, num_priv_obs, num_one_step_obs, num_actions;
} lsim_him_policy;
int lsim_policy_forward(const lsim_him_policy* p, const float* obs, const float* priv_obs, int64_t num_envs, float* mean_out,
                        float* values_out, void* stream);

/* lsim_policy_forward and lsim_rollout_act_at in ONE launch: the blocks that evaluate the networks also copy the observation rows into
 * storage row step_idx, sample the actions (same Philox draws, same log-probability sums as lsim_rollout_act) and store actions, values,
 * log-prob, mu, sigma.  num_envs = st->num_envs; the storage's observation widths and action count must equal the policy's. */
int lsim_policy_act_at(const lsim_him_policy* p, const lsim_rollout_storage* st, int64_t step_idx, int64_t draw_counter,
                       const float* obs, const float* priv_obs, const float* std, uint32_t seed, uint32_t rank,
                       float* mean_out, float* values_out, float* actions_out, void* stream);

/* lsim_policy_act_at that also performs the PREVIOUS step's lsim_rollout_post_at (prev_step < 0: none): the privileged observation the critic
 * blocks stage for this step is the previous step's next critic observation (termination rows patched in from prev_term_priv_obs where
 * prev_dones), and values_out still holds the previous step's values when they form  r + gamma * V * time_out.  The caller stores the last
 * step of a rollout with lsim_rollout_post_at.  Same results as the separate calls. */
int lsim_policy_act_post_at(const lsim_him_policy* p, const lsim_rollout_storage* st, int64_t step_idx, int64_t draw_counter,
                            const float* obs, const float* priv_obs, const float* std, uint32_t seed, uint32_t rank,
                            float* mean_out, float* values_out, float* actions_out,
                            int64_t prev_step, const uint8_t* prev_dones, const uint8_t* prev_time_outs, const float* prev_rewards,
                            const float* prev_term_priv_obs, float gamma, void* stream);

/* ---- the AMP rollout step in one launch (SURVEY.md 8f rank 4 "discriminator reward fused after the step"): what HybridPolicyRunner does between
 * env.step() and process_env_step() (rsl_rl/runners/hybrid_runner.py:183-200) --
 *   next' = where(dones, terminal_amp_states, next_amp_obs)                                  HYBR:191-192
 *   d = head(relu(W2 relu(W1 [norm(amp_obs) | norm(next')] + b1) + b2))                      amp_discriminator.py:55-62, utils/utils.py:124-130
 *   reward = reward_coef * max(1 - (d - 1)^2 / 4, 0);  lerp > 0: (1 - lerp) * reward + lerp * task_reward     amp_discriminator.py:63-72
 *   replay ring rows (cursor + env) % capacity <- (amp_obs, next')                            storage/replay_buffer.py:52-68
 *   amp_obs_carry <- next_amp_obs (the NEXT step's amp_obs, un-patched)                       HYBR:196
 * The two trunk layers use lsim_mlp_layer's padded 16 x 16-block layout (see lsim_him_policy); head_weight [hidden[1].n_pad] zero padded, head_bias [1];
 * norm_mean / norm_var: the running moments as the reference keeps them (float64 [amp_dim]; both NULL = no normaliser).
 * amp_dim <= 32, hidden[0].n_pad <= 1024, hidden[1].n_pad <= 1024; fp32 MFMA, activations in LDS, nothing but the outputs written. */
typedef struct lsim_amp_disc {
    lsim_mlp_layer hidden[2];    /* DISC:18-25: Linear + ReLU, Linear + ReLU */
    const float* head_weight;    /* DISC:27 amp_linear.weight */
    const float* head_bias;      /* amp_linear.bias [1], device */
    const double* norm_mean;     /* UT:78-106 */
    const double* norm_var;
    double norm_eps, norm_clip;  /* UT:114-116: 1e-4, 10 */
    double reward_coef, task_reward_lerp;   /* DISC:14-15 (python floats) */
    int32_t amp_dim, reserved;
} lsim_amp_disc;
/* workspace: lsim_amp_step_workspace() bytes, 16-byte aligned, ZERO-FILLED by the caller before its first use (every call leaves it ready for the next).
 * amp_obs, next_amp_obs, terminal_amp_states [num_envs, amp_dim]; dones u8 [num_envs] or NULL (no patching); task_rewards [num_envs] (may be NULL when
 * lerp == 0); rewards_out [num_envs]; disc_out [num_envs] or NULL (the discriminator's raw output d); amp_obs_carry [num_envs, amp_dim] or NULL, must not
 * alias amp_obs; replay_states / replay_next_states [replay_capacity, amp_dim] or both NULL, replay_capacity >= num_envs, 0 <= replay_cursor < capacity
 * (the caller advances its cursor by num_envs modulo capacity, RB:66-68). */
int lsim_amp_step_workspace(int64_t num_envs, size_t* bytes);
int lsim_amp_step(const lsim_amp_disc* d, const float* amp_obs, const float* next_amp_obs, const uint8_t* dones, const float* terminal_amp_states,
                  const float* task_rewards, int64_t num_envs, float* rewards_out, float* disc_out, float* amp_obs_carry,
                  float* replay_states, float* replay_next_states, int64_t replay_capacity, int64_t replay_cursor,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- learner-side kernel: weight / bias gradient of a small Linear layer over a tall minibatch,
 *   dw[n, k] = sum_b g[b, n] * x[b, k],   db[n] = sum_b g[b, n]      (torch.nn.Linear backward: grad_weight = g^T x, grad_bias = g.sum(0))
 * for ceil(n_out / 16) * ceil(k_in / 16) <= 32 (each <= 8): the heads and narrow layers of HAC:66-95 / HES:36-54 / DISC:18-25, whose
 * K = 102 400 reductions BLAS runs at a few percent of peak.  x [batch, k_in] with row stride ldx, g [batch, n_out] with row stride ldg
 * (floats), dw [n_out, k_in] and db [n_out] (db may be NULL) contiguous.  `workspace` is a caller-allocated device buffer of at least
 * lsim_linear_wgrad_workspace() bytes holding the per-wave partial results (summed in a fixed order: deterministic).
 * Returns LSIM_E_UNSUPPORTED for larger layers (use BLAS). */
int lsim_linear_wgrad_workspace(long batch, int k_in, int n_out, size_t* bytes, int* num_waves);
int lsim_linear_wgrad(const float* x, int64_t ldx, const float* g, int64_t ldg, int64_t batch, int k_in, int n_out,
                      float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* Forward of a Linear layer followed by ELU (alpha = 1): out[b, :] = elu(x[b, :] W^T + bias), the hidden layers of the actor, critic and
 * estimator MLPs (rsl_rl/modules/him_actor_critic.py:52-76, him_estimator.py:40-62 build them as nn.Linear + nn.ELU pairs; HIMPPO.update
 * runs them on minibatches of 102 400 rows, him_ppo.py:136-150).  fp32 MFMA with the activation applied to the accumulators: the layer's
 * output is written once instead of BLAS output + elementwise read + write.  x [batch, k_in] with leading dimension ldx, weight
 * [n_out, k_in] contiguous (nn.Linear.weight), bias [n_out] or NULL, out [batch, n_out] with leading dimension ldo.
 * LSIM_E_UNSUPPORTED unless n_out % 4 == 0, ldo % 4 == 0 and out is 16-byte aligned (use BLAS + ELU). */
int lsim_linear_elu_forward(const float* x, int64_t ldx, const float* weight, const float* bias, int64_t batch, int k_in, int n_out,
                            float* out, int64_t ldo, void* stream);

/* Backward of a Linear layer followed by ELU (y = x W^T + b, z = elu(y), alpha = 1) given the gradient of z: the gradient of the
 * pre-activation  grad_pre = grad_out * (z > 0 ? 1 : z + 1)  (torch's elu_backward on the saved OUTPUT) is formed on the fly as the MFMA
 * operand of the weight-gradient kernel and written once -- [batch, n_out] contiguous -- for the caller's input-gradient GEMM
 * (grad_pre @ W); dw / db as lsim_linear_wgrad.  One pass over grad_out and z instead of elu_backward + column sum + wgrad.
 * grad_pre may be NULL when no input gradient will be formed (the first layer of a network): the gradient of the pre-activation is then
 * only used inside the kernel and never written.
 * Same workspace as lsim_linear_wgrad; LSIM_E_UNSUPPORTED for shapes lsim_linear_wgrad handles in its single-wave form (<= 4096 outputs). */
int lsim_linear_elu_wgrad(const float* x, int64_t ldx, const float* grad_out, int64_t ldg, const float* elu_out, int64_t ldz, int64_t batch,
                          int k_in, int n_out, float* dw, float* db, float* grad_pre, void* workspace, size_t workspace_bytes, void* stream);

/* The same for a Linear layer followed by ReLU (the AMP discriminator's trunk, amp_discriminator.py:18-25), given the saved ReLU OUTPUT:
 * grad_pre = grad_out * [relu_out > 0] (torch's threshold_backward). */
int lsim_linear_relu_wgrad(const float* x, int64_t ldx, const float* grad_out, int64_t ldg, const float* relu_out, int64_t ldz, int64_t batch,
                           int k_in, int n_out, float* dw, float* db, float* grad_pre, void* workspace, size_t workspace_bytes, void* stream);

/* out[b, n] = mask_src[b, n] > 0 ? sum_k x[b, k] weight[n, k] : 0: a product consumed by a ReLU's backward, mask applied to the accumulators (no
 * separate threshold_backward pass).  Limits as lsim_linear_elu_forward; mask_src [batch, n_out] with leading dimension ldm % 4 == 0, 16-byte aligned. */
int lsim_linear_masked_forward(const float* x, int64_t ldx, const float* weight, const float* mask_src, int64_t ldm, int64_t batch, int k_in, int n_out,
                               float* out, int64_t ldo, void* stream);

/* ---- discriminator update (HybridPPO.update, hybrid_ppo.py:236-281): the elementwise passes between the GEMMs of the LSGAN loss / gradient penalty.
 * lsim_relu_head_backward: for the head d = relu_out . head_weight + b on relu_out [batch, n] (leading dimension ld) and grad_d [batch] = d loss / d d:
 *   grad_pre[b, j] = relu_out[b, j] > 0 ? grad_d[b] * head_weight[j] : 0  ([batch, n] contiguous),  grad_bias[j] = sum_b grad_pre[b, j],
 *   grad_head [n + 4]: [j < n] = sum_b relu_out[b, j] * grad_d[b],  [n] = sum_b grad_d[b], three zeros  -- one pass, sums in a fixed order.
 * lsim_masked_colsum: out[j] = sum_b (mask_src[b, j] > 0 ? v[b, j] : 0).
 * n % 4 == 0, n <= 1024, n / 4 a divisor of 256; 16-byte aligned rows; workspace lsim_relu_cols_workspace() bytes.  LSIM_E_UNSUPPORTED otherwise. */
int lsim_relu_cols_workspace(int64_t batch, int n, size_t* bytes);
int lsim_relu_head_backward(const float* relu_out, int64_t ld, const float* grad_d, const float* head_weight, int64_t batch, int n,
                            float* grad_pre, float* grad_bias, float* grad_head, void* workspace, size_t workspace_bytes, void* stream);
int lsim_masked_colsum(const float* v, int64_t ldv, const float* mask_src, int64_t ldm, int64_t batch, int n, float* out,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Normalizer.update (utils/utils.py:86-106; twice per minibatch at hybrid_ppo.py:279-281): the batch moments of x [batch, dim <= 64] merged into the
 * float64 running mean / var / count (device arrays [dim], [dim], [1], updated in place) with the parallel-variance formula; batch sums in float64.
 * Two launches, no host round trip.  workspace: lsim_running_moments_workspace() bytes, 8-byte aligned. */
int lsim_running_moments_workspace(size_t* bytes);
int lsim_running_moments_update(const float* x, int64_t ldx, int64_t batch, int dim, double* mean, double* var, double* count,
                                void* workspace, size_t workspace_bytes, void* stream);

/* The discriminator's input rows of one sampled block in HybridPPO.update (hybrid_ppo.py:247-251 normalize_torch on state and next state, then
 * amp_discriminator.py:57 / :37 torch.cat([state, next_state], dim=-1)):  out[b, 0:dim] = f(states[b]), out[b, dim:2 dim] = f(next_states[b]),
 * f(x) = clamp((x - float32(mean)) / sqrt(float32(var + eps)), +-clip) (utils/utils.py:124-130), or the identity when norm_mean == NULL (the gradient
 * penalty's un-normalised pair).  One launch; `out` may point into a larger stacked evaluation (row pitch ld_out >= 2 dim). */
int lsim_amp_pair_rows(const float* states, int64_t ld_states, const float* next_states, int64_t ld_next, const double* norm_mean,
                       const double* norm_var, double norm_eps, double norm_clip, int64_t batch, int dim, float* out, int64_t ld_out, void* stream);

/* Opt-in form of the two calls above for layers whose k_in and n_out are multiples of 128 (and whose operands are 16-byte aligned): the same fp32
 * sums on the bf16 matrix pipe.  Every fp32 operand is split exactly into three bf16 terms (3 x 8 significand bits) and the six products of
 * order <= 2 are accumulated in fp32: products exact, truncation 2^-24 |a||b| per product -- fp32's own rounding (measured against fp64 sums:
 * at or below the fp32 pipe's error, tools/micro/split_bf16.hip, tests/test_gpu_learner.py).  fp32 in, fp32 out, fp32 accumulate; a different
 * instruction stream than the default, hence a switch: on = 1 / 0 sets it, anything else only queries; returns the previous setting.
 * Initial value: environment variable LSIM_WGRAD_SPLIT_BF16=1, else off.  Changes what lsim_linear_wgrad_workspace() reports: set it before
 * sizing workspaces. */
int lsim_wgrad_split_bf16(int on);

/* The same two calls with the final, fixed-order sum of the partial results left to the caller: `pending` receives what that sum needs, the
 * workspace must stay untouched until lsim_wgrad_reduce_batch has run for it, and dw / db hold nothing until then.  A backward pass
 * collects the records of all its layers and sums them in ONE launch behind the last weight-gradient kernel (15 launches of a few
 * microseconds per minibatch become one); same arithmetic, same results bit for bit.  items: host array; any n (the call splits it). */
typedef struct lsim_wgrad_pending {
    const float* part; const float* part2;      /* partial results of dw [num_partials, count] and of db [num_partials, count2] (NULL: no bias) */
    float* out; float* out2;                    /* dw, db */
    int32_t num_partials, count, count2, reserved;
} lsim_wgrad_pending;
int lsim_linear_wgrad_deferred(const float* x, int64_t ldx, const float* g, int64_t ldg, int64_t batch, int k_in, int n_out,
                               float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream, lsim_wgrad_pending* pending);
int lsim_linear_elu_wgrad_deferred(const float* x, int64_t ldx, const float* grad_out, int64_t ldg, const float* elu_out, int64_t ldz, int64_t batch,
                                   int k_in, int n_out, float* dw, float* db, float* grad_pre, void* workspace, size_t workspace_bytes, void* stream,
                                   lsim_wgrad_pending* pending);
int lsim_linear_relu_wgrad_deferred(const float* x, int64_t ldx, const float* grad_out, int64_t ldg, const float* relu_out, int64_t ldz, int64_t batch,
                                    int k_in, int n_out, float* dw, float* db, float* grad_pre, void* workspace, size_t workspace_bytes, void* stream,
                                    lsim_wgrad_pending* pending);
int lsim_wgrad_reduce_batch(const lsim_wgrad_pending* items, int n, void* stream);

/* dst[r, :] = src[index[r], :] for 4-byte elements (rows of `cols` elements, both contiguous; index: int64 [n] on the device): the once-per-update
 * shuffle of the rollout storage through the minibatch permutation (HST:140-164) at copy bandwidth -- torch's advanced indexing computes an
 * offset per element (2.6 TB/s on these row widths). */
int lsim_gather_rows(const void* src, int64_t cols, const int64_t* index, int64_t n, void* dst, void* stream);
/* the same with destination rows `dst_ld` elements apart (dst_ld >= cols; elements cols .. dst_ld - 1 of a row are left alone): HIMRolloutStorage keeps its
 * shuffled observation fields in rows padded to 16 bytes so that the first layers of the networks read aligned rows */
int lsim_gather_rows_ld(const void* src, int64_t cols, const int64_t* index, int64_t n, void* dst, int64_t dst_ld, void* stream);

/* w[r, :] /= max(||w[r, :]||_2, eps) in place for a small matrix (rows * cols <= 4096): torch.nn.functional.normalize(w, dim=-1, p=2, eps) written
 * back, as HIMEstimator.update does with its prototypes before every loss evaluation (HES:80-81) -- one launch instead of clone, norm, clamp,
 * divide and copy.  Sums in column order: may differ from torch's reduction in the last bit. */
int lsim_normalize_rows(float* w, int rows, int cols, float eps, void* stream);

/* Clipped-PPO loss of HIMPPO.update (HIMP:136-176), forward AND backward in one pass: per-sample Gaussian log-prob, ratio, clipped
 * surrogate, clipped value loss, entropy bonus, and the KL estimate of the adaptive learning-rate rule (HIMP:144-156).
 *   out5 = { mean surrogate, mean value loss, mean entropy, mean KL, total = surrogate + value_loss_coef * value - entropy_coef * entropy }
 *   grad_mu [B, A], grad_sigma [B, A], grad_value [B] = d total / d (mu, sigma, value), torch's sub-gradient conventions.
 * mu, sigma, actions, old_mu, old_sigma [B, A]; value, old_logp, advantages, returns, target_values [B], all contiguous fp32.
 * target_values may be NULL when use_clipped_value_loss == 0.  workspace: lsim_ppo_loss_workspace() bytes; deterministic. */
int lsim_ppo_loss_workspace(long batch, size_t* bytes);
int lsim_ppo_loss(const float* mu, const float* sigma, const float* value, const float* actions, const float* old_logp, const float* advantages,
                  const float* returns, const float* target_values, const float* old_mu, const float* old_sigma, int64_t batch, int num_actions,
                  float clip_param, float value_loss_coef, float entropy_coef, int use_clipped_value_loss,
                  float* out5, float* grad_mu, float* grad_sigma, float* grad_value, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the state-independent standard deviation of HIMActorCritic (HAC:93: `std`, one value per action) passed as what it is: `std`
 * [A] instead of its broadcast sigma [B, A] (the reference forms mean * 0 + std, HAC:147), and grad_std [A] = the column sums of what
 * grad_sigma would hold, added up in a fixed order -- the broadcast, its backward and the column sum (five launches and 10 MB per minibatch)
 * disappear.  workspace: lsim_ppo_loss_std_workspace() bytes.  num_actions <= 60. */
int lsim_ppo_loss_std_workspace(long batch, int num_actions, size_t* bytes);
int lsim_ppo_loss_std(const float* mu, const float* std, const float* value, const float* actions, const float* old_logp, const float* advantages,
                      const float* returns, const float* target_values, const float* old_mu, const float* old_sigma, int64_t batch, int num_actions,
                      float clip_param, float value_loss_coef, float entropy_coef, int use_clipped_value_loss,
                      float* out5, float* grad_mu, float* grad_std, float* grad_value, void* workspace, size_t workspace_bytes, void* stream);


/* Adaptive learning rate of HIMPPO (HIMP:144-156) evaluated on the device: *lr_dev /= factor when *kl_mean_dev > 2 desired_kl,
 * *= factor when 0 < *kl_mean_dev < desired_kl / 2, clamped to [lr_min, lr_max].  Both scalars live in device memory (the optimisers read
 * the same lr tensor), so the reference's per-minibatch host read-back of the KL estimate is not needed. */
int lsim_adaptive_lr(const float* kl_mean_dev, float desired_kl, float lr_min, float lr_max, float factor, float* lr_dev, void* stream);

/* Sinkhorn-Knopp assignment of the estimator's prototype scores (HIMEstimator.sinkhorn, HES:119-133; no gradient flows through it):
 *   Q = exp(scores / eps)^T;  Q /= sum(Q);  iters x { Q /= rowsum; Q /= K; Q /= colsum; Q /= B };  out = (Q * B)^T
 * scores [batch, K] with row stride lds (floats), K <= 64, out [batch, K] contiguous.  Computed as E * u[k] * v[b] with 2 * iters + 1
 * small launches instead of ~26 passes over the matrix; `workspace` holds E and the per-block partial sums
 * (lsim_sinkhorn_workspace() bytes, 16-byte aligned like `out`); sums are formed in a fixed order (deterministic). */
int lsim_sinkhorn_workspace(long batch, int K, size_t* bytes);
int lsim_sinkhorn(const float* scores, int64_t lds, int64_t batch, int K, float eps, int iters, float* out,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Loss head of HIMEstimator.update (HES:76-108) forward AND backward, everything between the two encoder outputs and the scalar loss:
 *   enc_out [batch, 3 + latent] (row stride ld_enc): predicted base velocity | student latent;  tgt_out [batch, latent]: target latent;
 *   proto [K, latent] contiguous, rows already L2-normalised (HES:92-93);  vel [batch, 3] (row stride ld_vel): the regression target.
 *   z = F.normalize(latent), scores = z proto^T, q = Sinkhorn(scores) (no gradient), swap = -0.5 mean(q_s log_softmax(S_t / T) +
 *   q_t log_softmax(S_s / T)), est = mse(pred_vel, vel).
 *   losses3 = { est, swap, est + swap };  grad_enc [batch, 3 + latent], grad_tgt [batch, latent], grad_proto [K, latent] contiguous =
 *   d (est + swap) / d (enc_out, tgt_out, proto).  latent <= 32, K <= 64.  10 launches; sums in a fixed order (deterministic).
 * workspace: lsim_estimator_loss_workspace() bytes, 16-byte aligned. */
int lsim_estimator_loss_workspace(int64_t batch, int latent, int K, size_t* bytes);
int lsim_estimator_loss(const float* enc_out, int64_t ld_enc, const float* tgt_out, int64_t ld_tgt, const float* proto, const float* vel,
                        int64_t ld_vel, int64_t batch, int latent, int K, float temperature, float sinkhorn_eps, int sinkhorn_iters,
                        float* losses3, float* grad_enc, float* grad_tgt, float* grad_proto, void* workspace, size_t workspace_bytes,
                        void* stream);

/* Gradient clipping + Adam of one optimiser step in two launches (HIMP:183-184, HES:113-114: clip_grad_norm_(params, max_grad_norm) then
 * torch.optim.Adam.step(), no amsgrad / weight decay / maximize), on the caller's tensors: `count` <= 48 parameters with numel[i] elements
 * each, their gradients (rescaled in place by min(max_grad_norm / (||g|| + 1e-6), 1), as clip_grad_norm_ does; max_grad_norm <= 0: no
 * clipping), first / second moment estimates and per-parameter step counters (float scalars, incremented).  The pointer tables are HOST
 * arrays of device pointers.  Learning rate: *lr_dev if lr_dev != NULL (device scalar, see lsim_adaptive_lr), else lr_host.
 * grad_norm_out (device, may be NULL) receives the total norm before clipping.  workspace: lsim_adam_clip_step_workspace(count) bytes. */
int lsim_adam_clip_step_workspace(int count, size_t* bytes);
int lsim_adam_clip_step(int count, const int64_t* numel, float* const* params, float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, float* const* steps, const float* lr_dev, float lr_host, float beta1, float beta2,
                        float eps, float max_grad_norm, float* grad_norm_out, void* workspace, size_t workspace_bytes, void* stream);

/* lsim_adam_clip_step for an optimiser with several parameter groups (HybridPPO, HYBP:86-92 + HYBP:270-273: one Adam over the actor-critic,
 * the discriminator trunk with weight decay 1e-3 and its head with weight decay 1e-1, gradient clipping over the actor-critic's parameters
 * only): weight_decay[i] >= 0 per tensor (torch.optim.Adam's L2 form, grad + weight_decay * param, applied after the clipping and not written
 * back to the gradient; NULL = none), and only tensors [0, clip_count) enter the clipped norm and are rescaled.  grad_norm_out = that norm. */
int lsim_adam_clip_step_ex(int count, const int64_t* numel, float* const* params, float* const* grads, float* const* exp_avg,
                           float* const* exp_avg_sq, float* const* steps, const float* weight_decay, int clip_count,
                           const float* lr_dev, float lr_host, float beta1, float beta2, float eps, float max_grad_norm,
                           float* grad_norm_out, void* workspace, size_t workspace_bytes, void* stream);

/* Actor input of HIMActorCritic (HAC:136-141; HES:64-68 for the normalisation): out[b] = [ obs[b, :num_one_step_obs] | enc_out[b, :3] |
 * enc_out[b, 3:3+latent] / max(||.||, 1e-12) ], out [batch, num_one_step_obs + 3 + latent] contiguous; obs / enc_out with row strides
 * ld_obs / ld_enc (floats).  No gradient flows through this (the estimator's outputs are detached there). */
int lsim_actor_input(const float* obs, int64_t ld_obs, int num_one_step_obs, const float* enc_out, int64_t ld_enc, int latent,
                     int64_t batch, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LSIM_H */
