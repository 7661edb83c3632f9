/*
 * lsim_layout.h -- shape / dtype / name table of the LSIM_BUF_* buffers (interface data, no logic).
 * Shared by the HIP library, the test oracle and foreign bindings so that all three agree
 * on what lsim_get_buffer() describes.  Shapes follow the reference tensors cited in lsim.h.
 */
#ifndef LSIM_LAYOUT_H
#define LSIM_LAYOUT_H
#include <math.h>
#include "lsim.h"

#ifdef __cplusplus
extern "C" {
#endif

static inline size_t lsim_dtype_size(int dtype) {
    switch (dtype) {
        case LSIM_DT_F32: return 4;
        case LSIM_DT_I64: return 8;
        case LSIM_DT_U8: return 1;
        case LSIM_DT_I32: return 4;
        case LSIM_DT_I16: return 2;
        default: return 0;
    }
}

/* torch.div(a, b, rounding_mode="floor") on fp32 (c10::div_floor_floating): the floor of the EXACT quotient of the two floats, not of
 * their rounded fp32 quotient.  LR:1234 assigns terrain types with it: at N = 4096, 20 columns, env 1024 has 1024 / 204.8f = 4.99999993,
 * which plain fp32 division rounds to 5.0 while torch returns 4 (found by pinning E21, tests/test_init_golden.py). */
static inline float lsim_div_floor_f32(float a, float b) {
    if (b == 0.0f) return a / b;
    float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if (mod != 0.0f && ((b < 0.0f) != (mod < 0.0f))) div -= 1.0f;
    if (div == 0.0f) return copysignf(0.0f, a / b);
    float fl = floorf(div);
    if (div - fl > 0.5f) fl += 1.0f;
    return fl;
}
/* LeggedRobot._get_env_origins, LR:1234: terrain_types = floor(arange(N) / (N / num_cols)); the divisor is a Python float (fp64 quotient)
 * that torch converts to fp32 for the fp32 result */
static inline int64_t lsim_terrain_type_of_env(int env, int num_envs, int num_cols) {
    float b = (float)((double)num_envs / (double)num_cols);
    int64_t t = (int64_t)lsim_div_floor_f32((float)env, b);
    return t > num_cols - 1 ? num_cols - 1 : t;
}

/* returns 0 on success; shape entries beyond ndim are set to 1 */
static inline int lsim_buffer_desc(const lsim_config* cfg, int id, int64_t shape[4], int* ndim, int* dtype) {
    const int64_t N = cfg->num_envs;
    int64_t s0 = N, s1 = 1, s2 = 1;
    int nd = 1, dt = LSIM_DT_F32;
    switch (id) {
        case LSIM_BUF_OBS: s1 = LSIM_NUM_OBS; nd = 2; break;
        case LSIM_BUF_PRIV_OBS: s1 = LSIM_NUM_PRIV_OBS; nd = 2; break;
        case LSIM_BUF_REW: break;
        case LSIM_BUF_RESET: dt = LSIM_DT_U8; break;
        case LSIM_BUF_TIME_OUT: dt = LSIM_DT_U8; break;
        case LSIM_BUF_EXTRAS_TIME_OUTS: dt = LSIM_DT_U8; break;
        case LSIM_BUF_EPISODE_LENGTH: dt = LSIM_DT_I64; break;
        case LSIM_BUF_ROOT_STATES: s1 = 13; nd = 2; break;
        case LSIM_BUF_DOF_STATE: s1 = LSIM_NUM_DOF; s2 = 2; nd = 3; break;
        case LSIM_BUF_RIGID_BODY_STATES: s1 = LSIM_NUM_BODIES; s2 = 13; nd = 3; break;
        case LSIM_BUF_CONTACT_FORCES: s1 = LSIM_NUM_BODIES; s2 = 3; nd = 3; break;
        case LSIM_BUF_TORQUES:
        case LSIM_BUF_ACTIONS:
        case LSIM_BUF_LAST_ACTIONS:
        case LSIM_BUF_LAST_LAST_ACTIONS:
        case LSIM_BUF_LAST_DOF_POS:
        case LSIM_BUF_LAST_DOF_VEL:
        case LSIM_BUF_LAST_TORQUES:
        case LSIM_BUF_MOTOR_STRENGTH: s1 = LSIM_NUM_DOF; nd = 2; break;
        case LSIM_BUF_LAST_ROOT_VEL: s1 = 6; nd = 2; break;
        case LSIM_BUF_COMMANDS: s1 = 4; nd = 2; break;
        case LSIM_BUF_BASE_LIN_VEL:
        case LSIM_BUF_BASE_ANG_VEL:
        case LSIM_BUF_PROJECTED_GRAVITY:
        case LSIM_BUF_PENDING_FORCE:
        case LSIM_BUF_ENV_ORIGINS:
        case LSIM_BUF_COM_DISPLACEMENT: s1 = 3; nd = 2; break;
        case LSIM_BUF_FEET_AIR_TIME: s1 = 4; nd = 2; break;
        case LSIM_BUF_LAST_CONTACTS:
        case LSIM_BUF_CONTACT_FILT: s1 = 4; nd = 2; dt = LSIM_DT_U8; break;
        case LSIM_BUF_MEASURED_HEIGHTS: s1 = LSIM_NUM_HEIGHT_PTS; nd = 2; break;
        case LSIM_BUF_TERRAIN_LEVELS:
        case LSIM_BUF_TERRAIN_TYPES: dt = LSIM_DT_I64; break;
        case LSIM_BUF_KP_FACTORS:
        case LSIM_BUF_KD_FACTORS:
        case LSIM_BUF_MOTOR_STRENGTH_FACTORS:
        case LSIM_BUF_FRICTION:
        case LSIM_BUF_RESTITUTION:
        case LSIM_BUF_PAYLOAD: break;
        case LSIM_BUF_EPISODE_SUMS: s1 = LSIM_NUM_REWARD_TERMS; nd = 2; break;
        case LSIM_BUF_TERM_PRIV_OBS: s1 = LSIM_NUM_PRIV_OBS; nd = 2; break;
        case LSIM_BUF_TERM_AMP_OBS:
        case LSIM_BUF_AMP_OBS: s1 = LSIM_NUM_AMP_OBS; nd = 2; break;
        case LSIM_BUF_DELAY_STEPS: dt = LSIM_DT_I32; break;
        case LSIM_BUF_CONTACT_COUNT: s1 = 2; nd = 2; dt = LSIM_DT_I32; break;
        case LSIM_BUF_SUBSTEP_TORQUES: s1 = cfg->decimation > 0 ? cfg->decimation : 1; s2 = LSIM_NUM_DOF; nd = 3; break;
        case LSIM_BUF_STATS: s0 = 2; s1 = LSIM_STATS_SIZE; nd = 2; break;
        case LSIM_BUF_NONFINITE: s0 = 2; dt = LSIM_DT_I64; break;
        case LSIM_BUF_HEIGHT_GRID:
            s0 = cfg->grid_rows > 0 ? cfg->grid_rows : 1; s1 = cfg->grid_cols > 0 ? cfg->grid_cols : 1;
            nd = 2; dt = LSIM_DT_I16; break;
        case LSIM_BUF_TERRAIN_MESH:
            s0 = cfg->grid_rows > 0 ? cfg->grid_rows : 1; s1 = cfg->grid_cols > 0 ? cfg->grid_cols : 1;
            nd = 2; dt = LSIM_DT_I32; break;
        case LSIM_BUF_TERRAIN_ORIGINS:
            s0 = cfg->terrain_num_rows > 0 ? cfg->terrain_num_rows : 1;
            s1 = cfg->terrain_num_cols > 0 ? cfg->terrain_num_cols : 1; s2 = 3; nd = 3; break;
        default: return LSIM_E_INVALID;
    }
    shape[0] = s0; shape[1] = s1; shape[2] = s2; shape[3] = 1;
    *ndim = nd; *dtype = dt;
    return LSIM_OK;
}

static inline size_t lsim_buffer_bytes(const lsim_config* cfg, int id) {
    int64_t sh[4]; int nd, dt;
    if (lsim_buffer_desc(cfg, id, sh, &nd, &dt) != LSIM_OK) return 0;
    return (size_t)(sh[0] * sh[1] * sh[2] * sh[3]) * lsim_dtype_size(dt);
}

static const char* const lsim_buffer_names[LSIM_NUM_BUFFERS] = {
    "obs", "priv_obs", "rew", "reset", "time_out", "extras_time_outs", "episode_length", "root_states", "dof_state",
    "rigid_body_states", "contact_forces", "torques", "actions", "last_actions", "last_last_actions", "last_dof_pos",
    "last_dof_vel", "last_torques", "last_root_vel", "commands", "base_lin_vel", "base_ang_vel", "projected_gravity",
    "feet_air_time", "last_contacts", "contact_filt", "measured_heights", "pending_force", "terrain_levels",
    "terrain_types", "env_origins", "kp_factors", "kd_factors", "motor_strength", "motor_strength_factors", "friction",
    "restitution", "payload", "com_displacement", "episode_sums", "term_priv_obs", "term_amp_obs", "amp_obs",
    "delay_steps", "contact_count", "substep_torques", "stats", "nonfinite", "height_grid", "terrain_origins", "terrain_mesh"};

static const char* const lsim_reward_names[LSIM_NUM_REWARD_TERMS] = {
    "action_rate", "ang_vel_xy", "ang_vel_xy_up", "base_height", "base_height_up", "calf_pose", "calf_pose_up",
    "collision", "collision_up", "dof_acc", "dof_pos_dif", "dof_pos_limits", "dof_vel", "dof_vel_limits",
    "feet_air_time", "feet_contact_forces", "feet_mirror", "feet_mirror_up", "feet_slide", "feet_slide_up",
    "feet_stumble", "feet_stumble_up", "foot_clearance_base", "foot_clearance_base_up", "foot_clearance_terrain",
    "foot_clearance_terrain_up", "has_contact", "hip_action_magnitude", "hip_pos", "hip_pos_up", "joint_power",
    "lin_vel_z", "lin_vel_z_up", "orientation", "orientation_up", "power", "power_distribution", "smoothness",
    "stand_nice", "stand_still", "stuck", "termination", "thigh_pose", "thigh_pose_up", "torque_limits", "torques",
    "torques_dif", "torques_distribution", "tracking_ang_vel", "tracking_lin_vel", "upward"};

#ifdef __cplusplus
}
#endif
#endif /* LSIM_LAYOUT_H */
