/*
 * lsim_oracle.c -- TEST INFRASTRUCTURE, not part of the product.
 *
 * Scalar CPU restatement (plain C, one env per loop iteration, fp32, no FMA contraction) of the
 * reference's environment step: LeggedRobot.step() and everything it calls
 * (legged_gym/legged_gym/envs/base/legged_robot.py, "LR" below).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the checker / reported baseline.
 *
 * Pinning: every function marked [pinned] is checked against golden vectors captured by running the
 * reference's own torch code under an isaacgym stub (tools/gen_golden.py -> tests/golden/ npz files).
 * The articulated-body dynamics (orc_physics.c) replace the closed-source PhysX call gym.simulate()
 * (LR:149); no reference source or test exists for them: PARITY UNPINNED for dynamics.
 *
 * Random numbers: the reference draws from torch's global generator; here every draw site is a
 * counter-based Philox stream (lsim.h lsim_rng_tag).  The golden generator injects the same uniforms
 * into the reference so that outputs can be compared value-for-value.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "orc_internal.h"
#include "orc_philox.h"

#define N_DOF LSIM_NUM_DOF
#define N_BODY LSIM_NUM_BODIES
#define N_HP LSIM_NUM_HEIGHT_PTS

/* ------------------------------------------------------------------ helpers */

static float u01(const orc_sim* s, int env, uint32_t stepw, uint32_t tag, uint32_t idx) {
    return orc_u01(s->cfg.seed, s->cfg.rank, (uint32_t)env, stepw, tag, idx);
}
/* isaacgym.torch_utils.torch_rand_float: (upper - lower) * rand + lower, span formed in double (python floats) */
static float rand_range(float u, double lo, double hi) { return (float)(hi - lo) * u + (float)lo; }

static float clipf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* isaacgym.torch_utils.quat_rotate_inverse (xyzw) */
static void quat_rotate_inverse(const float q[4], const float v[3], float out[3]) {
    float w = q[3];
    float s = 2.0f * (w * w) - 1.0f;
    float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
    float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    float a0 = v[0] * s, a1 = v[1] * s, a2 = v[2] * s;
    float b0 = cx * w * 2.0f, b1 = cy * w * 2.0f, b2 = cz * w * 2.0f;
    float c0 = q[0] * d * 2.0f, c1 = q[1] * d * 2.0f, c2 = q[2] * d * 2.0f;
    out[0] = a0 - b0 + c0; out[1] = a1 - b1 + c1; out[2] = a2 - b2 + c2;
}
/* isaacgym.torch_utils.quat_apply */
static void quat_apply(const float q[4], const float v[3], float out[3]) {
    float tx = (q[1] * v[2] - q[2] * v[1]) * 2.0f, ty = (q[2] * v[0] - q[0] * v[2]) * 2.0f, tz = (q[0] * v[1] - q[1] * v[0]) * 2.0f;
    float ux = q[1] * tz - q[2] * ty, uy = q[2] * tx - q[0] * tz, uz = q[0] * ty - q[1] * tx;
    out[0] = v[0] + q[3] * tx + ux; out[1] = v[1] + q[3] * ty + uy; out[2] = v[2] + q[3] * tz + uz;
}
/* legged_gym.utils.math.quat_apply_yaw (MTH:38-42) */
static void quat_apply_yaw(const float q[4], const float v[3], float out[3]) {
    float qy[4] = {0.0f, 0.0f, q[2], q[3]};
    float n = sqrtf(qy[2] * qy[2] + qy[3] * qy[3]);
    if (n < 1e-9f) n = 1e-9f;
    qy[2] /= n; qy[3] /= n;
    quat_apply(qy, v, out);
}
/* legged_gym.utils.math.wrap_to_pi (MTH:45-48); torch.remainder semantics */
static float wrap_to_pi(float a) {
    const float two_pi = (float)(2.0 * M_PI);
    float r = fmodf(a, two_pi);
    if (r != 0.0f && r < 0.0f) r += two_pi;
    if (r > (float)M_PI) r -= two_pi;
    return r;
}
/* isaacgym.torch_utils.quat_from_euler_xyz */
static void quat_from_euler_xyz(float roll, float pitch, float yaw, float q[4]) {
    float cy = cosf(yaw * 0.5f), sy = sinf(yaw * 0.5f), cr = cosf(roll * 0.5f), sr = sinf(roll * 0.5f);
    float cp = cosf(pitch * 0.5f), sp = sinf(pitch * 0.5f);
    q[3] = cy * cr * cp + sy * sr * sp;
    q[0] = cy * sr * cp - sy * cr * sp;
    q[1] = cy * cr * sp + sy * sr * cp;
    q[2] = sy * cr * cp - cy * sr * sp;
}
static float norm2(float a, float b) { return sqrtf(a * a + b * b); }
static float norm3(const float* v) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* ------------------------------------------------------------------ E8/E9: height sampling [pinned] */

/* LR:1342-1355: (x + border) / hscale truncated toward zero, clip, min of 3 samples, * vscale */
static float sample_height_min3(const orc_sim* s, float x, float y) {
    const lsim_config* c = &s->cfg;
    float fx = (x + c->border_size) / c->horizontal_scale;
    float fy = (y + c->border_size) / c->horizontal_scale;
    long px = (long)fx, py = (long)fy; /* .long() truncation */
    if (px < 0) px = 0; if (px > c->grid_rows - 2) px = c->grid_rows - 2;
    if (py < 0) py = 0; if (py > c->grid_cols - 2) py = c->grid_cols - 2;
    const int16_t* g = ORC_I16(s, LSIM_BUF_HEIGHT_GRID);
    int16_t h1 = g[px * c->grid_cols + py], h2 = g[(px + 1) * c->grid_cols + py], h3 = g[px * c->grid_cols + py + 1];
    int16_t h = h1 < h2 ? h1 : h2;
    h = h < h3 ? h : h3;
    return (float)h * c->vertical_scale;
}

/* LeggedRobot._get_heights (LR:1318-1355) for one env */
static void get_heights(const orc_sim* s, int e, float* out /*187*/) {
    const lsim_config* c = &s->cfg;
    if (c->mesh_type == 0) { for (int i = 0; i < N_HP; ++i) out[i] = 0.0f; return; }
    const float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
    for (int ix = 0; ix < c->num_points_x; ++ix)
        for (int iy = 0; iy < c->num_points_y; ++iy) {
            float p[3] = {c->measured_points_x[ix], c->measured_points_y[iy], 0.0f}, w[3];
            quat_apply_yaw(root + 3, p, w);
            out[ix * c->num_points_y + iy] = sample_height_min3(s, w[0] + root[0], w[1] + root[1]);
        }
}

/* LeggedRobot._get_base_heights (LR:1357-1398): mean over 7x9 points of (z - h) */
static float get_base_height(const orc_sim* s, int e) {
    static const float xs[7] = {-0.15f, -0.1f, -0.05f, 0.f, 0.05f, 0.1f, 0.15f};
    static const float ys[9] = {-0.2f, -0.15f, -0.1f, -0.05f, 0.f, 0.05f, 0.1f, 0.15f, 0.2f};
    const float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
    if (s->cfg.mesh_type == 0) return root[2];
    float acc = 0.0f;
    for (int ix = 0; ix < 7; ++ix)
        for (int iy = 0; iy < 9; ++iy) {
            float p[3] = {xs[ix], ys[iy], 0.0f}, w[3];
            quat_apply_yaw(root + 3, p, w);
            acc += root[2] - sample_height_min3(s, w[0] + root[0], w[1] + root[1]);
        }
    return acc / 63.0f;
}

/* ------------------------------------------------------------------ E3: PD torques [pinned] */

/* LeggedRobot._compute_torques (LR:658-688) */
static void compute_torques(const orc_sim* s, int e, const float act[N_DOF], float tau[N_DOF]) {
    const lsim_config* c = &s->cfg;
    const float* ms = ORC_F(s, LSIM_BUF_MOTOR_STRENGTH) + N_DOF * e;
    const float* dof = ORC_F(s, LSIM_BUF_DOF_STATE) + 2 * N_DOF * e;
    const float* last_vel = ORC_F(s, LSIM_BUF_LAST_DOF_VEL) + N_DOF * e;
    float kpf = ORC_F(s, LSIM_BUF_KP_FACTORS)[e], kdf = ORC_F(s, LSIM_BUF_KD_FACTORS)[e];
    for (int j = 0; j < N_DOF; ++j) {
        float a = ms[j] * act[j];
        float as = a * c->action_scale;
        if (j % 3 == 0) as *= c->hip_reduction;
        float target = c->default_dof_pos[j] + as;
        float q = dof[2 * j], qd = dof[2 * j + 1], t;
        if (c->control_type == 0)
            t = c->p_gains[j] * kpf * (target - q) - c->d_gains[j] * kdf * qd;
        else if (c->control_type == 1)
            t = c->p_gains[j] * (as - qd) - c->d_gains[j] * (qd - last_vel[j]) / c->sim_dt;
        else
            t = as;
        tau[j] = clipf(t, -c->torque_limits[j], c->torque_limits[j]);
    }
}

/* ------------------------------------------------------------------ E7: command resampling [pinned] */

/* LeggedRobot._resample_commands (LR:634-656) for one env */
static void resample_commands(orc_sim* s, int e, uint32_t stepw, uint32_t tag) {
    const lsim_config* c = &s->cfg;
    float* cmd = ORC_F(s, LSIM_BUF_COMMANDS) + 4 * e;
    cmd[0] = rand_range(u01(s, e, stepw, tag, 0), -1.0, 1.0);
    cmd[1] = rand_range(u01(s, e, stepw, tag, 1), s->command_ranges[1][0], s->command_ranges[1][1]);
    if (c->heading_command)
        cmd[3] = rand_range(u01(s, e, stepw, tag, 2), s->command_ranges[3][0], s->command_ranges[3][1]);
    else
        cmd[2] = rand_range(u01(s, e, stepw, tag, 2), s->command_ranges[2][0], s->command_ranges[2][1]);
    if ((float)e < (float)((double)c->num_envs * 0.2)) { /* LR:649 env_ids < num_envs * 0.2 */
        cmd[0] = rand_range(u01(s, e, stepw, tag, 3), s->command_ranges[0][0], s->command_ranges[0][1]);
        cmd[1] *= (fabsf(cmd[0]) < 1.0f) ? 1.0f : 0.0f; /* norm of a 1-vector */
    }
    float m = (norm2(cmd[0], cmd[1]) > 0.2f) ? 1.0f : 0.0f;
    cmd[0] *= m; cmd[1] *= m;
}

/* ------------------------------------------------------------------ E13/E14: observation vector [pinned] */

/* the 238-vector of compute_observations / compute_termination_observations (LR:382-404, LR:439-460) */
static void build_obs238(const orc_sim* s, int e, uint32_t stepw, uint32_t tag, const float disturbance[3], float* o) {
    const lsim_config* c = &s->cfg;
    const float* cmd = ORC_F(s, LSIM_BUF_COMMANDS) + 4 * e;
    const float* av = ORC_F(s, LSIM_BUF_BASE_ANG_VEL) + 3 * e;
    const float* lv = ORC_F(s, LSIM_BUF_BASE_LIN_VEL) + 3 * e;
    const float* pg = ORC_F(s, LSIM_BUF_PROJECTED_GRAVITY) + 3 * e;
    const float* dof = ORC_F(s, LSIM_BUF_DOF_STATE) + 2 * N_DOF * e;
    const float* act = ORC_F(s, LSIM_BUF_ACTIONS) + N_DOF * e;
    const float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
    const float* mh = ORC_F(s, LSIM_BUF_MEASURED_HEIGHTS) + N_HP * e;
    o[0] = cmd[0] * c->obs_scale_lin_vel; o[1] = cmd[1] * c->obs_scale_lin_vel; o[2] = cmd[2] * c->obs_scale_ang_vel;
    for (int k = 0; k < 3; ++k) o[3 + k] = av[k] * c->obs_scale_ang_vel;
    for (int k = 0; k < 3; ++k) o[6 + k] = pg[k];
    for (int j = 0; j < N_DOF; ++j) o[9 + j] = (dof[2 * j] - c->default_dof_pos[j]) * c->obs_scale_dof_pos;
    for (int j = 0; j < N_DOF; ++j) o[21 + j] = dof[2 * j + 1] * c->obs_scale_dof_vel;
    for (int j = 0; j < N_DOF; ++j) o[33 + j] = act[j];
    if (c->add_noise) {
        for (int k = 0; k < 45; ++k) {
            float nv = 0.0f;
            if (k >= 3 && k < 6) nv = c->noise_vec_ang_vel;
            else if (k >= 6 && k < 9) nv = c->noise_vec_gravity;
            else if (k >= 9 && k < 21) nv = c->noise_vec_dof_pos;
            else if (k >= 21 && k < 33) nv = c->noise_vec_dof_vel;
            o[k] += (2.0f * u01(s, e, stepw, tag, (uint32_t)k) - 1.0f) * nv;
        }
    }
    for (int k = 0; k < 3; ++k) o[45 + k] = lv[k] * c->obs_scale_lin_vel;
    for (int k = 0; k < 3; ++k) o[48 + k] = disturbance[k];
    if (c->measure_heights) {
        for (int k = 0; k < N_HP; ++k) {
            float h = clipf(root[2] - 0.5f - mh[k], -1.0f, 1.0f) * c->obs_scale_height;
            /* LR:400 adds the noise term unconditionally (not gated by add_noise) */
            h += (2.0f * u01(s, e, stepw, tag, (uint32_t)(45 + k)) - 1.0f) * c->noise_vec_height;
            o[51 + k] = h;
        }
    }
}

/* ------------------------------------------------------------------ R: reward terms [pinned] */

typedef struct step_ctx { /* per-env quantities shared by the reward terms */
    const float *root, *dof, *cf, *body, *cmd, *act, *last_act, *last_last_act, *last_dof_pos, *last_dof_vel, *tau, *last_tau;
    const float *lin_vel, *ang_vel, *grav;
    float feet_pos[4][3];   /* copy: quirk 3 (LR:1722-1725) mutates it in place */
    const uint8_t* contact_filt;
    int e;
} step_ctx;

static float up_factor(const step_ctx* x) { return clipf(-x->grav[2], 0.0f, 1.0f); }

static float foot_slide_like(const orc_sim* s, const step_ctx* x, int with_height) {
    /* LR:1610-1619 (with_height=0) and LR:1682-1698 (with_height=1) */
    float acc = 0.0f;
    for (int f = 0; f < 4; ++f) {
        int b = s->model.feet_bodies[f];
        const float* bs = x->body + 13 * b;
        float dv[3] = {bs[7] - x->root[7], bs[8] - x->root[8], bs[9] - x->root[9]}, vb[3];
        quat_rotate_inverse(x->root + 3, dv, vb);
        float lat = sqrtf(vb[0] * vb[0] + vb[1] * vb[1]);
        if (with_height) {
            float dp[3] = {x->feet_pos[f][0] - x->root[0], x->feet_pos[f][1] - x->root[1], x->feet_pos[f][2] - x->root[2]}, pb[3];
            quat_rotate_inverse(x->root + 3, dp, pb);
            float he = pb[2] - s->cfg.foot_height_target_base;
            acc += (he * he) * lat;
        } else {
            acc += (x->contact_filt[f] ? 1.0f : 0.0f) * lat;
        }
    }
    return acc;
}

static float foot_clearance_terrain(const orc_sim* s, step_ctx* x) {
    /* LR:1717-1743, including the in-place `points += border_size` on self.feet_pos (quirk 3) */
    const lsim_config* c = &s->cfg;
    float acc = 0.0f;
    for (int f = 0; f < 4; ++f) {
        int b = s->model.feet_bodies[f];
        const float* bs = x->body + 13 * b;
        float fh;
        if (c->mesh_type == 0) {
            fh = x->feet_pos[f][2];
        } else {
            for (int k = 0; k < 3; ++k) x->feet_pos[f][k] += c->border_size;
            float fx = x->feet_pos[f][0] / c->horizontal_scale, fy = x->feet_pos[f][1] / c->horizontal_scale;
            long px = (long)fx, py = (long)fy;
            if (px < 0) px = 0; if (px > c->grid_rows - 2) px = c->grid_rows - 2;
            if (py < 0) py = 0; if (py > c->grid_cols - 2) py = c->grid_cols - 2;
            const int16_t* g = ORC_I16(s, LSIM_BUF_HEIGHT_GRID);
            int16_t h1 = g[px * c->grid_cols + py], h2 = g[(px + 1) * c->grid_cols + py], h3 = g[px * c->grid_cols + py + 1];
            int16_t h = h1 < h2 ? h1 : h2; h = h < h3 ? h : h3;
            fh = x->feet_pos[f][2] - (float)h * c->vertical_scale;
        }
        float lat = sqrtf(bs[7] * bs[7] + bs[8] * bs[8]);
        float d = fh - c->foot_height_target_terrain;
        acc += lat * (d * d);
    }
    return acc;
}

static float sum_abs_dev(const orc_sim* s, const step_ctx* x, int first) { /* joints first, first+3, ... */
    float acc = 0.0f;
    for (int l = 0; l < 4; ++l) { int j = 3 * l + first; acc += fabsf(x->dof[2 * j] - s->cfg.default_dof_pos[j]); }
    return acc;
}

static float stumble(const orc_sim* s, const step_ctx* x, float ratio) { /* LR:1589-1608 */
    const lsim_config* c = &s->cfg;
    int any = 0;
    for (int f = 0; f < 4; ++f) {
        const float* F = x->cf + 3 * s->model.feet_bodies[f];
        if (norm2(F[0], F[1]) > ratio * fabsf(F[2])) any = 1;
    }
    float r = (any && ORC_I64(s, LSIM_BUF_TERRAIN_LEVELS)[x->e] > 3) ? 1.0f : 0.0f;
    int in_slice = (x->e >= c->stairsup_start_idx && x->e < c->stairsup_end_idx) || (x->e >= c->pit_start_idx && x->e < c->gap_end_idx);
    return in_slice ? r : 0.0f;
}

static float variance12(const float v[N_DOF]) { /* torch.var, unbiased */
    float m = 0.0f;
    for (int j = 0; j < N_DOF; ++j) m += v[j];
    m /= 12.0f;
    float a = 0.0f;
    for (int j = 0; j < N_DOF; ++j) a += (v[j] - m) * (v[j] - m);
    return a / 11.0f;
}

/* one `_reward_<name>()` value (LR:1444-1770) for env x->e */
static float reward_term(orc_sim* s, step_ctx* x, int id) {
    const lsim_config* c = &s->cfg;
    const float dt = c->sim_dt * (float)c->decimation; /* self.dt, LR:1253 */
    float acc = 0.0f;
    switch (id) {
        case LSIM_R_TRACKING_LIN_VEL: { /* LR:1444-1452 */
            float small = norm2(x->cmd[0], x->cmd[1]) < 0.1f ? 0.0f : 1.0f;
            float ex = x->cmd[0] * small - x->lin_vel[0], ey = x->cmd[1] * small - x->lin_vel[1];
            return expf(-(ex * ex + ey * ey) / c->tracking_sigma);
        }
        case LSIM_R_TRACKING_ANG_VEL: { float e = x->cmd[2] - x->ang_vel[2]; return expf(-(e * e) / c->tracking_sigma); }
        case LSIM_R_FEET_AIR_TIME: { /* LR:1459-1470; mutates last_contacts and feet_air_time */
            float* air = ORC_F(s, LSIM_BUF_FEET_AIR_TIME) + 4 * x->e;
            uint8_t* lc = ORC_U8(s, LSIM_BUF_LAST_CONTACTS) + 4 * x->e;
            float r = 0.0f;
            uint8_t filt[4];
            for (int f = 0; f < 4; ++f) {
                uint8_t contact = x->cf[3 * s->model.feet_bodies[f] + 2] > 1.0f;
                filt[f] = contact | lc[f];
                lc[f] = contact;
                float first = (air[f] > 0.0f && filt[f]) ? 1.0f : 0.0f;
                air[f] += dt;
                r += (air[f] - 0.5f) * first;
            }
            r *= (norm2(x->cmd[0], x->cmd[1]) > 0.1f) ? 1.0f : 0.0f;
            for (int f = 0; f < 4; ++f) air[f] *= filt[f] ? 0.0f : 1.0f;
            return r;
        }
        case LSIM_R_UPWARD: return 1.0f - x->grav[2];
        case LSIM_R_HAS_CONTACT: {
            float n = 0.0f;
            for (int f = 0; f < 4; ++f) n += x->contact_filt[f] ? 1.0f : 0.0f;
            return ((norm2(x->cmd[0], x->cmd[1]) < 0.1f) ? 1.0f : 0.0f) * n / 4.0f;
        }
        case LSIM_R_LIN_VEL_Z: return x->lin_vel[2] * x->lin_vel[2];
        case LSIM_R_LIN_VEL_Z_UP: return x->lin_vel[2] * x->lin_vel[2] * up_factor(x);
        case LSIM_R_ANG_VEL_XY: return x->ang_vel[0] * x->ang_vel[0] + x->ang_vel[1] * x->ang_vel[1];
        case LSIM_R_ANG_VEL_XY_UP: return (x->ang_vel[0] * x->ang_vel[0] + x->ang_vel[1] * x->ang_vel[1]) * up_factor(x);
        case LSIM_R_ORIENTATION: return x->grav[0] * x->grav[0] + x->grav[1] * x->grav[1];
        case LSIM_R_ORIENTATION_UP: return (x->grav[0] * x->grav[0] + x->grav[1] * x->grav[1]) * up_factor(x);
        case LSIM_R_BASE_HEIGHT: { float d = get_base_height(s, x->e) - c->base_height_target; return d * d; }
        case LSIM_R_BASE_HEIGHT_UP: { float d = get_base_height(s, x->e) - c->base_height_target; return d * d * up_factor(x); }
        case LSIM_R_DOF_VEL: for (int j = 0; j < N_DOF; ++j) acc += x->dof[2 * j + 1] * x->dof[2 * j + 1]; return acc;
        case LSIM_R_DOF_ACC:
            for (int j = 0; j < N_DOF; ++j) { float a = (x->last_dof_vel[j] - x->dof[2 * j + 1]) / dt; acc += a * a; }
            return acc;
        case LSIM_R_DOF_VEL_LIMITS:
            for (int j = 0; j < N_DOF; ++j)
                acc += clipf(fabsf(x->dof[2 * j + 1]) - s->model.dof_vel_limit[j] * c->soft_dof_vel_limit, 0.0f, 1.0f);
            return acc;
        case LSIM_R_DOF_POS_DIF:
            for (int j = 0; j < N_DOF; ++j) { float d = x->last_dof_pos[j] - x->dof[2 * j]; acc += d * d; }
            return acc;
        case LSIM_R_DOF_POS_LIMITS: /* soft limits, LR:574-578 */
            for (int j = 0; j < N_DOF; ++j) {
                float lo = s->model.dof_pos_lower[j], hi = s->model.dof_pos_upper[j];
                float m = (lo + hi) / 2.0f, r = hi - lo;
                float slo = m - 0.5f * r * c->soft_dof_pos_limit, shi = m + 0.5f * r * c->soft_dof_pos_limit;
                float q = x->dof[2 * j];
                float o = -fminf(q - slo, 0.0f);
                o += fmaxf(q - shi, 0.0f);
                acc += o;
            }
            return acc;
        case LSIM_R_ACTION_RATE:
            for (int j = 0; j < N_DOF; ++j) { float d = x->last_act[j] - x->act[j]; acc += d * d; }
            return acc;
        case LSIM_R_SMOOTHNESS:
            for (int j = 0; j < N_DOF; ++j) { float d = x->act[j] - x->last_act[j] - x->last_act[j] + x->last_last_act[j]; acc += d * d; }
            return acc;
        case LSIM_R_TORQUES: for (int j = 0; j < N_DOF; ++j) acc += x->tau[j] * x->tau[j]; return acc;
        case LSIM_R_TORQUES_DISTRIBUTION: { float v[N_DOF]; for (int j = 0; j < N_DOF; ++j) v[j] = fabsf(x->tau[j]); return variance12(v); }
        case LSIM_R_TORQUES_DIF: for (int j = 0; j < N_DOF; ++j) { float d = x->tau[j] - x->last_tau[j]; acc += d * d; } return acc;
        case LSIM_R_TORQUE_LIMITS:
            for (int j = 0; j < N_DOF; ++j) acc += fmaxf(fabsf(x->tau[j]) - c->torque_limits[j] * c->soft_torque_limit, 0.0f);
            return acc;
        case LSIM_R_JOINT_POWER: for (int j = 0; j < N_DOF; ++j) acc += fabsf(x->dof[2 * j + 1]) * fabsf(x->tau[j]); return acc;
        case LSIM_R_POWER: for (int j = 0; j < N_DOF; ++j) acc += fabsf(x->tau[j] * x->dof[2 * j + 1]); return acc;
        case LSIM_R_POWER_DISTRIBUTION: { float v[N_DOF]; for (int j = 0; j < N_DOF; ++j) v[j] = fabsf(x->tau[j] * x->dof[2 * j + 1]); return variance12(v); }
        case LSIM_R_COLLISION:
        case LSIM_R_COLLISION_UP: /* LR:1573-1578 */
            for (int b = 0; b < N_BODY; ++b)
                if ((s->model.penalised_body_mask >> b) & 1u) acc += (norm3(x->cf + 3 * b) > 0.1f) ? 1.0f : 0.0f;
            return id == LSIM_R_COLLISION ? acc : acc * up_factor(x);
        case LSIM_R_TERMINATION: /* LR:1580-1582 */
            return (ORC_U8(s, LSIM_BUF_RESET)[x->e] && !ORC_U8(s, LSIM_BUF_TIME_OUT)[x->e]) ? 1.0f : 0.0f;
        case LSIM_R_FEET_CONTACT_FORCES: /* LR:1628-1630 */
            for (int f = 0; f < 4; ++f) acc += fmaxf(norm3(x->cf + 3 * s->model.feet_bodies[f]) - c->max_contact_force, 0.0f);
            return acc;
        case LSIM_R_FEET_STUMBLE: return stumble(s, x, 5.0f);
        case LSIM_R_FEET_STUMBLE_UP: return stumble(s, x, 4.0f) * up_factor(x);
        case LSIM_R_FEET_SLIDE: return foot_slide_like(s, x, 0);
        case LSIM_R_FEET_SLIDE_UP: return foot_slide_like(s, x, 0) * up_factor(x);
        case LSIM_R_FEET_MIRROR:
        case LSIM_R_FEET_MIRROR_UP: { /* LR:1632-1640 */
            const float* d = x->dof;
            float a1 = d[2 * 1] - d[2 * 10], a2 = d[2 * 2] - d[2 * 11], b1 = d[2 * 4] - d[2 * 7], b2 = d[2 * 5] - d[2 * 8];
            float r = 0.5f * ((a1 * a1 + a2 * a2) + (b1 * b1 + b2 * b2));
            return id == LSIM_R_FEET_MIRROR ? r : r * up_factor(x);
        }
        case LSIM_R_STAND_STILL:
        case LSIM_R_STAND_NICE: { /* LR:1643-1649 */
            for (int j = 0; j < N_DOF; ++j) acc += fabsf(x->dof[2 * j] - c->default_dof_pos[j]);
            acc *= (norm2(x->cmd[0], x->cmd[1]) < 0.1f) ? 1.0f : 0.0f;
            return id == LSIM_R_STAND_STILL ? acc : acc * (1.0f - x->grav[2]);
        }
        case LSIM_R_STUCK: return ((fabsf(x->lin_vel[0]) < 0.1f) && (fabsf(x->cmd[0]) > 0.1f)) ? 1.0f : 0.0f;
        case LSIM_R_HIP_ACTION_MAGNITUDE:
            for (int l = 0; l < 4; ++l) { float m = fmaxf(fabsf(x->act[3 * l]) - 1.0f, 0.0f); acc += m * m; }
            return acc;
        case LSIM_R_HIP_POS: return sum_abs_dev(s, x, 0);
        case LSIM_R_HIP_POS_UP: return sum_abs_dev(s, x, 0) * up_factor(x);
        case LSIM_R_THIGH_POSE: return sum_abs_dev(s, x, 1);
        case LSIM_R_THIGH_POSE_UP: return sum_abs_dev(s, x, 1) * up_factor(x);
        case LSIM_R_CALF_POSE: return sum_abs_dev(s, x, 2);
        case LSIM_R_CALF_POSE_UP: return sum_abs_dev(s, x, 2) * up_factor(x);
        case LSIM_R_FOOT_CLEARANCE_BASE: return foot_slide_like(s, x, 1);
        case LSIM_R_FOOT_CLEARANCE_BASE_UP: return foot_slide_like(s, x, 1) * up_factor(x);
        case LSIM_R_FOOT_CLEARANCE_TERRAIN: return foot_clearance_terrain(s, x);
        case LSIM_R_FOOT_CLEARANCE_TERRAIN_UP: return foot_clearance_terrain(s, x) * up_factor(x);
        default: return 0.0f;
    }
}

/* ------------------------------------------------------------------ E17/E18: resets [pinned] */

static void reset_dofs(orc_sim* s, int e, uint32_t stepw) { /* LR:690-716 */
    const lsim_config* c = &s->cfg;
    float* dof = ORC_F(s, LSIM_BUF_DOF_STATE) + 2 * N_DOF * e;
    for (int j = 0; j < N_DOF; ++j) {
        if (c->has_dof_init_pos_ratio)
            dof[2 * j] = c->default_dof_pos[j] * rand_range(u01(s, e, stepw, LSIM_RNG_RESET_DOF, (uint32_t)j),
                                                            c->dof_init_pos_ratio_range[0], c->dof_init_pos_ratio_range[1]);
        else
            dof[2 * j] = c->default_dof_pos[j];
        if (c->randomize_dof_vel) { /* rand_like * |hi - lo| + min(lo, hi), LR:709 */
            float lo = c->dof_init_vel_range[0], hi = c->dof_init_vel_range[1];
            dof[2 * j + 1] = u01(s, e, stepw, LSIM_RNG_RESET_DOF, (uint32_t)(12 + j)) * fabsf(hi - lo) + fminf(lo, hi);
        } else
            dof[2 * j + 1] = 0.0f;
    }
}

static void reset_root_states(orc_sim* s, int e, uint32_t stepw) { /* LR:718-820 (custom_origins branch: terrain meshes) */
    const lsim_config* c = &s->cfg;
    float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
    const float* org = ORC_F(s, LSIM_BUF_ENV_ORIGINS) + 3 * e;
    for (int k = 0; k < 13; ++k) root[k] = c->base_init_state[k];
    for (int k = 0; k < 3; ++k) root[k] += org[k];
    if (c->mesh_type != 0) {
        if (c->has_base_init_pos_range) {
            for (int k = 0; k < 3; ++k)
                root[k] += rand_range(u01(s, e, stepw, LSIM_RNG_RESET_ROOT, (uint32_t)k), c->base_init_pos_range[k][0], c->base_init_pos_range[k][1]);
        } else {
            for (int k = 0; k < 2; ++k) root[k] += rand_range(u01(s, e, stepw, LSIM_RNG_RESET_ROOT, (uint32_t)k), -1.0, 1.0);
        }
    }
    if (c->has_base_init_rot_range) {
        float rpy[3];
        for (int k = 0; k < 3; ++k)
            rpy[k] = rand_range(u01(s, e, stepw, LSIM_RNG_RESET_ROOT, (uint32_t)(3 + k)), c->base_init_rot_range[k][0], c->base_init_rot_range[k][1]);
        quat_from_euler_xyz(rpy[0], rpy[1], rpy[2], root + 3);
    }
    for (int k = 0; k < 6; ++k)
        root[7 + k] = rand_range(u01(s, e, stepw, LSIM_RNG_RESET_ROOT, (uint32_t)(6 + k)), c->base_init_vel_range[k][0], c->base_init_vel_range[k][1]);
}

static void update_terrain_curriculum(orc_sim* s, int e, uint32_t stepw) { /* LR:846-866 */
    const lsim_config* c = &s->cfg;
    if (!s->init_done) return;
    const float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
    float* org = ORC_F(s, LSIM_BUF_ENV_ORIGINS) + 3 * e;
    const float* cmd = ORC_F(s, LSIM_BUF_COMMANDS) + 4 * e;
    int64_t* lvl = ORC_I64(s, LSIM_BUF_TERRAIN_LEVELS) + e;
    int64_t type = ORC_I64(s, LSIM_BUF_TERRAIN_TYPES)[e];
    float dist = norm2(root[0] - org[0], root[1] - org[1]);
    int up = dist > c->terrain_length / 2.0f;
    int down = (dist < norm2(cmd[0], cmd[1]) * c->episode_length_s * 0.5f) && !up;
    *lvl += (int64_t)up - (int64_t)down;
    if (*lvl >= c->terrain_num_rows)
        *lvl = (int64_t)(u01(s, e, stepw, LSIM_RNG_RESET_LEVEL, 0) * (float)c->terrain_num_rows);
    else if (*lvl < 0)
        *lvl = 0;
    const float* to = ORC_F(s, LSIM_BUF_TERRAIN_ORIGINS) + ((*lvl) * c->terrain_num_cols + type) * 3;
    for (int k = 0; k < 3; ++k) org[k] = to[k];
}

/* ------------------------------------------------------------------ the step */

static float* stats_row(orc_sim* s) { return ORC_F(s, LSIM_BUF_STATS) + s->stats_row * LSIM_STATS_SIZE; }

static void refresh_stats_ranges(orc_sim* s) {
    float* st = stats_row(s);
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 2; ++k) st[LSIM_STATS_CMD_RANGES + 2 * i + k] = (float)s->command_ranges[i][k];
}

/* reset_idx(env_ids) (LR:288-361) for the envs flagged in `mask`; n_reset = len(env_ids) */
static void reset_idx(orc_sim* s, const uint8_t* mask, int n_reset, uint32_t stepw) {
    const lsim_config* c = &s->cfg;
    const int N = c->num_envs;
    float* st = stats_row(s);
    if (n_reset == 0) return; /* LR:298 */
    if (c->terrain_curriculum && c->mesh_type != 0)
        for (int e = 0; e < N; ++e) if (mask[e]) update_terrain_curriculum(s, e, stepw);
    /* LR:307-308 + LR:868-880: command curriculum (global over the reset set) */
    if (c->commands_curriculum && (s->step_counter % c->max_episode_length == 0)) {
        float acc = 0.0f;
        for (int e = 0; e < N; ++e) if (mask[e]) acc += ORC_F(s, LSIM_BUF_EPISODE_SUMS)[e * LSIM_NUM_REWARD_TERMS + LSIM_R_TRACKING_LIN_VEL];
        float mean = acc / (float)n_reset;
        if (mean / (float)c->max_episode_length > 0.8f * c->reward_scales[LSIM_R_TRACKING_LIN_VEL]) {
            double(*r)[2] = s->command_ranges;
            r[0][0] = fmax(fmin(r[0][0] - 0.1, 0.0), -(double)c->max_backward_curriculum);
            r[0][1] = fmax(fmin(r[0][1] + 0.1, (double)c->max_forward_curriculum), 0.0);
            r[1][0] = fmax(fmin(r[1][0] - 0.1, 0.0), -(double)c->max_lat_curriculum);
            r[1][1] = fmax(fmin(r[1][1] + 0.1, (double)c->max_lat_curriculum), 0.0);
        }
    }
    for (int e = 0; e < N; ++e) {
        if (!mask[e]) continue;
        reset_dofs(s, e, stepw);
        reset_root_states(s, e, stepw);
        resample_commands(s, e, stepw, LSIM_RNG_RESET_CMD);
        for (int j = 0; j < N_DOF; ++j) {
            ORC_F(s, LSIM_BUF_LAST_ACTIONS)[N_DOF * e + j] = 0.0f;
            ORC_F(s, LSIM_BUF_LAST_LAST_ACTIONS)[N_DOF * e + j] = 0.0f;
            ORC_F(s, LSIM_BUF_LAST_DOF_POS)[N_DOF * e + j] = 0.0f;
            ORC_F(s, LSIM_BUF_LAST_DOF_VEL)[N_DOF * e + j] = 0.0f;
            ORC_F(s, LSIM_BUF_LAST_TORQUES)[N_DOF * e + j] = 0.0f;
        }
        for (int f = 0; f < 4; ++f) ORC_F(s, LSIM_BUF_FEET_AIR_TIME)[4 * e + f] = 0.0f;
        ORC_U8(s, LSIM_BUF_RESET)[e] = 1;
    }
    if (c->measure_heights) { /* LR:332-333: all envs */
        #pragma omp parallel for schedule(static)
        for (int e = 0; e < N; ++e) get_heights(s, e, ORC_F(s, LSIM_BUF_MEASURED_HEIGHTS) + N_HP * e);
    }
    for (int e = 0; e < N; ++e) {
        if (!mask[e]) continue;
        if (c->randomize_kp) ORC_F(s, LSIM_BUF_KP_FACTORS)[e] = rand_range(u01(s, e, stepw, LSIM_RNG_RESET_DR, 0), c->kp_range[0], c->kp_range[1]);
        if (c->randomize_kd) ORC_F(s, LSIM_BUF_KD_FACTORS)[e] = rand_range(u01(s, e, stepw, LSIM_RNG_RESET_DR, 1), c->kd_range[0], c->kd_range[1]);
        if (c->randomize_motor_strength)
            ORC_F(s, LSIM_BUF_MOTOR_STRENGTH_FACTORS)[e] = rand_range(u01(s, e, stepw, LSIM_RNG_RESET_DR, 2), c->motor_strength_range[0], c->motor_strength_range[1]);
        if (c->randomize_friction) ORC_F(s, LSIM_BUF_FRICTION)[e] = rand_range(u01(s, e, stepw, LSIM_RNG_RESET_DR, 3), c->friction_range[0], c->friction_range[1]);
        if (c->randomize_restitution) ORC_F(s, LSIM_BUF_RESTITUTION)[e] = rand_range(u01(s, e, stepw, LSIM_RNG_RESET_DR, 4), c->restitution_range[0], c->restitution_range[1]);
    }
    /* LR:346-356 episode statistics (sums; the host divides by the count and dt) */
    st[LSIM_STATS_RESET_COUNT] = (float)n_reset;
    st[LSIM_STATS_RESET_STEPS] += 1.0f;
    for (int k = 0; k < LSIM_NUM_REWARD_TERMS; ++k) st[LSIM_STATS_EPISODE_SUMS + k] = 0.0f;
    for (int e = 0; e < N; ++e) {
        if (!mask[e]) continue;
        int64_t len = ORC_I64(s, LSIM_BUF_EPISODE_LENGTH)[e];
        float den = (float)(len < 1 ? 1 : len);
        for (int k = 0; k < LSIM_NUM_REWARD_TERMS; ++k) {
            float* es = ORC_F(s, LSIM_BUF_EPISODE_SUMS) + e * LSIM_NUM_REWARD_TERMS + k;
            st[LSIM_STATS_EPISODE_SUMS + k] += *es / den;
            *es = 0.0f;
        }
    }
    float lsum = 0.0f;
    for (int e = 0; e < N; ++e) lsum += (float)ORC_I64(s, LSIM_BUF_TERRAIN_LEVELS)[e];
    st[LSIM_STATS_LEVEL_SUM] = lsum;
    refresh_stats_ranges(s);
    if (c->send_timeouts) memcpy(ORC_U8(s, LSIM_BUF_EXTRAS_TIME_OUTS), ORC_U8(s, LSIM_BUF_TIME_OUT), (size_t)N);
    for (int e = 0; e < N; ++e) if (mask[e]) ORC_I64(s, LSIM_BUF_EPISODE_LENGTH)[e] = 0;
}

/* LeggedRobot.post_physics_step (LR:178-247) */
static void post_physics_step(orc_sim* s, uint32_t flags) {
    const lsim_config* c = &s->cfg;
    const int N = c->num_envs;
    const float dt = c->sim_dt * (float)c->decimation;
    (void)dt;
    s->step_counter += 1;
    const uint32_t stepw = (uint32_t)s->step_counter;
    float* st = stats_row(s);
    st[LSIM_STATS_RESET_COUNT] = 0.0f;
    refresh_stats_ranges(s);
    uint8_t* reset = ORC_U8(s, LSIM_BUF_RESET);
    uint8_t* tout = ORC_U8(s, LSIM_BUF_TIME_OUT);
    int n_reset = 0;
    const float gvec[3] = {0.0f, 0.0f, -1.0f};
    const float fwd[3] = {1.0f, 0.0f, 0.0f};

    #pragma omp parallel for schedule(static) reduction(+ : n_reset)
    for (int e = 0; e < N; ++e) {
        float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
        const float* cf = ORC_F(s, LSIM_BUF_CONTACT_FORCES) + 3 * N_BODY * e;
        const float* body = ORC_F(s, LSIM_BUF_RIGID_BODY_STATES) + 13 * N_BODY * e;
        float* cmd = ORC_F(s, LSIM_BUF_COMMANDS) + 4 * e;
        int64_t* eplen = ORC_I64(s, LSIM_BUF_EPISODE_LENGTH) + e;
        *eplen += 1; /* LR:193 */
        /* LR:197-200 */
        quat_rotate_inverse(root + 3, root + 7, ORC_F(s, LSIM_BUF_BASE_LIN_VEL) + 3 * e);
        quat_rotate_inverse(root + 3, root + 10, ORC_F(s, LSIM_BUF_BASE_ANG_VEL) + 3 * e);
        quat_rotate_inverse(root + 3, gvec, ORC_F(s, LSIM_BUF_PROJECTED_GRAVITY) + 3 * e);
        /* LR:207-209 */
        uint8_t* lc = ORC_U8(s, LSIM_BUF_LAST_CONTACTS) + 4 * e;
        uint8_t* filt = ORC_U8(s, LSIM_BUF_CONTACT_FILT) + 4 * e;
        for (int f = 0; f < 4; ++f) {
            uint8_t contact = cf[3 * s->model.feet_bodies[f] + 2] > 1.0f;
            filt[f] = contact | lc[f];
            lc[f] = contact;
        }
        /* _post_physics_step_callback, LR:607-632 */
        if (*eplen % c->resampling_steps == 0) resample_commands(s, e, stepw, LSIM_RNG_CMD);
        if (c->heading_command) {
            float f3[3];
            quat_apply(root + 3, fwd, f3);
            float heading = atan2f(f3[1], f3[0]);
            cmd[2] = clipf(0.5f * wrap_to_pi(cmd[3] - heading), -2.0f, 2.0f);
        }
        if (c->measure_heights) get_heights(s, e, ORC_F(s, LSIM_BUF_MEASURED_HEIGHTS) + N_HP * e);
        if (c->push_robots && (s->step_counter % c->push_interval == 0)) { /* LR:822-828 */
            root[7] = rand_range(u01(s, e, stepw, LSIM_RNG_PUSH, 0), -(double)c->max_push_vel_xy, (double)c->max_push_vel_xy);
            root[8] = rand_range(u01(s, e, stepw, LSIM_RNG_PUSH, 1), -(double)c->max_push_vel_xy, (double)c->max_push_vel_xy);
        }
        float disturbance[3] = {0.0f, 0.0f, 0.0f};
        if (c->disturbance && (s->step_counter % c->disturbance_interval == 0)) { /* LR:838-844 */
            float* pf = ORC_F(s, LSIM_BUF_PENDING_FORCE) + 3 * e;
            for (int k = 0; k < 3; ++k) {
                disturbance[k] = rand_range(u01(s, e, stepw, LSIM_RNG_DISTURB, (uint32_t)k), c->disturbance_range[0], c->disturbance_range[1]);
                pf[k] = disturbance[k];
            }
        }
        /* check_termination, LR:249-286 */
        uint8_t r = 0;
        for (int b = 0; b < N_BODY; ++b)
            if (((s->model.termination_body_mask >> b) & 1u) && norm3(cf + 3 * b) > 1.0f) r = 1;
        tout[e] = *eplen > c->max_episode_length;
        r |= tout[e];
        const float* blv = ORC_F(s, LSIM_BUF_BASE_LIN_VEL) + 3 * e;
        if (c->term_base_vel_violate_commands) {
            float ve = blv[0] - cmd[0];
            uint8_t v = ((ve > 2.0f) && (cmd[0] < 0.0f)) || ((ve < -2.0f) && (cmd[0] > 0.0f));
            v = v && (ORC_I64(s, LSIM_BUF_TERRAIN_LEVELS)[e] > 3);
            r |= v;
        }
        if (c->term_out_of_border) { /* TER:220-227 */
            float xs = c->terrain_length * (float)c->terrain_num_rows + c->border_size / 2.0f;
            float ys = c->terrain_width * (float)c->terrain_num_cols + c->border_size / 2.0f;
            uint8_t in = root[0] >= 0.0f && root[1] >= 0.0f && root[0] < xs && root[1] < ys;
            r |= !in;
        }
        if (c->term_fall_down) r |= root[9] < -5.0f;
        reset[e] = r;

        /* compute_reward, LR:363-380 */
        step_ctx x;
        x.e = e; x.root = root; x.dof = ORC_F(s, LSIM_BUF_DOF_STATE) + 2 * N_DOF * e; x.cf = cf; x.body = body; x.cmd = cmd;
        x.act = ORC_F(s, LSIM_BUF_ACTIONS) + N_DOF * e; x.last_act = ORC_F(s, LSIM_BUF_LAST_ACTIONS) + N_DOF * e;
        x.last_last_act = ORC_F(s, LSIM_BUF_LAST_LAST_ACTIONS) + N_DOF * e;
        x.last_dof_pos = ORC_F(s, LSIM_BUF_LAST_DOF_POS) + N_DOF * e; x.last_dof_vel = ORC_F(s, LSIM_BUF_LAST_DOF_VEL) + N_DOF * e;
        x.tau = ORC_F(s, LSIM_BUF_TORQUES) + N_DOF * e; x.last_tau = ORC_F(s, LSIM_BUF_LAST_TORQUES) + N_DOF * e;
        x.lin_vel = blv; x.ang_vel = ORC_F(s, LSIM_BUF_BASE_ANG_VEL) + 3 * e; x.grav = ORC_F(s, LSIM_BUF_PROJECTED_GRAVITY) + 3 * e;
        x.contact_filt = filt;
        for (int f = 0; f < 4; ++f) for (int k = 0; k < 3; ++k) x.feet_pos[f][k] = body[13 * s->model.feet_bodies[f] + k];
        float rew = 0.0f;
        float* es = ORC_F(s, LSIM_BUF_EPISODE_SUMS) + e * LSIM_NUM_REWARD_TERMS;
        for (int i = 0; i < s->num_active; ++i) {
            int id = s->active_terms[i];
            float v = reward_term(s, &x, id) * c->reward_scales[id];
            rew += v;
            es[id] += v;
        }
        if (c->only_positive_rewards) rew = fmaxf(rew, 0.0f);
        if (c->reward_scales[LSIM_R_TERMINATION] != 0.0f) {
            float v = reward_term(s, &x, LSIM_R_TERMINATION) * c->reward_scales[LSIM_R_TERMINATION];
            rew += v;
            es[LSIM_R_TERMINATION] += v;
        }
        ORC_F(s, LSIM_BUF_REW)[e] = rew;

        /* LR:227-228: termination observations / terminal AMP states of the pre-reset state */
        if (r) {
            ++n_reset;
            build_obs238(s, e, stepw, LSIM_RNG_TERM_NOISE, disturbance, ORC_F(s, LSIM_BUF_TERM_PRIV_OBS) + LSIM_NUM_PRIV_OBS * e);
            float* ta = ORC_F(s, LSIM_BUF_TERM_AMP_OBS) + LSIM_NUM_AMP_OBS * e;
            for (int j = 0; j < N_DOF; ++j) ta[j] = x.dof[2 * j];
            for (int k = 0; k < 3; ++k) { ta[12 + k] = blv[k]; ta[15 + k] = x.ang_vel[k]; }
            for (int j = 0; j < N_DOF; ++j) ta[18 + j] = x.dof[2 * j + 1];
        }
        /* the drawn disturbance (self.disturbance[:,0,:], LR:843) is re-read from PENDING_FORCE by the
           observation pass below; physics consumes and clears it in the next step's first sub-step */
    }

    /* LR:229 */
    if (!(flags & LSIM_STEP_NO_RESET)) reset_idx(s, reset, n_reset, stepw);

    /* LR:232 compute_observations + LR:167-171 clip + LR:235-241 tail */
    const int disturbed = c->disturbance && (s->step_counter % c->disturbance_interval == 0);
    #pragma omp parallel for schedule(static)
    for (int e = 0; e < N; ++e) {
        float cur[LSIM_NUM_PRIV_OBS];
        float dist[3] = {0.0f, 0.0f, 0.0f};
        if (disturbed) for (int k = 0; k < 3; ++k) dist[k] = ORC_F(s, LSIM_BUF_PENDING_FORCE)[3 * e + k];
        build_obs238(s, e, stepw, LSIM_RNG_OBS_NOISE, dist, cur);
        float* obs = ORC_F(s, LSIM_BUF_OBS) + LSIM_NUM_OBS * e;
        memmove(obs + 45, obs, sizeof(float) * (LSIM_NUM_OBS - 45)); /* LR:403: [cur45 | old[:-45]] */
        for (int k = 0; k < 45; ++k) obs[k] = cur[k];
        for (int k = 0; k < LSIM_NUM_OBS; ++k) obs[k] = clipf(obs[k], -c->clip_observations, c->clip_observations);
        float* priv = ORC_F(s, LSIM_BUF_PRIV_OBS) + LSIM_NUM_PRIV_OBS * e;
        for (int k = 0; k < LSIM_NUM_PRIV_OBS; ++k) priv[k] = clipf(cur[k], -c->clip_observations, c->clip_observations);
        /* AMP features of the post-step state (LR:406-416) */
        const float* dof = ORC_F(s, LSIM_BUF_DOF_STATE) + 2 * N_DOF * e;
        float* amp = ORC_F(s, LSIM_BUF_AMP_OBS) + LSIM_NUM_AMP_OBS * e;
        for (int j = 0; j < N_DOF; ++j) amp[j] = dof[2 * j];
        for (int k = 0; k < 3; ++k) { amp[12 + k] = ORC_F(s, LSIM_BUF_BASE_LIN_VEL)[3 * e + k]; amp[15 + k] = ORC_F(s, LSIM_BUF_BASE_ANG_VEL)[3 * e + k]; }
        for (int j = 0; j < N_DOF; ++j) amp[18 + j] = dof[2 * j + 1];
        /* tail, LR:235-241 */
        const float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
        for (int j = 0; j < N_DOF; ++j) {
            ORC_F(s, LSIM_BUF_LAST_LAST_ACTIONS)[N_DOF * e + j] = ORC_F(s, LSIM_BUF_LAST_ACTIONS)[N_DOF * e + j];
            ORC_F(s, LSIM_BUF_LAST_ACTIONS)[N_DOF * e + j] = ORC_F(s, LSIM_BUF_ACTIONS)[N_DOF * e + j];
            ORC_F(s, LSIM_BUF_LAST_DOF_POS)[N_DOF * e + j] = dof[2 * j];
            ORC_F(s, LSIM_BUF_LAST_DOF_VEL)[N_DOF * e + j] = dof[2 * j + 1];
            ORC_F(s, LSIM_BUF_LAST_TORQUES)[N_DOF * e + j] = ORC_F(s, LSIM_BUF_TORQUES)[N_DOF * e + j];
        }
        for (int k = 0; k < 6; ++k) ORC_F(s, LSIM_BUF_LAST_ROOT_VEL)[6 * e + k] = root[7 + k];
    }
}

/* ------------------------------------------------------------------ public API (mirrors lsim.h, host pointers) */

int orc_sizeof_config(void) { return (int)sizeof(lsim_config); }
int orc_sizeof_model(void) { return (int)sizeof(lsim_robot_model); }

static int check_cfg(const lsim_config* c) {
    if (c->abi_version != LSIM_ABI_VERSION) return LSIM_E_ABI;
    if (c->num_envs <= 0 || c->decimation <= 0 || c->decimation > 16) return LSIM_E_INVALID;
    if (c->mesh_type != 0 && (c->grid_rows < 2 || c->grid_cols < 2)) return LSIM_E_INVALID;
    if (c->measure_heights && c->num_points_x * c->num_points_y != LSIM_NUM_HEIGHT_PTS) return LSIM_E_INVALID;
    if (c->resampling_steps <= 0 || c->max_episode_length <= 0) return LSIM_E_INVALID;
    if (c->terrain_num_rows > LSIM_TERRAIN_LEVELS_MAX || c->terrain_num_cols > LSIM_TERRAIN_TYPES_MAX) return LSIM_E_INVALID;
    if (c->solver_type != LSIM_SOLVER_PGS && c->solver_type != LSIM_SOLVER_TGS) return LSIM_E_INVALID;
    if (c->solver_type == LSIM_SOLVER_TGS && (c->num_position_iterations < 1 || c->num_position_iterations > LSIM_MAX_POSITION_ITERATIONS)) return LSIM_E_INVALID;
    if (c->tgs_limit_passes < 0 || c->tgs_limit_passes > LSIM_MAX_POSITION_ITERATIONS || (c->lin_vel_at_com != 0 && c->lin_vel_at_com != 1)) return LSIM_E_INVALID;
    return LSIM_OK;
}

int orc_create(const lsim_config* cfg, const lsim_robot_model* model, const int16_t* grid, const float* origins, orc_sim** out) {
    int rc = check_cfg(cfg);
    if (rc != LSIM_OK) return rc;
    orc_sim* s = (orc_sim*)calloc(1, sizeof(orc_sim));
    if (!s) return LSIM_E_NOMEM;
    s->cfg = *cfg; s->model = *model;
    const lsim_config* c = &s->cfg;
    const int N = c->num_envs;
    for (int id = 0; id < LSIM_NUM_BUFFERS; ++id) {
        size_t b = lsim_buffer_bytes(c, id);
        int64_t shp[4]; int nd, dt;
        (void)lsim_buffer_desc(c, id, shp, &nd, &dt);
        if (shp[0] == N && b >= (size_t)N) {
            /* per-env buffer: zeroed -- i.e. its pages first touched, and so placed on the NUMA node of -- by the thread that will own the env block
             * in every schedule(static) loop over envs below (cpu_bench.py pins the threads); 64-byte aligned: blocks of different threads share
             * at most one cache line per buffer */
            const size_t row = b / (size_t)N;
            void* p = NULL;
            if (posix_memalign(&p, 64, b) != 0) return LSIM_E_NOMEM;
            s->buf[id] = p;
            #pragma omp parallel for schedule(static)
            for (int e = 0; e < N; ++e) memset((char*)p + row * (size_t)e, 0, row);
        } else {
            s->buf[id] = calloc(1, b ? b : 1);
            if (!s->buf[id]) return LSIM_E_NOMEM;
        }
    }
    if (c->mesh_type != 0) {
        if (!grid || !origins) return LSIM_E_INVALID;
        memcpy(s->buf[LSIM_BUF_HEIGHT_GRID], grid, lsim_buffer_bytes(c, LSIM_BUF_HEIGHT_GRID));
        orc_build_mesh_cache(s);
        memcpy(s->buf[LSIM_BUF_TERRAIN_ORIGINS], origins, lsim_buffer_bytes(c, LSIM_BUF_TERRAIN_ORIGINS));
    }
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 2; ++k) s->command_ranges[i][k] = (double)c->command_ranges[i][k];
    s->num_active = 0;
    for (int id = 0; id < LSIM_NUM_REWARD_TERMS; ++id)
        if (id != LSIM_R_TERMINATION && c->reward_scales[id] != 0.0f) s->active_terms[s->num_active++] = id;
    /* init-time draws (LR:999-1032, LR:1172-1179, LR:506-513, LR:1232-1239) */
    const uint32_t W = 0xFFFFFFFFu;
    #pragma omp parallel for schedule(static)      /* every draw is a pure function of (env, tag, index) */
    for (int e = 0; e < N; ++e) {
        for (int j = 0; j < N_DOF; ++j)
            ORC_F(s, LSIM_BUF_MOTOR_STRENGTH)[N_DOF * e + j] =
                c->randomize_motor_strength ? rand_range(u01(s, e, W, LSIM_RNG_INIT, (uint32_t)j), c->motor_strength_range[0], c->motor_strength_range[1]) : 1.0f;
        ORC_F(s, LSIM_BUF_KP_FACTORS)[e] = c->randomize_kp ? rand_range(u01(s, e, W, LSIM_RNG_INIT, 12), c->kp_range[0], c->kp_range[1]) : 1.0f;
        ORC_F(s, LSIM_BUF_KD_FACTORS)[e] = c->randomize_kd ? rand_range(u01(s, e, W, LSIM_RNG_INIT, 13), c->kd_range[0], c->kd_range[1]) : 1.0f;
        ORC_F(s, LSIM_BUF_MOTOR_STRENGTH_FACTORS)[e] =
            c->randomize_motor_strength ? rand_range(u01(s, e, W, LSIM_RNG_INIT, 14), c->motor_strength_range[0], c->motor_strength_range[1]) : 1.0f;
        ORC_F(s, LSIM_BUF_PAYLOAD)[e] = c->randomize_payload_mass ? rand_range(u01(s, e, W, LSIM_RNG_INIT, 15), c->payload_mass_range[0], c->payload_mass_range[1]) : 0.0f;
        for (int k = 0; k < 3; ++k)
            ORC_F(s, LSIM_BUF_COM_DISPLACEMENT)[3 * e + k] =
                c->randomize_com_displacement ? rand_range(u01(s, e, W, LSIM_RNG_INIT, (uint32_t)(16 + k)), c->com_displacement_range[0], c->com_displacement_range[1]) : 0.0f;
        if (c->randomize_friction) { /* 64 buckets, LR:506-513 */
            int bucket = (int)(u01(s, e, W, LSIM_RNG_INIT, 19) * 64.0f);
            ORC_F(s, LSIM_BUF_FRICTION)[e] = rand_range(u01(s, bucket, W, LSIM_RNG_INIT_BUCKET, 0), c->friction_range[0], c->friction_range[1]);
        } else
            ORC_F(s, LSIM_BUF_FRICTION)[e] = 1.0f;
        ORC_F(s, LSIM_BUF_RESTITUTION)[e] = 0.0f;
        ORC_U8(s, LSIM_BUF_RESET)[e] = 1; /* BT:72 */
        if (c->mesh_type != 0) {
            int max_init = c->terrain_curriculum ? c->max_init_terrain_level : c->terrain_num_rows - 1;
            int64_t lvl = (int64_t)(u01(s, e, W, LSIM_RNG_INIT, 20) * (float)(max_init + 1));
            int64_t type = lsim_terrain_type_of_env(e, N, c->terrain_num_cols); /* LR:1234 */
            ORC_I64(s, LSIM_BUF_TERRAIN_LEVELS)[e] = lvl;
            ORC_I64(s, LSIM_BUF_TERRAIN_TYPES)[e] = type;
            const float* to = ORC_F(s, LSIM_BUF_TERRAIN_ORIGINS) + (lvl * c->terrain_num_cols + type) * 3;
            for (int k = 0; k < 3; ++k) ORC_F(s, LSIM_BUF_ENV_ORIGINS)[3 * e + k] = to[k];
        } else { /* grid of robots, LR:1243-1250: left to the host (plane is not a BASELINE config) */
            for (int k = 0; k < 3; ++k) ORC_F(s, LSIM_BUF_ENV_ORIGINS)[3 * e + k] = 0.0f;
        }
        /* actors are created at base_init_state + origin (LR:1185-1193); quaternion identity */
        float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
        for (int k = 0; k < 13; ++k) root[k] = c->base_init_state[k];
        for (int k = 0; k < 3; ++k) root[k] += ORC_F(s, LSIM_BUF_ENV_ORIGINS)[3 * e + k];
        for (int j = 0; j < N_DOF; ++j) ORC_F(s, LSIM_BUF_DOF_STATE)[2 * N_DOF * e + 2 * j] = 0.0f;
        orc_refresh_body_states(s, e);
    }
    refresh_stats_ranges(s);
    s->stats_row = 1; refresh_stats_ranges(s); s->stats_row = 0;
    s->step_counter = 0;
    s->init_done = 1; /* __init__ completes before the runner's first reset (LR:116, HIMR:84) */
    *out = s;
    return LSIM_OK;
}

int orc_get_buffer(orc_sim* s, int id, void** ptr, int64_t shape[4], int* ndim, int* dtype) {
    if (!s || id < 0 || id >= LSIM_NUM_BUFFERS) return LSIM_E_INVALID;
    *ptr = s->buf[id];
    return lsim_buffer_desc(&s->cfg, id, shape, ndim, dtype);
}

int orc_get_stats_row(orc_sim* s, int* row) { *row = s->stats_row; return LSIM_OK; }

int orc_reset_all(orc_sim* s) { /* BT:113: reset_idx(arange(N)) */
    const int N = s->cfg.num_envs;
    s->stats_row ^= 1;
    refresh_stats_ranges(s);
    uint8_t* mask = (uint8_t*)malloc((size_t)N);
    memset(mask, 1, (size_t)N);
    reset_idx(s, mask, N, (uint32_t)s->step_counter);
    free(mask);
    return LSIM_OK;
}

int orc_reset_envs(orc_sim* s, const uint8_t* mask) { /* LR:290: reset_idx(env_ids) called from outside a step; mask[e] != 0 <=> e in env_ids */
    const int N = s->cfg.num_envs;
    int n = 0;
    for (int e = 0; e < N; ++e) n += mask[e] != 0;
    s->stats_row ^= 1;
    float* st = stats_row(s);
    st[LSIM_STATS_RESET_COUNT] = 0.0f;
    for (int k = 0; k < LSIM_NUM_REWARD_TERMS; ++k) st[LSIM_STATS_EPISODE_SUMS + k] = 0.0f;
    refresh_stats_ranges(s);
    reset_idx(s, mask, n, (uint32_t)s->step_counter ^ ((++s->reset_calls) * 0x9E3779B9u));
    return LSIM_OK;
}

int orc_step_ex(orc_sim* s, const float* actions, uint32_t flags) {
    const lsim_config* c = &s->cfg;
    const int N = c->num_envs;
    const uint32_t stepw = (uint32_t)(s->step_counter + 1);
    s->stats_row ^= 1;
    /* envs are independent (no robot-robot contact, LR:1193) and every random draw is a pure function of (env, step, tag): the
       result does not depend on the thread count.  OMP_NUM_THREADS = 1 is the scalar port. */
    /* schedule(static): a contiguous block of envs per thread.  (Round 1-5: dynamic, 1 -- neighbouring envs, i.e. neighbouring rows of every [N, k]
     * buffer and often the same cache line, on different threads: false sharing on every store of the step; 256 threads gave 7 x one core.) */
    #pragma omp parallel for schedule(static)
    for (int e = 0; e < N; ++e) {
        float* act = ORC_F(s, LSIM_BUF_ACTIONS) + N_DOF * e;
        const float* last = ORC_F(s, LSIM_BUF_LAST_ACTIONS) + N_DOF * e;
        for (int j = 0; j < N_DOF; ++j) act[j] = clipf(actions[N_DOF * e + j], -c->clip_actions, c->clip_actions); /* LR:129-130 */
        int delay = (int)(u01(s, e, stepw, LSIM_RNG_DELAY, 0) * (float)c->decimation); /* LR:134 */
        ORC_I32(s, LSIM_BUF_DELAY_STEPS)[e] = delay;
        ORC_I32(s, LSIM_BUF_CONTACT_COUNT)[2 * e] = 0; ORC_I32(s, LSIM_BUF_CONTACT_COUNT)[2 * e + 1] = 0;
        const int at_com = c->lin_vel_at_com && !(flags & LSIM_STEP_SKIP_PHYSICS);
        if (at_com) orc_root_lin_vel_to_origin(s, e);
        for (int sub = 0; sub < c->decimation; ++sub) { /* LR:144-152 */
            float a[N_DOF], tau[N_DOF];
            for (int j = 0; j < N_DOF; ++j)
                a[j] = c->delay ? last[j] + (act[j] - last[j]) * ((sub >= delay) ? 1.0f : 0.0f) : act[j]; /* LR:138 */
            compute_torques(s, e, a, tau);
            memcpy(ORC_F(s, LSIM_BUF_TORQUES) + N_DOF * e, tau, sizeof(tau));
            if (flags & LSIM_STEP_RECORD_SUBSTEPS) memcpy(ORC_F(s, LSIM_BUF_SUBSTEP_TORQUES) + N_DOF * (c->decimation * e + sub), tau, sizeof(tau));
            if (!(flags & LSIM_STEP_SKIP_PHYSICS)) orc_physics_substep(s, e, tau, sub == 0);
        }
        if (!(flags & LSIM_STEP_SKIP_PHYSICS)) orc_refresh_body_states(s, e);
        if (at_com) /* the root tensor's linear velocity is row 0's of the body tensor: the centre of mass's from here on */
            for (int k = 0; k < 3; ++k) ORC_F(s, LSIM_BUF_ROOT_STATES)[13 * e + 7 + k] = ORC_F(s, LSIM_BUF_RIGID_BODY_STATES)[13 * N_BODY * e + 7 + k];
    }
    /* robots whose simulated state is not finite (LSIM_BUF_NONFINITE / LSIM_STATS_NONFINITE, include/lsim.h) */
    int bad = 0;
    if (!(flags & LSIM_STEP_SKIP_PHYSICS))
        for (int e = 0; e < N; ++e) {
            int b = 0;
            for (int k = 0; k < 13; ++k) b |= !isfinite(ORC_F(s, LSIM_BUF_ROOT_STATES)[13 * e + k]);
            for (int k = 0; k < 2 * N_DOF; ++k) b |= !isfinite(ORC_F(s, LSIM_BUF_DOF_STATE)[2 * N_DOF * e + k]);
            bad += b;
        }
    stats_row(s)[LSIM_STATS_NONFINITE] = (float)bad;
    if (bad) { ORC_I64(s, LSIM_BUF_NONFINITE)[0] += bad; ORC_I64(s, LSIM_BUF_NONFINITE)[1] = s->step_counter + 1; }
    post_physics_step(s, flags);
    return LSIM_OK;
}

int orc_step(orc_sim* s, const float* actions) { return orc_step_ex(s, actions, LSIM_STEP_DEFAULT); }

int orc_get_step_counter(orc_sim* s, int64_t* out) { *out = s->step_counter; return LSIM_OK; }
int orc_set_step_counter(orc_sim* s, int64_t v) { s->step_counter = v; return LSIM_OK; }
int orc_set_init_done(orc_sim* s, int v) { s->init_done = v; return LSIM_OK; }
int orc_get_command_ranges(orc_sim* s, double out[8]) { memcpy(out, s->command_ranges, sizeof(double) * 8); return LSIM_OK; }

/* ---- the CPU baseline's timing loop (oracle/cpu_bench.py; SURVEY.md 8d: three action sources), entirely in C: n_steps x { actions for every env,
 * orc_step } with the wall clock around it.  mode 0: zero actions (standing); 1: the caller's table of pre-drawn N(0,1) action sets [table_len][N][12],
 * cycled (= an untrained policy's samples, init_noise_std = 1, AGC:299); 2: closed loop with the caller's HIMActorCritic weights (HAC:136-163,
 * HES:64-68: encoder 270 -> .. -> 3 + 16, L2-normalised latent, actor on [obs[:45], v, z]) -- mean action + the table's noise.  fp32, one env per
 * thread at a time; weights [out][in] row-major as torch.nn.Linear keeps them. */
typedef struct orc_mlp_layer { const float* w; const float* b; int n_in, n_out, elu; } orc_mlp_layer;
typedef struct orc_policy { orc_mlp_layer enc[3]; orc_mlp_layer act[4]; } orc_policy;
static void mlp_layer(const orc_mlp_layer* L, const float* x, float* y) {
    for (int o = 0; o < L->n_out; ++o) {
        const float* w = L->w + (size_t)o * L->n_in;
        float a = 0.0f;
        #pragma omp simd reduction(+ : a)      /* (lets the compiler vectorise the dot product: a different summation order than torch's, irrelevant for a timing loop) */
        for (int i = 0; i < L->n_in; ++i) a += w[i] * x[i];
        a += L->b[o];
        y[o] = (L->elu && a < 0.0f) ? expm1f(a) : a;
    }
}
static void policy_mean(const orc_policy* p, const float* obs, float* out12) {
    float a[512], b[512];
    mlp_layer(&p->enc[0], obs, a); mlp_layer(&p->enc[1], a, b); mlp_layer(&p->enc[2], b, a);     /* a[0..2] velocity, a[3..18] latent */
    float in[64], ss = 0.0f;
    for (int k = 0; k < 16; ++k) ss += a[3 + k] * a[3 + k];
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    for (int k = 0; k < 45; ++k) in[k] = obs[k];
    for (int k = 0; k < 3; ++k) in[45 + k] = a[k];
    for (int k = 0; k < 16; ++k) in[48 + k] = a[3 + k] * inv;
    mlp_layer(&p->act[0], in, a); mlp_layer(&p->act[1], a, b); mlp_layer(&p->act[2], b, a); mlp_layer(&p->act[3], a, out12);
}
/* the number of threads an OpenMP parallel region of this library really runs with (cpu_bench.py reports it as `cores`) */
int orc_parallel_threads(void) {
    int n = 1;
#ifdef _OPENMP            /* (the sanitizer build of tests/test_sanitizers.py compiles this file without OpenMP) */
    #pragma omp parallel
    {
        #pragma omp master
        n = omp_get_num_threads();
    }
#endif
    return n;
}
int orc_run_steps(orc_sim* s, int n_steps, int mode, const float* table, int table_len, const orc_policy* pol, double* seconds) {
    if (!s || n_steps < 0 || !seconds || (mode != 0 && (!table || table_len <= 0)) || (mode == 2 && !pol)) return LSIM_E_INVALID;
    if (mode == 2 && (pol->enc[0].n_in != LSIM_NUM_OBS || pol->enc[2].n_out != 19 || pol->act[0].n_in != 64 || pol->act[3].n_out != N_DOF ||
                      pol->enc[0].n_out > 512 || pol->enc[1].n_out > 512 || pol->act[0].n_out > 512 || pol->act[1].n_out > 512 || pol->act[2].n_out > 512))
        return LSIM_E_UNSUPPORTED;
    const int N = s->cfg.num_envs;
    float* act = NULL;
    if (posix_memalign((void**)&act, 64, (size_t)N * N_DOF * sizeof(float)) != 0) return LSIM_E_NOMEM;
    #pragma omp parallel for schedule(static)
    for (int e = 0; e < N; ++e) memset(act + N_DOF * e, 0, N_DOF * sizeof(float));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_steps; ++t) {
        if (mode != 0) {
            const float* z = table + (size_t)(t % table_len) * N * N_DOF;
            #pragma omp parallel for schedule(static)
            for (int e = 0; e < N; ++e) {
                float m[N_DOF] = {0};
                if (mode == 2) policy_mean(pol, ORC_F(s, LSIM_BUF_OBS) + LSIM_NUM_OBS * e, m);
                for (int j = 0; j < N_DOF; ++j) act[N_DOF * e + j] = m[j] + z[N_DOF * e + j];
            }
        }
        int rc = orc_step_ex(s, act, LSIM_STEP_DEFAULT);
        if (rc != LSIM_OK) { free(act); return rc; }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    free(act);
    return LSIM_OK;
}

/* Diagnostic for the modelling-distance tables of DESIGN.md section 4 (tools/trained_policy_physics.py), not part of the step: clearance of an arbitrary
 * set of points RIGIDLY ATTACHED TO A BODY -- pts [n][4] = (x, y, z in the body frame, radius) -- from the terrain surface, for every env:
 * out[e] = min over the points of (signed distance of the point's centre - radius).  Used with a dense sampling of the trunk's true box to see
 * whether terrain passes between the shipped model's trunk sample points (negative clearance while the simulated base reports no contact). */
int orc_body_clearance(orc_sim* s, int body, const float* pts, int n, float* out) {
    if (!s || !pts || !out || body < 0 || body >= N_BODY) return LSIM_E_INVALID;
    const int N = s->cfg.num_envs;
    #pragma omp parallel for schedule(static)
    for (int e = 0; e < N; ++e) {
        const float* b = ORC_F(s, LSIM_BUF_RIGID_BODY_STATES) + 13 * (N_BODY * e + body);
        const double qx = b[3], qy = b[4], qz = b[5], qw = b[6];
        const double R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)},
                                {2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)},
                                {2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)}};
        double best = 1e30;
        for (int i = 0; i < n; ++i) {
            const float* p = pts + 4 * i;
            double cw[3], dist, nn[3];
            for (int k = 0; k < 3; ++k) cw[k] = b[k] + R[k][0] * p[0] + R[k][1] * p[1] + R[k][2] * p[2];
            orc_terrain_contact(s, cw, (double)p[3], &dist, nn);
            if (dist - p[3] < best) best = dist - p[3];
        }
        out[e] = (float)best;
    }
    return LSIM_OK;
}

void orc_destroy(orc_sim* s) {
    if (!s) return;
    for (int id = 0; id < LSIM_NUM_BUFFERS; ++id) free(s->buf[id]);
    free(s->vmove); free(s->cell_walls);
    free(s);
}
