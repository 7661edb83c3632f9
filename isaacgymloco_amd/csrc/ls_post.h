// ls_post.h -- the reference's torch post-physics stack as wave phases (one wavefront per robot).
// Every function cites the lines of legged_gym/envs/base/legged_robot.py ("LR") it restates; the CPU oracle
// (oracle/lsim_oracle.c) states the same logic as scalar loops and both are pinned by tests/golden.
#pragma once
#include "ls_shared.h"

#define LSB(cx, id, T) LS_G(T, (cx).buf[id])
#define LS_NHP LSIM_NUM_HEIGHT_PTS

LS_FN float ls_draw(const LsCtx& cx, int env, uint32_t stepw, uint32_t tag, uint32_t idx) {
    return ls_u01(cx.cfg.seed, cx.cfg.rank, (uint32_t)env, stepw, tag, idx);
}

// body index of foot f for a per-lane f: four uniform (scalar) loads and selects -- indexing the model table in global memory by a lane
// value is a vector load, and behind the post-physics stack's stores it waits for all of them (vmcnt retires in order)
LS_FN int ls_foot_body(const LsCtx& cx, int f) {
    const int b0 = cx.model.feet_bodies[0], b1 = cx.model.feet_bodies[1], b2 = cx.model.feet_bodies[2], b3 = cx.model.feet_bodies[3];
    return f == 0 ? b0 : (f == 1 ? b1 : (f == 2 ? b2 : b3));
}

// LR:1342-1355: (x + border) / hscale truncated toward zero, clip, min of 3 samples, * vscale
LS_FN float ls_sample_height_min3(const LsCtx& cx, float x, float y) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const lsim_config& c = cx.cfg;
    float fx = ls_div_exact(x + c.border_size, c.horizontal_scale);
    float fy = ls_div_exact(y + c.border_size, c.horizontal_scale);
    int px = ls_f2i(fx), py = ls_f2i(fy);
    px = px < 0 ? 0 : (px > c.grid_rows - 2 ? c.grid_rows - 2 : px);
    py = py < 0 ? 0 : (py > c.grid_cols - 2 ? c.grid_cols - 2 : py);
    LS_GLOBAL const int16_t* g = LSB(cx, LSIM_BUF_HEIGHT_GRID, const int16_t);
    int16_t h1 = g[px * c.grid_cols + py], h2 = g[(px + 1) * c.grid_cols + py], h3 = g[px * c.grid_cols + py + 1];
    int16_t h = h1 < h2 ? h1 : h2;
    h = h < h3 ? h : h3;
    return (float)h * c.vertical_scale;
}

// Height-sample points (LR:1336-1340: quat_apply_yaw(base_quat, points) + root position), evaluated so that every point lands in the SAME
// grid cell as the reference's: IEEE-rounded square root / division for the yaw quaternion and no fused multiply-adds, term by term as
// torch evaluates normalize() and quat_apply() (MTH:38-42; t = cross(xyz, b) * 2; b + w * t + cross(xyz, t) with xyz = (0, 0, qz)).
// With the simulator's fast division / contraction the points moved by an ulp, and of the 2.3 M samples of an N = 4096 step about one per
// step fell into the neighbouring cell -- a stair riser or a pit wall away (found by the N = 4096 golden fixture, round 3).
struct LsYawQuat { float z, w; };
LS_FN LsYawQuat ls_yaw_quat(const float* q) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    float n = ls_sqrt_exact(q[2] * q[2] + q[3] * q[3]);
    if (n < 1e-9f) n = 1e-9f;
    LsYawQuat y;
    y.z = ls_div_exact(q[2], n); y.w = ls_div_exact(q[3], n);
    return y;
}
LS_FN V3 ls_yaw_point(const float* root, LsYawQuat y, float px, float py) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float tx = (0.0f - y.z * py) * 2.0f, ty = (y.z * px) * 2.0f;
    const float ux = 0.0f - y.z * ty, uy = y.z * tx;
    const float wx = px + y.w * tx + ux, wy = py + y.w * ty + uy;
    return v3(wx + root[0], wy + root[1], 0.0f);
}

// ---- the same two samplers split in two for kernel A, whose post-physics stack begins with the state stores: the grid loads of this lane's
//      points are ISSUED before the first store (ph_heights_issue, last phase of the physics) and consumed two phases later
//      (ph_heights_finish), so that neither their latency nor the drain of the stores in front of them sits on the wave's critical path.
//      Lane l owns height points l, l + 64, l + 128 (187 in all) and base-height point l (63): 4 x (a 4-byte pair + a 2-byte sample) in LaneRegs::hraw.
#define LS_HEIGHT_PASSES ((LS_NHP + 63) / 64)
LS_FN void ls_height_samples3(const LsCtx& cx, float x, float y, int* h2) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const lsim_config& c = cx.cfg;
    float fx = ls_div_exact(x + c.border_size, c.horizontal_scale);
    float fy = ls_div_exact(y + c.border_size, c.horizontal_scale);
    int px = ls_f2i(fx), py = ls_f2i(fy);
    px = px < 0 ? 0 : (px > c.grid_rows - 2 ? c.grid_rows - 2 : px);
    py = py < 0 ? 0 : (py > c.grid_cols - 2 ? c.grid_cols - 2 : py);
    // two registers per point and NO arithmetic on them here (anything that looks at a loaded value makes the compiler wait for it in this
    // phase): the pair (x, y), (x, y + 1) -- neighbours in the row-major grid -- as one 4-byte load, (x + 1, y) as a sign-extended 2-byte load
    LS_GLOBAL const int16_t* g = LSB(cx, LSIM_BUF_HEIGHT_GRID, const int16_t);
    LS_GLOBAL const int16_t* p0 = g + px * c.grid_cols + py;
    struct __attribute__((packed, aligned(2))) Pair { uint32_t w; };          // 2-byte aligned 4-byte load (global memory allows it)
    h2[0] = (int)((LS_GLOBAL const Pair*)p0)->w;
    h2[1] = (int)p0[c.grid_cols];
}
LS_FN void ph_heights_issue(const LsCtx& cx, const WaveShared& sh, LaneRegs& rg, int lane) {
    const lsim_config& c = cx.cfg;
    if (c.mesh_type == 0) return;
    const LsYawQuat yq = ls_yaw_quat(sh.root + 3);
    if (c.measure_heights)
        for (int it = 0; it < LS_HEIGHT_PASSES; ++it) {
            const int k = lane + 64 * it, kk = k < LS_NHP ? k : 0;
            const int ix = kk / c.num_points_y, iy = kk - ix * c.num_points_y;
            const V3 w = ls_yaw_point(sh.root, yq, sh.mpx[ix], sh.mpy[iy]);
            ls_height_samples3(cx, w.x, w.y, rg.hraw + 2 * it);
        }
    {
        const int l = lane < LSIM_NUM_BASE_HEIGHT_PTS ? lane : 0;
        const int ix = l / 9, iy = l - 9 * ix;
        // the reference's literal lists (LR:1308-1309: x -0.15 .. 0.15, y -0.2 .. 0.2 in steps of 0.05) as selects: a table indexed by a lane
        // value would be a load from constant memory, and the grid address depends on it
        const float m5[5] = {0.0f, 0.05f, 0.1f, 0.15f, 0.2f};
        const int ax = ix < 3 ? 3 - ix : ix - 3, ay = iy < 4 ? 4 - iy : iy - 4;
        const float mx = ax == 0 ? m5[0] : (ax == 1 ? m5[1] : (ax == 2 ? m5[2] : m5[3]));
        const float my = ay == 0 ? m5[0] : (ay == 1 ? m5[1] : (ay == 2 ? m5[2] : (ay == 3 ? m5[3] : m5[4])));
        const V3 w = ls_yaw_point(sh.root, yq, ix < 3 ? -mx : mx, iy < 4 ? -my : my);
        ls_height_samples3(cx, w.x, w.y, rg.hraw + 2 * LS_HEIGHT_PASSES);
    }
}
LS_FN float ls_min3_height(const LsCtx& cx, const int* h2) {
    const int h1 = (int)(int16_t)(h2[0] & 0xffff), h3 = h2[0] >> 16, hx = h2[1];     // (x, y), (x, y + 1), (x + 1, y)
    int h = h1 < hx ? h1 : hx;
    h = h < h3 ? h : h3;
    return (float)h * cx.cfg.vertical_scale;
}
LS_FN void ph_heights_finish(const LsCtx& cx, WaveShared& sh, const LaneRegs& rg, int lane, int env) {
    const lsim_config& c = cx.cfg;
    if (c.measure_heights) {
        LS_GLOBAL float* mh = LSB(cx, LSIM_BUF_MEASURED_HEIGHTS, float) + LS_NHP * env;
        for (int it = 0; it < LS_HEIGHT_PASSES; ++it) {
            const int k = lane + 64 * it;
            if (k >= LS_NHP) continue;
            const float h = c.mesh_type != 0 ? ls_min3_height(cx, rg.hraw + 2 * it) : 0.0f;
            sh.heights[k] = h;
            mh[k] = h;
        }
    }
    if (lane < LSIM_NUM_BASE_HEIGHT_PTS) sh.bh[lane] = c.mesh_type == 0 ? sh.root[2] : sh.root[2] - ls_min3_height(cx, rg.hraw + 2 * LS_HEIGHT_PASSES);
}

// LeggedRobot._get_heights (LR:1318-1355) in one phase (reset_idx: the terrain under the new pose): lane l owns points l, l + 64, l + 128.
// Every sample of every pass is requested before the first result is stored -- a pass that stored its heights before the next pass's
// loads went out made those wait for the stores (vmcnt retires in order), and the point tables were read per lane from the context, one
// more dependent round trip per pass: 64 k of a resetting wave's 346 k ticks (round 4).  mpx / mpy: measured_points_x / y (kernel A: its
// LDS copies).
LS_FN void ph_heights(const LsCtx& cx, WaveShared& sh, int lane, int env, bool store_global, const float* mpx, const float* mpy) {
    const lsim_config& c = cx.cfg;
    LS_GLOBAL float* mh = LSB(cx, LSIM_BUF_MEASURED_HEIGHTS, float) + LS_NHP * env;
    int raw[2 * LS_HEIGHT_PASSES];
    if (c.mesh_type != 0) {
        const LsYawQuat yq = ls_yaw_quat(sh.root + 3);
        for (int it = 0; it < LS_HEIGHT_PASSES; ++it) {
            const int k = lane + 64 * it, kk = k < LS_NHP ? k : 0;
            const int ix = kk / c.num_points_y, iy = kk - ix * c.num_points_y;
            const V3 w = ls_yaw_point(sh.root, yq, mpx[ix], mpy[iy]);
            ls_height_samples3(cx, w.x, w.y, raw + 2 * it);
        }
    }
    for (int it = 0; it < LS_HEIGHT_PASSES; ++it) {
        const int k = lane + 64 * it;
        if (k >= LS_NHP) continue;
        const float h = c.mesh_type != 0 ? ls_min3_height(cx, raw + 2 * it) : 0.0f;
        sh.heights[k] = h;
        if (store_global) mh[k] = h;
    }
}
// LeggedRobot._get_base_heights (LR:1357-1398): 7 x 9 points, (z - h) per point; the mean is taken by the reward lane
LS_FN void ph_base_height_pts(const LsCtx& cx, WaveShared& sh, int lane) {
    if (lane >= LSIM_NUM_BASE_HEIGHT_PTS) return;
    if (cx.cfg.mesh_type == 0) { sh.bh[lane] = sh.root[2]; return; }
    int ix = lane / 9, iy = lane - 9 * ix;
    float px = -0.15f + 0.05f * (float)ix, py = -0.2f + 0.05f * (float)iy;
    // the reference builds the grid from literal lists (LR:1308-1309); use the same fp32 literals
    const float xs[7] = {-0.15f, -0.1f, -0.05f, 0.f, 0.05f, 0.1f, 0.15f};
    const float ys[9] = {-0.2f, -0.15f, -0.1f, -0.05f, 0.f, 0.05f, 0.1f, 0.15f, 0.2f};
    px = xs[ix]; py = ys[iy];
    V3 w = ls_yaw_point(sh.root, ls_yaw_quat(sh.root + 3), px, py);
    sh.bh[lane] = sh.root[2] - ls_sample_height_min3(cx, w.x, w.y);
}

// LeggedRobot._resample_commands (LR:634-656) for this env; `ranges` = live command ranges [4][2]
// ... from the four uniforms of block 0 of its stream (drawn by the caller)
LS_FN void ls_resample_commands_u(const LsCtx& cx, int env, const float* u, const float* ranges, float* cmd);
LS_FN void ls_resample_commands(const LsCtx& cx, int env, uint32_t stepw, uint32_t tag, const float* ranges, float* cmd) {
    float u[4];
    ls_u01x4(cx.cfg.seed, cx.cfg.rank, (uint32_t)env, stepw, tag, 0, u);
    ls_resample_commands_u(cx, env, u, ranges, cmd);
}
LS_FN void ls_resample_commands_u(const LsCtx& cx, int env, const float* u, const float* ranges, float* cmd) {
    cmd[0] = rand_range(u[0], -1.0f, 1.0f);
    cmd[1] = rand_range(u[1], ranges[2], ranges[3]);
    if (cx.cfg.heading_command) cmd[3] = rand_range(u[2], ranges[6], ranges[7]);
    else cmd[2] = rand_range(u[2], ranges[4], ranges[5]);
    if ((float)env < (float)((double)cx.cfg.num_envs * 0.2)) {
        cmd[0] = rand_range(u[3], ranges[0], ranges[1]);
        cmd[1] *= (fabsf(cmd[0]) < 1.0f) ? 1.0f : 0.0f;
    }
    float m = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.2f) ? 1.0f : 0.0f;
    cmd[0] *= m; cmd[1] *= m;
}

// ---- Q1: derived base state, contact filter, episode counter (LR:193-209)
LS_FN void ph_post_state(const LsCtx& cx, WaveShared& sh, int lane, int env) {
    if (lane == 0) {
        const int v = sh.pre_eplen + 1;
        LSB(cx, LSIM_BUF_EPISODE_LENGTH, int64_t)[env] = (int64_t)v;
        sh.eplen = v;
    } else if (lane == 1) {
        V3 v = quat_rotate_inverse(sh.root + 3, v3p(sh.root + 7));
        v3st(sh.blv, v); v3st(LSB(cx, LSIM_BUF_BASE_LIN_VEL, float) + 3 * env, v);
    } else if (lane == 2) {
        V3 v = quat_rotate_inverse(sh.root + 3, v3p(sh.root + 10));
        v3st(sh.bav, v); v3st(LSB(cx, LSIM_BUF_BASE_ANG_VEL, float) + 3 * env, v);
    } else if (lane == 3) {
        V3 v = quat_rotate_inverse(sh.root + 3, v3(0.0f, 0.0f, -1.0f));
        v3st(sh.grav, v); v3st(LSB(cx, LSIM_BUF_PROJECTED_GRAVITY, float) + 3 * env, v);
    } else if (lane < 8) {
        int f = lane - 4;
        const uint8_t lc = (uint8_t)((sh.pre_lc >> (8 * f)) & 0xffu);
        uint8_t contact = sh.cf[ls_foot_body(cx, f)][2] > 1.0f;
        const uint8_t filt = contact | lc;
        sh.filt[f] = filt;
        LSB(cx, LSIM_BUF_CONTACT_FILT, uint8_t)[4 * env + f] = filt;
        LSB(cx, LSIM_BUF_LAST_CONTACTS, uint8_t)[4 * env + f] = contact;
    }
}

// ---- Q2: _post_physics_step_callback minus heights (LR:607-632): lane 0 commands, lane 1 push, lane 2 disturbance
LS_FN void ph_callback(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a, const float* ranges) {
    const lsim_config& c = cx.cfg;
    const uint32_t stepw = (uint32_t)a.step_counter;
    if (lane == 0) {
        LS_GLOBAL float* cmd = LSB(cx, LSIM_BUF_COMMANDS, float) + 4 * env;
        float cm[4] = {sh.pre_cmd[0], sh.pre_cmd[1], sh.pre_cmd[2], sh.pre_cmd[3]};
        if (sh.eplen % c.resampling_steps == 0) ls_resample_commands(cx, env, stepw, LSIM_RNG_CMD, ranges, cm);
        if (c.heading_command) {
            V3 f = quat_apply(sh.root + 3, v3(1.0f, 0.0f, 0.0f));
            float heading = atan2f(f.y, f.x);
            cm[2] = clampf(0.5f * wrap_to_pi(cm[3] - heading), -2.0f, 2.0f);
        }
        for (int k = 0; k < 4; ++k) { cmd[k] = cm[k]; sh.cmd[k] = cm[k]; }
    } else if (lane == 1) {
        if (c.push_robots && (a.step_counter % c.push_interval == 0)) {  // LR:822-828
            float u[4];
            ls_u01x4(c.seed, c.rank, (uint32_t)env, stepw, LSIM_RNG_PUSH, 0, u);
            sh.root[7] = rand_range(u[0], -c.max_push_vel_xy, c.max_push_vel_xy);
            sh.root[8] = rand_range(u[1], -c.max_push_vel_xy, c.max_push_vel_xy);
        }
    } else if (lane == 2) {
        float d[3] = {0.0f, 0.0f, 0.0f};
        if (c.disturbance && (a.step_counter % c.disturbance_interval == 0)) {  // LR:838-844
            float u[4];
            ls_u01x4(c.seed, c.rank, (uint32_t)env, stepw, LSIM_RNG_DISTURB, 0, u);
            LS_GLOBAL float* pf = LSB(cx, LSIM_BUF_PENDING_FORCE, float) + 3 * env;
            for (int k = 0; k < 3; ++k) { d[k] = rand_range(u[k], c.disturbance_range[0], c.disturbance_range[1]); pf[k] = d[k]; }
        }
        for (int k = 0; k < 3; ++k) sh.disturbance[k] = d[k];
    }
}

// ---- Q4: check_termination (LR:249-286), lane 0
LS_FN void ph_termination(const LsCtx& cx, WaveShared& sh, int lane, int env) {
    if (lane != 0) return;
    const lsim_config& c = cx.cfg;
    int r = 0;
    for (int b = 0; b < LS_NB; ++b)
        if ((cx.model.termination_body_mask >> b) & 1u) {
            V3 f = v3p(sh.cf[b]);
            if (sqrtf(dot(f, f)) > 1.0f) r = 1;
        }
    int to = sh.eplen > c.max_episode_length;
    r |= to;
    if (c.term_base_vel_violate_commands) {
        float ve = sh.blv[0] - sh.cmd[0];
        int v = ((ve > 2.0f) && (sh.cmd[0] < 0.0f)) || ((ve < -2.0f) && (sh.cmd[0] > 0.0f));
        v = v && (sh.pre_level > 3);
        r |= v;
    }
    if (c.term_out_of_border) {  // TER:220-227
        float xs = c.terrain_length * (float)c.terrain_num_rows + c.border_size / 2.0f;
        float ys = c.terrain_width * (float)c.terrain_num_cols + c.border_size / 2.0f;
        int in = sh.root[0] >= 0.0f && sh.root[1] >= 0.0f && sh.root[0] < xs && sh.root[1] < ys;
        r |= !in;
    }
    if (c.term_fall_down) r |= sh.root[9] < -5.0f;
    sh.reset = r; sh.timeout = to;
    LSB(cx, LSIM_BUF_RESET, uint8_t)[env] = (uint8_t)r;
    LSB(cx, LSIM_BUF_TIME_OUT, uint8_t)[env] = (uint8_t)to;
}

// ---------------------------------------------------------------------------------------------- rewards
struct LsRewCtx {
    const float *dof, *act, *last_act, *last_last_act, *last_dof_pos, *last_dof_vel, *tau, *last_tau;   // all staged in LDS
    const uint8_t* filt;
};
LS_FN float ls_up(const WaveShared& sh) { return clampf(-sh.grav[2], 0.0f, 1.0f); }
LS_FN float ls_cmd_norm(const WaveShared& sh) { return sqrtf(sh.cmd[0] * sh.cmd[0] + sh.cmd[1] * sh.cmd[1]); }

// ---- reward terms that are sums over joints / legs / feet / height samples are evaluated in two steps: their summands ("parts") by one lane
//      each (ph_reward_parts: lane = (term, part) pair from a host-built item table), then the term's lane adds them up IN THE ORDER of the
//      reference's torch.sum and applies the term's factor (ph_reward_terms).  One lane per term alone ran all 21 term bodies one after the
//      other -- 12-joint loops on a single lane each -- a tenth of kernel A's instructions.
LS_FN float ls_foot_slide_part(const LsCtx& cx, const WaveShared& sh, const LsRewCtx& x, bool with_height, int f) {  // LR:1610-1619 / LR:1682-1698
    const float* bs = sh.feet[f];
    V3 vb = quat_rotate_inverse(sh.root + 3, v3(bs[3] - sh.root[7], bs[4] - sh.root[8], bs[5] - sh.root[9]));
    float lat = sqrtf(vb.x * vb.x + vb.y * vb.y);
    if (with_height) {
        V3 pb = quat_rotate_inverse(sh.root + 3, v3(bs[0] - sh.root[0], bs[1] - sh.root[1], bs[2] - sh.root[2]));
        float he = pb.z - cx.cfg.foot_height_target_base;
        return (he * he) * lat;
    }
    return (x.filt[f] ? 1.0f : 0.0f) * lat;
}
LS_FN float ls_foot_clearance_terrain_part(const LsCtx& cx, const WaveShared& sh, int shifts, int f) {  // LR:1717-1743 incl. quirk 3 (in-place += border)
    const lsim_config& c = cx.cfg;
    const float* bs = sh.feet[f];
    float fh;
    if (c.mesh_type == 0) fh = bs[2];
    else {
        float px = bs[0], py = bs[1], pz = bs[2];
        for (int s = 0; s < shifts; ++s) { px += c.border_size; py += c.border_size; pz += c.border_size; }
        int ix = ls_f2i(ls_div_exact(px, c.horizontal_scale)), iy = ls_f2i(ls_div_exact(py, c.horizontal_scale));
        ix = ix < 0 ? 0 : (ix > c.grid_rows - 2 ? c.grid_rows - 2 : ix);
        iy = iy < 0 ? 0 : (iy > c.grid_cols - 2 ? c.grid_cols - 2 : iy);
        LS_GLOBAL const int16_t* g = LSB(cx, LSIM_BUF_HEIGHT_GRID, const int16_t);
        int16_t h1 = g[ix * c.grid_cols + iy], h2 = g[(ix + 1) * c.grid_cols + iy], h3 = g[ix * c.grid_cols + iy + 1];
        int16_t h = h1 < h2 ? h1 : h2; h = h < h3 ? h : h3;
        fh = pz - (float)h * c.vertical_scale;
    }
    float lat = sqrtf(bs[3] * bs[3] + bs[4] * bs[4]);
    float d = fh - c.foot_height_target_terrain;
    return lat * (d * d);
}
LS_FN float ls_stumble(const LsCtx& cx, const WaveShared& sh, int env, float ratio) {  // LR:1589-1608
    const lsim_config& c = cx.cfg;
    int any = 0;
    for (int f = 0; f < 4; ++f) {
        const float* F = sh.cf[cx.model.feet_bodies[f]];
        if (sqrtf(F[0] * F[0] + F[1] * F[1]) > ratio * fabsf(F[2])) any = 1;
    }
    float r = (any && sh.pre_level > 3) ? 1.0f : 0.0f;
    int in_slice = (env >= c.stairsup_start_idx && env < c.stairsup_end_idx) || (env >= c.pit_start_idx && env < c.gap_end_idx);
    return in_slice ? r : 0.0f;
}
LS_FN float ls_var12(const float* v) {
    float m = 0.0f;
    for (int j = 0; j < 12; ++j) m += v[j];
    m /= 12.0f;
    float a = 0.0f;
    for (int j = 0; j < 12; ++j) a += (v[j] - m) * (v[j] - m);
    return a / 11.0f;
}

// part j of term id: one summand of the `_reward_<name>()` (LR:1444-1770)
LS_FN float ls_reward_part(const LsCtx& cx, WaveShared& sh, const LsRewCtx& x, int id, int j, int env, int fct_shifts) {
    const lsim_config& c = cx.cfg;
    const float dt = c.sim_dt * (float)c.decimation;
    switch (id) {
        case LSIM_R_DOF_VEL: return x.dof[2 * j + 1] * x.dof[2 * j + 1];
        case LSIM_R_DOF_ACC: { float a = (x.last_dof_vel[j] - x.dof[2 * j + 1]) / dt; return a * a; }
        case LSIM_R_DOF_VEL_LIMITS: return clampf(fabsf(x.dof[2 * j + 1]) - sh.jc_vmax[j] * c.soft_dof_vel_limit, 0.0f, 1.0f);
        case LSIM_R_DOF_POS_DIF: { float d = x.last_dof_pos[j] - x.dof[2 * j]; return d * d; }
        case LSIM_R_DOF_POS_LIMITS: {
            float lo = sh.jc_lo[j], hi = sh.jc_hi[j];
            float m = (lo + hi) / 2.0f, r = hi - lo;
            float slo = m - 0.5f * r * c.soft_dof_pos_limit, shi = m + 0.5f * r * c.soft_dof_pos_limit;
            float q = x.dof[2 * j];
            float o = -fminf(q - slo, 0.0f);
            o += fmaxf(q - shi, 0.0f);
            return o;
        }
        case LSIM_R_ACTION_RATE: { float d = x.last_act[j] - x.act[j]; return d * d; }
        case LSIM_R_SMOOTHNESS: { float d = x.act[j] - x.last_act[j] - x.last_act[j] + x.last_last_act[j]; return d * d; }
        case LSIM_R_TORQUES: return x.tau[j] * x.tau[j];
        case LSIM_R_TORQUES_DISTRIBUTION: return fabsf(x.tau[j]);
        case LSIM_R_TORQUES_DIF: { float d = x.tau[j] - x.last_tau[j]; return d * d; }
        case LSIM_R_TORQUE_LIMITS: return fmaxf(fabsf(x.tau[j]) - sh.jc_taumax[j] * c.soft_torque_limit, 0.0f);
        case LSIM_R_JOINT_POWER: return fabsf(x.dof[2 * j + 1]) * fabsf(x.tau[j]);
        case LSIM_R_POWER: return fabsf(x.tau[j] * x.dof[2 * j + 1]);
        case LSIM_R_POWER_DISTRIBUTION: return fabsf(x.tau[j] * x.dof[2 * j + 1]);
        case LSIM_R_STAND_STILL:
        case LSIM_R_STAND_NICE: return fabsf(x.dof[2 * j] - sh.jc_q0[j]);
        case LSIM_R_BASE_HEIGHT:
        case LSIM_R_BASE_HEIGHT_UP: {
            float s = 0.0f;
            if (c.mesh_type != 0) for (int k = 0; k < LSIM_NUM_BASE_HEIGHT_PTS / LS_BH_PARTS; ++k) s += sh.bh[(LSIM_NUM_BASE_HEIGHT_PTS / LS_BH_PARTS) * j + k];
            return s;
        }
        case LSIM_R_HIP_ACTION_MAGNITUDE: { float m = fmaxf(fabsf(x.act[3 * j]) - 1.0f, 0.0f); return m * m; }
        case LSIM_R_HIP_POS: case LSIM_R_HIP_POS_UP: return fabsf(x.dof[2 * (3 * j)] - sh.jc_q0[3 * j]);
        case LSIM_R_THIGH_POSE: case LSIM_R_THIGH_POSE_UP: return fabsf(x.dof[2 * (3 * j + 1)] - sh.jc_q0[3 * j + 1]);
        case LSIM_R_CALF_POSE: case LSIM_R_CALF_POSE_UP: return fabsf(x.dof[2 * (3 * j + 2)] - sh.jc_q0[3 * j + 2]);
        case LSIM_R_FEET_AIR_TIME: {  // LR:1459-1470 (mutates last_contacts and feet_air_time)
            // quirk 2: the reference recomputes contact | last_contacts here AFTER post_physics_step already set last_contacts = contact
            // (LR:207-209), so the filter of this term is the raw contact flag and rewriting last_contacts changes nothing
            const bool contact = sh.cf[ls_foot_body(cx, j)][2] > 1.0f;
            float a = sh.pre_air[j];
            const float first = (a > 0.0f && contact) ? 1.0f : 0.0f;
            a += dt;
            LSB(cx, LSIM_BUF_FEET_AIR_TIME, float)[4 * env + j] = a * (contact ? 0.0f : 1.0f);
            return (a - 0.5f) * first;
        }
        case LSIM_R_FEET_CONTACT_FORCES: { V3 F = v3p(sh.cf[ls_foot_body(cx, j)]); return fmaxf(sqrtf(dot(F, F)) - c.max_contact_force, 0.0f); }
        case LSIM_R_FEET_SLIDE: case LSIM_R_FEET_SLIDE_UP: return ls_foot_slide_part(cx, sh, x, false, j);
        case LSIM_R_FOOT_CLEARANCE_BASE: case LSIM_R_FOOT_CLEARANCE_BASE_UP: return ls_foot_slide_part(cx, sh, x, true, j);
        case LSIM_R_FOOT_CLEARANCE_TERRAIN: case LSIM_R_FOOT_CLEARANCE_TERRAIN_UP: return ls_foot_clearance_terrain_part(cx, sh, fct_shifts, j);
        default: return 0.0f;
    }
}
// a term with parts from its n parts p[0..n) (added in index order: the order of the reference's reductions)
LS_FN float ls_reward_finish(const LsCtx& cx, const WaveShared& sh, int id, const float* p, int n) {
    const lsim_config& c = cx.cfg;
    if (id == LSIM_R_TORQUES_DISTRIBUTION || id == LSIM_R_POWER_DISTRIBUTION) return ls_var12(p);
    float acc = 0.0f;
    for (int j = 0; j < n; ++j) acc += p[j];
    switch (id) {
        case LSIM_R_BASE_HEIGHT:
        case LSIM_R_BASE_HEIGHT_UP: {
            const float bh = c.mesh_type == 0 ? sh.root[2] : acc / 63.0f;
            const float d = bh - c.base_height_target;
            return id == LSIM_R_BASE_HEIGHT ? d * d : d * d * ls_up(sh);
        }
        case LSIM_R_STAND_STILL: return acc * ((ls_cmd_norm(sh) < 0.1f) ? 1.0f : 0.0f);
        case LSIM_R_STAND_NICE: return acc * ((ls_cmd_norm(sh) < 0.1f) ? 1.0f : 0.0f) * (1.0f - sh.grav[2]);
        case LSIM_R_FEET_AIR_TIME: return acc * ((ls_cmd_norm(sh) > 0.1f) ? 1.0f : 0.0f);
        case LSIM_R_HIP_POS_UP: case LSIM_R_THIGH_POSE_UP: case LSIM_R_CALF_POSE_UP: case LSIM_R_FEET_SLIDE_UP:
        case LSIM_R_FOOT_CLEARANCE_BASE_UP: case LSIM_R_FOOT_CLEARANCE_TERRAIN_UP:
            return acc * ls_up(sh);
        default: return acc;
    }
}
// a scalar `_reward_<name>()` (LR:1444-1770)
LS_FN float ls_reward_scalar(const LsCtx& cx, WaveShared& sh, const LsRewCtx& x, int id, int env) {
    const lsim_config& c = cx.cfg;
    float acc = 0.0f;
    switch (id) {
        case LSIM_R_TRACKING_LIN_VEL: {
            float small = ls_cmd_norm(sh) < 0.1f ? 0.0f : 1.0f;
            float ex = sh.cmd[0] * small - sh.blv[0], ey = sh.cmd[1] * small - sh.blv[1];
            return expf(-(ex * ex + ey * ey) / c.tracking_sigma);
        }
        case LSIM_R_TRACKING_ANG_VEL: { float e = sh.cmd[2] - sh.bav[2]; return expf(-(e * e) / c.tracking_sigma); }
        case LSIM_R_UPWARD: return 1.0f - sh.grav[2];
        case LSIM_R_HAS_CONTACT: {
            float n = 0.0f;
            for (int f = 0; f < 4; ++f) n += x.filt[f] ? 1.0f : 0.0f;
            return ((ls_cmd_norm(sh) < 0.1f) ? 1.0f : 0.0f) * n / 4.0f;
        }
        case LSIM_R_LIN_VEL_Z: return sh.blv[2] * sh.blv[2];
        case LSIM_R_LIN_VEL_Z_UP: return sh.blv[2] * sh.blv[2] * ls_up(sh);
        case LSIM_R_ANG_VEL_XY: return sh.bav[0] * sh.bav[0] + sh.bav[1] * sh.bav[1];
        case LSIM_R_ANG_VEL_XY_UP: return (sh.bav[0] * sh.bav[0] + sh.bav[1] * sh.bav[1]) * ls_up(sh);
        case LSIM_R_ORIENTATION: return sh.grav[0] * sh.grav[0] + sh.grav[1] * sh.grav[1];
        case LSIM_R_ORIENTATION_UP: return (sh.grav[0] * sh.grav[0] + sh.grav[1] * sh.grav[1]) * ls_up(sh);
        case LSIM_R_COLLISION:
        case LSIM_R_COLLISION_UP:
            for (int b = 0; b < LS_NB; ++b)
                if ((cx.model.penalised_body_mask >> b) & 1u) { V3 f = v3p(sh.cf[b]); acc += (sqrtf(dot(f, f)) > 0.1f) ? 1.0f : 0.0f; }
            return id == LSIM_R_COLLISION ? acc : acc * ls_up(sh);
        case LSIM_R_TERMINATION: return (sh.reset && !sh.timeout) ? 1.0f : 0.0f;
        case LSIM_R_FEET_STUMBLE: return ls_stumble(cx, sh, env, 5.0f);
        case LSIM_R_FEET_STUMBLE_UP: return ls_stumble(cx, sh, env, 4.0f) * ls_up(sh);
        case LSIM_R_FEET_MIRROR:
        case LSIM_R_FEET_MIRROR_UP: {
            const float* d = x.dof;
            float a1 = d[2] - d[20], a2 = d[4] - d[22], b1 = d[8] - d[14], b2 = d[10] - d[16];
            float r = 0.5f * ((a1 * a1 + a2 * a2) + (b1 * b1 + b2 * b2));
            return id == LSIM_R_FEET_MIRROR ? r : r * ls_up(sh);
        }
        case LSIM_R_STUCK: return ((fabsf(sh.blv[0]) < 0.1f) && (fabsf(sh.cmd[0]) > 0.1f)) ? 1.0f : 0.0f;
        default: return 0.0f;
    }
}

LS_FN LsRewCtx ls_rew_ctx(const WaveShared& sh) {
    LsRewCtx x;
    x.dof = sh.dofs;
    x.act = sh.act;
    x.last_act = sh.last_act;
    x.last_last_act = sh.pre_lla;
    x.last_dof_pos = sh.pre_ldp;
    x.last_dof_vel = sh.pre_ldv;
    x.tau = sh.tau;
    x.last_tau = sh.pre_ltau;
    x.filt = sh.filt;
    return x;
}
// quirk 3: each earlier active foot_clearance_terrain* term has already shifted self.feet_pos by +border
LS_FN int ls_fct_shifts(const LsCtx& cx, int id) {
    return (id == LSIM_R_FOOT_CLEARANCE_TERRAIN_UP && cx.cfg.reward_scales[LSIM_R_FOOT_CLEARANCE_TERRAIN] != 0.0f) ? 2 : 1;
}
// ---- Q5a: the parts of the terms that have them.  Item = (term id << 10) | (term's index in the active list << 4) | part, built by the host
//      (ls_api_impl.h) and staged in LDS by ph_load_a; same-term items are neighbours, so a pass of 64 lanes runs only a few term bodies.
LS_FN void ph_reward_parts(const LsCtx& cx, WaveShared& sh, const uint16_t* items, int lane, int env) {
    const LsRewCtx x = ls_rew_ctx(sh);
    for (int it = 0; it < (LS_MAX_PART_ITEMS + 63) / 64; ++it) {
        const int k = lane + 64 * it;
        if (64 * it >= cx.num_part_items) break;
        if (k >= cx.num_part_items) continue;
        const int item = items[k];
        const int id = item >> 10, ai = (item >> 4) & 63, j = item & 15;
        sh.u.r.rj[ai][j] = ls_reward_part(cx, sh, x, id, j, env, ls_fct_shifts(cx, id));
    }
}
// ---- Q5b: compute_reward (LR:363-380).  Lane i owns active term i; lane 0 then accumulates in the reference's order.
LS_FN void ph_reward_terms(const LsCtx& cx, WaveShared& sh, const LaneRegs& rg, int lane, int env) {
    if (lane >= cx.num_active) return;
    const int id = rg.term_id;                   // cx.active_terms[lane], fetched by ph_late_load
    const LsRewCtx x = ls_rew_ctx(sh);
    const int n = ls_reward_num_parts(id);
    float v;
    if (n == 0) v = ls_reward_scalar(cx, sh, x, id, env);
    else {
        float* p = sh.u.r.rj[lane];
        if (!((cx.parted_mask >> lane) & 1ull))      // more parts than the item table holds (every term switched on): this lane does its own
            for (int j = 0; j < n; ++j) p[j] = ls_reward_part(cx, sh, x, id, j, env, ls_fct_shifts(cx, id));
        v = ls_reward_finish(cx, sh, id, p, n);
    }
    v *= rg.term_scale;                          // cx.cfg.reward_scales[id]
    sh.rewv[lane] = v;
    const float es = sh.pre_es[id] + v;
    sh.pre_es[id] = es;
    LSB(cx, LSIM_BUF_EPISODE_SUMS, float)[env * LSIM_NUM_REWARD_TERMS + id] = es;
}
LS_FN void ph_reward_total(const LsCtx& cx, WaveShared& sh, int lane, int env) {
    if (lane != 0) return;
    const lsim_config& c = cx.cfg;
    float rew = 0.0f;
    for (int i = 0; i < cx.num_active; ++i) rew += sh.rewv[i];
    if (c.only_positive_rewards) rew = fmaxf(rew, 0.0f);
    if (c.reward_scales[LSIM_R_TERMINATION] != 0.0f) {
        float v = ((sh.reset && !sh.timeout) ? 1.0f : 0.0f) * c.reward_scales[LSIM_R_TERMINATION];
        rew += v;
        const float es = sh.pre_es[LSIM_R_TERMINATION] + v;
        sh.pre_es[LSIM_R_TERMINATION] = es;            // the fused tail (ls_kernels.h) takes the episode sums of a resetting env from LDS
        LSB(cx, LSIM_BUF_EPISODE_SUMS, float)[env * LSIM_NUM_REWARD_TERMS + LSIM_R_TERMINATION] = es;
    }
    LSB(cx, LSIM_BUF_REW, float)[env] = rew;
}

// ---------------------------------------------------------------------------------------------- observations
// entry e of the 238-vector of compute_observations (LR:382-401) before noise
LS_FN float ls_obs_entry(const LsCtx& cx, const WaveShared& sh, int e, const float* dof, const float* act, const float* heights, const float* q0) {
    const lsim_config& c = cx.cfg;
    if (e < 2) return sh.cmd[e] * c.obs_scale_lin_vel;
    if (e == 2) return sh.cmd[2] * c.obs_scale_ang_vel;
    if (e < 6) return sh.bav[e - 3] * c.obs_scale_ang_vel;
    if (e < 9) return sh.grav[e - 6];
    if (e < 21) return (dof[2 * (e - 9)] - q0[e - 9]) * c.obs_scale_dof_pos;
    if (e < 33) return dof[2 * (e - 21) + 1] * c.obs_scale_dof_vel;
    if (e < 45) return act[e - 33];
    if (e < 48) return sh.blv[e - 45] * c.obs_scale_lin_vel;
    if (e < 51) return sh.disturbance[e - 48];
    return clampf(sh.root[2] - 0.5f - heights[e - 51], -1.0f, 1.0f) * c.obs_scale_height;
}
LS_FN float ls_noise_scale(const lsim_config& c, int idx) {  // noise_scale_vec (LR:883-910) by draw index
    if (idx < 3) return 0.0f;
    if (idx < 6) return c.noise_vec_ang_vel;
    if (idx < 9) return c.noise_vec_gravity;
    if (idx < 21) return c.noise_vec_dof_pos;
    if (idx < 33) return c.noise_vec_dof_vel;
    if (idx < 45) return 0.0f;
    return c.noise_vec_height;
}
// lanes 0..57 own one Philox block (4 noisy entries) each, lanes 58..63 the six noise-free entries 45..50
// q0: the default joint angles -- kernel A's copy in LDS (sh.jc_q0), or the config's in global memory (kernel B, which stages no joint constants)
LS_FN void ph_build_obs(const LsCtx& cx, WaveShared& sh, int lane, int env, uint32_t stepw, uint32_t tag, float* out, const float* q0) {
    const lsim_config& c = cx.cfg;
    const float* dof = sh.dofs;
    const float* act = sh.act;
    if (lane >= 58) {
        int e = 45 + (lane - 58);
        out[e] = ls_obs_entry(cx, sh, e, dof, act, sh.heights, q0);
        return;
    }
    float u[4];
    ls_u01x4(c.seed, c.rank, (uint32_t)env, stepw, tag, (uint32_t)lane, u);
    for (int i = 0; i < 4; ++i) {
        int idx = 4 * lane + i;
        int e = idx < 45 ? idx : idx + 6;
        if (e >= LSIM_NUM_PRIV_OBS) break;
        if (e >= 51 && !c.measure_heights) { out[e] = 0.0f; continue; }
        float v = ls_obs_entry(cx, sh, e, dof, act, sh.heights, q0);
        if (idx >= 45 || c.add_noise) v += (2.0f * u[i] - 1.0f) * ls_noise_scale(c, idx);  // LR:400 is not gated by add_noise
        out[e] = v;
    }
}
