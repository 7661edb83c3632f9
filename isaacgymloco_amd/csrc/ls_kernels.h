// ls_kernels.h -- the two per-step kernels as sequences of wave phases (one wavefront = one robot).
//
//   kernel A  ls_wave_step_a : LR:129-152 (clip, delay model, 4 x {PD torque, physics sub-step}) + LR:187-228
//                              (derived state, callback, termination, rewards, termination observations)
//   kernel B  ls_wave_step_b : LR:229-241 (reset_idx, compute_observations, last_* roll) + LR:167-171 (clip)
//   The kernel boundary between A and B is the one global dependency of the step: whether ANY env reset
//   (extras["time_outs"], LR:358) and the command-curriculum reduction over the reset set (LR:307-308, 868-880).
//
// LS_PHASE(call): on the GPU every lane runs `call` then the wave synchronises on LDS; under LS_EMU (tests only)
// the 64 lanes are looped.  `lane`, `rg` (LaneRegs&) are in scope inside `call`.
#pragma once
#include "ls_physics.h"
#include "ls_post.h"

#if defined(LS_EMU)
#define LS_LANES_PARAM LaneRegs* L
#define LS_PHASE(call) do { for (int lane = 0; lane < 64; ++lane) { LaneRegs& rg = L[lane]; (void)rg; call; } } while (0)
#define LS_COLLECTIVE(gpu_call, emu_call) do { emu_call; } while (0)
#define LS_KINEMATICS() LS_PHASE(ph_kinematics(sh, lane))
#define LS_TORQUES_KINEMATICS() LS_PHASE(ph_torques(cx, sh, lane, env, sub, a.flags); ph_kinematics(sh, lane))
#define LS_ATOMIC_ADD(ptr, v) (*(ptr) += (v))
#define LS_ATOMIC_ADD_I64(ptr, v) (*(ptr) += (v))
#define LS_ATOMIC_FETCH_ADD_I64(ptr, v) ls_emu_fetch_add((ptr), (v))
#define LS_ATOMIC_READ_I64(ptr) (*(ptr))
#define LS_THREADFENCE() do { } while (0)
static inline long long ls_emu_fetch_add(long long* p, long long v) { long long o = *p; *p = o + v; return o; }
#define LS_WAVE_FN static inline
#define LS_LDS_FENCE() do { } while (0)
#define LS_SETPRIO(n) do { } while (0)
#define LS_TICK_INIT() do { } while (0)
#define LS_TICK_FLUSH() do { } while (0)
#define LS_CP(i) do { } while (0)
#else
// Every phase re-derives its lane id from an opaque copy: the compiler then cannot hoist the dozens of per-phase lane
// predicates and LDS addresses out of the sub-step loop (it did, and spilled ~50 VGPRs + 128 SGPRs to keep them alive).
__device__ __forceinline__ int ls_opaque_lane(int l) { asm volatile("" : "+v"(l)); return l; }
#define LS_LANES_PARAM LaneRegs& rg, const int lane0
#if defined(LS_PHASE_TIMING)   // diagnostics build only (tools/phase_profile.py): shader-clock ticks per phase site, summed over waves
__device__ unsigned long long g_ls_phase_ticks[128];
__device__ unsigned long long g_ls_phase_calls[128];
__device__ unsigned long long g_ls_phase_ticks_by[3][129];    // the same by kind of wave: [0] at most 3 contacts, [1] 6 or more, [2] resetting; [k][128] = waves
#define LS_TICK(site) do { if (lane0 == 0) { unsigned long long t_ = clock64(); ls_ticks[(site) & 127] += (unsigned int)(t_ - ls_t_prev); ls_calls[(site) & 127] += 1; \
                                             ls_t_prev = t_; } } while (0)
#define LS_TICK_INIT() __shared__ unsigned int ls_ticks[128]; __shared__ unsigned short ls_calls[128]; \
                       ls_ticks[lane0] = 0; ls_calls[lane0] = 0; ls_ticks[64 + lane0] = 0; ls_calls[64 + lane0] = 0; __syncthreads(); unsigned long long ls_t_prev = clock64()
#define LS_TICK_FLUSH() do { __syncthreads(); const int kind_ = sh.reset ? 2 : (sh.nact_max >= 6 ? 1 : (sh.nact_max <= 3 ? 0 : -1)); \
                             for (int s_ = lane0; s_ < 128; s_ += 64) if (ls_calls[s_]) { atomicAdd(&g_ls_phase_ticks[s_], (unsigned long long)ls_ticks[s_]); \
                                                                      atomicAdd(&g_ls_phase_calls[s_], (unsigned long long)ls_calls[s_]); \
                                                                      if (kind_ >= 0) atomicAdd(&g_ls_phase_ticks_by[kind_][s_], (unsigned long long)ls_ticks[s_]); } \
                             if (lane0 == 0 && kind_ >= 0) atomicAdd(&g_ls_phase_ticks_by[kind_][128], 1ull); } while (0)
#else
#define LS_TICK(site) do { } while (0)
#define LS_TICK_INIT() do { } while (0)
#define LS_TICK_FLUSH() do { } while (0)
#endif
#if defined(LS_WAVE_TIMES)
__device__ unsigned long long g_ls_wave_times[4 * 65536];     // per env: start, end (100 MHz wall clock), shader-clock ticks, HW_ID | XCC_ID << 16
__device__ unsigned int g_ls_wave_cp[16 * 65536];              // per env: shader-clock ticks since the wave's start at up to 16 checkpoints (LS_CP)
#if LS_WAVE_TIMES == 2      // the 16 checkpoints behind the phases of sub-step 1 instead (tools/wave_times.py --phases)
#define LS_CP(i) do { } while (0)
#define LS_SUBCP() do { if (ls_sub == 1 && ls_k < 16) ls_cp[ls_k++] = (unsigned int)(clock64() - ls_wc0); } while (0)
#else
#define LS_CP(i) do { ls_cp[i] = (unsigned int)(clock64() - ls_wc0); } while (0)
#define LS_SUBCP() do { } while (0)
#endif
#else
#define LS_CP(i) do { } while (0)
#define LS_SUBCP() do { } while (0)
#endif
#if defined(LS_PHASE_MARKS)    // diagnostics only (tools/phase_static.py): a comment in the assembly behind every phase site
#define LS_STR2(x) #x
#define LS_STR(x) LS_STR2(x)
#define LS_MARK() asm volatile("; LS_MARK " LS_STR(__LINE__))
#else
#define LS_MARK() do { } while (0)
#endif
#if defined(LS_EXP_TWICE)      // diagnostics only (tools/phase_cost.py): the phase at source line LS_EXP_TWICE runs twice -- its marginal cost is the
#define LS_AGAIN(call) do { if (__LINE__ == LS_EXP_TWICE) { { const int lane = ls_opaque_lane(lane0); call; } LS_WAVE_SYNC(); } } while (0)   /* change of the kernel time (idempotent phases only) */
#else
#define LS_AGAIN(call) do { } while (0)
#endif
#define LS_PHASE(call) do { { const int lane = ls_opaque_lane(lane0); call; } LS_WAVE_SYNC(); LS_AGAIN(call); LS_MARK(); LS_TICK(__LINE__ - ls_line0); LS_SUBCP(); } while (0)
#define LS_COLLECTIVE(gpu_call, emu_call) do { { const int lane = ls_opaque_lane(lane0); gpu_call; } LS_WAVE_SYNC(); LS_AGAIN(gpu_call); LS_MARK(); LS_TICK(__LINE__ - ls_line0); LS_SUBCP(); } while (0)
#define LS_KINEMATICS() LS_COLLECTIVE(wc_kinematics(sh, lane), (void)0)
#define LS_TORQUES_KINEMATICS() LS_COLLECTIVE(ph_torques(cx, sh, lane, env, sub, a.flags); wc_kinematics(sh, lane), (void)0)
#define LS_ATOMIC_ADD(ptr, v) atomicAdd((ptr), (v))
// 64-bit integer atomics on the fixed-point accumulators (device scope: they are performed at the memory side, coherent across the XCDs)
#define LS_ATOMIC_ADD_I64(ptr, v) ((void)atomicAdd((unsigned long long*)(ptr), (unsigned long long)(v)))
#define LS_ATOMIC_FETCH_ADD_I64(ptr, v) ((long long)atomicAdd((unsigned long long*)(ptr), (unsigned long long)(v)))
#define LS_ATOMIC_READ_I64(ptr) ((long long)atomicAdd((unsigned long long*)(ptr), 0ull))
#define LS_THREADFENCE() __threadfence()
#define LS_WAVE_FN __device__ __forceinline__
#define LS_LDS_FENCE() LS_WAVE_SYNC()        // inside a phase, in wave-uniform control flow only
#define LS_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif

// Order-independent reductions over waves (float atomics made extras["episode"] and the command-curriculum decision depend on the
// arrival order of the waves): values are added as 2^-32 fixed-point int64, integer addition being associative.  One value is clamped to
// +-2^20 (an episode sum of a million is not a reward any more), so a single conversion cannot overflow and the int64 sum holds
// 2^63 / 2^52 = 2048 envs AT the clamp, or every env of a 2^22-env batch at |sum| <= 512 -- round 2's 2^-40 scale left a factor 1.6 at
// N = 262 144 with |sum| = 20.  The conversion error 2^-33 is far below the fp32 resolution of the sums that are reported.
#define LS_FIX_SCALE 4294967296.0f
LS_FN long long ls_to_fix(float v) { return (long long)llrintf(fminf(fmaxf(v, -1048576.0f), 1048576.0f) * LS_FIX_SCALE); }
LS_FN float ls_from_fix(long long f) { return (float)((double)f * (1.0 / 4294967296.0)); }
LS_FN long long* ls_fix_row(const LsCtx& cx, int row) { return (long long*)(cx.accum + row * LSIM_STATS_SIZE + LSIM_STATS_FIX); }

// ---- load the robot's state into LDS, clip the actions (LR:129-130), draw the action delay (LR:134)
// Load phases: every global load is issued first, unconditionally -- a lane with no use for a value reads element 0 of the same row, always
// a valid address -- and the LDS writes follow.  `if (lane < n) sh.x[lane] = buf[...]` per quantity compiles to one branch per quantity with
// the load AND its wait inside, i.e. a chain of ~20 serialised memory round trips (12 k of the 27 k ticks of a kernel-B wave).
LS_FN void ph_load_a(const LsCtx& cx, WaveShared& sh, LaneRegs& rg, int lane, int env, const LsStepArgs& a) {
    const lsim_config& c = cx.cfg;
    const int l12 = lane < 12 ? lane : 0;
    const float v_root = LSB(cx, LSIM_BUF_ROOT_STATES, float)[13 * env + (lane < 13 ? lane : 0)];
    const float v_q = LSB(cx, LSIM_BUF_DOF_STATE, float)[24 * env + 2 * l12];
    const float v_qd = LSB(cx, LSIM_BUF_DOF_STATE, float)[24 * env + 2 * l12 + 1];
    const float v_act = LS_G(const float, a.actions)[12 * env + l12];
    const float v_last = LSB(cx, LSIM_BUF_LAST_ACTIONS, float)[12 * env + l12];
    const float v_ms = LSB(cx, LSIM_BUF_MOTOR_STRENGTH, float)[12 * env + l12];
    // per-env scalars: the same address in every lane
    const float v_kpf = LSB(cx, LSIM_BUF_KP_FACTORS, float)[env], v_kdf = LSB(cx, LSIM_BUF_KD_FACTORS, float)[env];
    const float v_fric = LSB(cx, LSIM_BUF_FRICTION, float)[env], v_payload = LSB(cx, LSIM_BUF_PAYLOAD, float)[env];
    const int l3 = lane < 3 ? lane : 0;
    const float v_comd = LSB(cx, LSIM_BUF_COM_DISPLACEMENT, float)[3 * env + l3];
    const float v_pend = LSB(cx, LSIM_BUF_PENDING_FORCE, float)[3 * env + l3];
    // inputs of the post-physics stack (see WaveShared::pre_*): their latency hides behind the physics, and no load has to queue behind the
    // state stores that follow the last sub-step
    const int which = lane < 48 ? lane / 12 : 0, j = lane < 48 ? lane - 12 * which : 0;
    LS_GLOBAL const float* src = LSB(cx, which == 0 ? LSIM_BUF_LAST_LAST_ACTIONS : (which == 1 ? LSIM_BUF_LAST_DOF_POS : (which == 2 ? LSIM_BUF_LAST_DOF_VEL : LSIM_BUF_LAST_TORQUES)), const float);
    const float v_pre = src[12 * env + j];
    const int l4 = lane & 3;
    const float v_cmd = LSB(cx, LSIM_BUF_COMMANDS, float)[4 * env + l4];
    const float v_air = LSB(cx, LSIM_BUF_FEET_AIR_TIME, float)[4 * env + l4];
    const int v_eplen = (int)LSB(cx, LSIM_BUF_EPISODE_LENGTH, int64_t)[env];
    const unsigned int v_lc = LSB(cx, LSIM_BUF_LAST_CONTACTS, unsigned int)[env];
    const int v_level = (int)LSB(cx, LSIM_BUF_TERRAIN_LEVELS, int64_t)[env];
    const int v_type = (int)LSB(cx, LSIM_BUF_TERRAIN_TYPES, int64_t)[env];
    const float v_org = LSB(cx, LSIM_BUF_ENV_ORIGINS, float)[3 * env + l3];
    const float v_es = LSB(cx, LSIM_BUF_EPISODE_SUMS, float)[env * LSIM_NUM_REWARD_TERMS + (lane < LSIM_NUM_REWARD_TERMS ? lane : 0)];
    const float v_rng = LS_G(const float, cx.accum)[a.row_in * LSIM_STATS_SIZE + LSIM_STATS_CMD_RANGES + (lane & 7)];   // live command ranges (no kernel of this step writes row_in's)
    const float v_mp = lane < LSIM_MAX_HEIGHT_PTS_X ? c.measured_points_x[lane] : c.measured_points_y[lane - LSIM_MAX_HEIGHT_PTS_X];
    static_assert(LSIM_MAX_HEIGHT_PTS_X + LSIM_MAX_HEIGHT_PTS_Y == 64, "one lane per measured-point coordinate");
    // lsim_config.lin_vel_at_com: the root tensor's linear velocity is the centre of mass's (PhysX); the dynamics work on the link origin's,
    // v_origin = v_com - w x (R c).  Lane 14 fetches what that needs (the same cache line as v_root) with everything else and converts below:
    // until round 6 this was a one-lane phase of its own between the load and the first kinematics phase (an LDS round trip + a barrier per step)
    const bool to_origin = c.lin_vel_at_com != 0 && !(a.flags & LSIM_STEP_SKIP_PHYSICS);
    float v_r10[10], v_cd3[3];
    for (int k = 0; k < 10; ++k) v_r10[k] = LSB(cx, LSIM_BUF_ROOT_STATES, float)[13 * env + 3 + k];
    for (int k = 0; k < 3; ++k) v_cd3[k] = LSB(cx, LSIM_BUF_COM_DISPLACEMENT, float)[3 * env + k];
    float v_jc[7];
    {
        const lsim_robot_model& m = cx.model;
        v_jc[0] = c.default_dof_pos[l12]; v_jc[1] = c.p_gains[l12]; v_jc[2] = c.d_gains[l12]; v_jc[3] = c.torque_limits[l12];
        v_jc[4] = m.dof_pos_lower[l12]; v_jc[5] = m.dof_pos_upper[l12]; v_jc[6] = m.dof_vel_limit[l12];
    }
    // ---- LDS writes
    // the tail of the row array Y that no spatial inertia ever overlays (ls_shared.h, union u): the solver reads whole triples of limit slots,
    // i.e. up to two slots that hold no row yet, and multiplies what it finds by a zero impulse -- harmless only if it is finite
    static_assert(sizeof(sh.u.I6) <= sizeof(sh.u.c.Y) && sizeof(sh.u.c.Y) - sizeof(sh.u.I6) <= 2 * LS_NV * sizeof(float), "rows of Y beyond the inertias: at most the two zeroed here");
    if (lane < 2 * LS_NV) (&sh.u.c.Y[LS_MAXR - 2][0])[lane] = 0.0f;
    if (lane < LSIM_MAX_HEIGHT_PTS_X) sh.mpx[lane] = v_mp; else sh.mpy[lane - LSIM_MAX_HEIGHT_PTS_X] = v_mp;
    if (lane < 13 && !(to_origin && lane >= 7 && lane < 10)) sh.root[lane] = v_root;
    if (to_origin && lane == 14) {
        const lsim_body& b0 = cx.model.bodies[0];
        const V3 cl = v3(b0.com[0] + v_cd3[0], b0.com[1] + v_cd3[1], b0.com[2] + v_cd3[2]);            // ls_body_com_local(sh, 0)
        const V3 r = quat_apply(v_r10, cl);
        const V3 v = v3p(v_r10 + 4) - cross(v3p(v_r10 + 7), r);
        v3st(sh.root + 7, v);
    }
    if (lane < 12) {
        sh.jc_q0[lane] = v_jc[0]; sh.jc_kp[lane] = v_jc[1]; sh.jc_kd[lane] = v_jc[2]; sh.jc_taumax[lane] = v_jc[3];
        sh.jc_lo[lane] = v_jc[4]; sh.jc_hi[lane] = v_jc[5]; sh.jc_vmax[lane] = v_jc[6];
        sh.q[lane] = v_q;
        sh.qd[lane] = v_qd;
        const float act = clampf(v_act, -c.clip_actions, c.clip_actions);    // LR:129-130
        sh.act[lane] = act;
        LSB(cx, LSIM_BUF_ACTIONS, float)[12 * env + lane] = act;
        sh.last_act[lane] = v_last;
        sh.ms[lane] = v_ms;
    }
    if (lane == 13) {
        sh.kpf = v_kpf;
        sh.kdf = v_kdf;
        sh.mu = 0.5f * (c.terrain_friction + v_fric);   // PhysX default combine mode: average
        sh.payload = v_payload;
    }
    if (lane < 3) {
        sh.comd[lane] = v_comd;
        sh.pend[lane] = v_pend;
        sh.pre_org[lane] = v_org;
        if (!(a.flags & LSIM_STEP_SKIP_PHYSICS)) LSB(cx, LSIM_BUF_PENDING_FORCE, float)[3 * env + lane] = 0.0f;   // consumed by the first sub-step
    }
    if (lane == 15) {
        int delay = (int)(ls_draw(cx, env, (uint32_t)a.step_counter, LSIM_RNG_DELAY, 0) * (float)c.decimation);   // LR:134
        sh.delay = delay;
        sh.nact = 0; sh.nact_max = 0;
        LSB(cx, LSIM_BUF_DELAY_STEPS, int32_t)[env] = delay;
    }
    if (lane >= 16 && lane < 16 + LS_NB) {
        const lsim_body& b = cx.model.bodies[lane - 16];
        LsBodyLds& o = sh.body[lane - 16];
        o.mass = b.mass;
        for (int k = 0; k < 3; ++k) { o.com[k] = b.com[k]; o.jpos[k] = b.joint_pos[k]; o.axis[k] = b.joint_axis[k]; }
        for (int k = 0; k < 6; ++k) o.inertia[k] = b.inertia[k];
    }
    if (lane < 48) sh.pre4[lane] = v_pre;               // pre_lla .. pre_ltau
    else if (lane < 52) sh.pre_cmd[lane - 48] = v_cmd;
    else if (lane < 56) sh.pre_air[lane - 52] = v_air;
    else if (lane == 56) sh.pre_eplen = v_eplen;
    else if (lane == 57) sh.pre_lc = v_lc;
    else if (lane == 58) sh.pre_level = v_level;
    else if (lane == 59) sh.pre_type = v_type;
    if (lane < LSIM_NUM_REWARD_TERMS) sh.pre_es[lane] = v_es;
    if (lane < 8) sh.ranges[lane] = v_rng;
    rg.cp_active = 0;
    rg.row_kind = -1;
}

// ---- LeggedRobot._compute_torques (LR:658-688) with the delayed action of sub-step `sub` (LR:138); lane = dof
LS_FN void ph_torques(const LsCtx& cx, WaveShared& sh, int lane, int env, int sub, uint32_t flags) {
    if (lane >= 12) return;
    const lsim_config& c = cx.cfg;
    float act = sh.act[lane], last = sh.last_act[lane];
    float a = c.delay ? last + (act - last) * ((sub >= sh.delay) ? 1.0f : 0.0f) : act;
    a = sh.ms[lane] * a;
    float as = a * c.action_scale;
    if (lane % 3 == 0) as *= c.hip_reduction;
    float target = sh.jc_q0[lane] + as;
    float q = sh.q[lane], qd = sh.qd[lane], t;
    if (c.control_type == 0) t = sh.jc_kp[lane] * sh.kpf * (target - q) - sh.jc_kd[lane] * sh.kdf * qd;
    else if (c.control_type == 1) t = sh.jc_kp[lane] * (as - qd) - sh.jc_kd[lane] * (qd - sh.pre_ldv[lane]) / c.sim_dt;
    else t = as;
    t = clampf(t, -sh.jc_taumax[lane], sh.jc_taumax[lane]);
    sh.tau[lane] = t;
    if (flags & LSIM_STEP_RECORD_SUBSTEPS) LSB(cx, LSIM_BUF_SUBSTEP_TORQUES, float)[12 * (c.decimation * env + sub) + lane] = t;   // test hook (wave-uniform branch)
}

// ---- after the last sub-step: publish the simulator state tensors (LR:187-190 refresh_* equivalents)
LS_FN void ph_store_sim_state(const LsCtx& cx, WaveShared& sh, int lane, int env) {
    if (lane < 13) LSB(cx, LSIM_BUF_ROOT_STATES, float)[13 * env + lane] = sh.root[lane];
    if (lane < 12) {
        LS_GLOBAL float* dof = LSB(cx, LSIM_BUF_DOF_STATE, float) + 24 * env;
        dof[2 * lane] = sh.q[lane]; dof[2 * lane + 1] = sh.qd[lane];
        sh.dofs[2 * lane] = sh.q[lane]; sh.dofs[2 * lane + 1] = sh.qd[lane];
        LSB(cx, LSIM_BUF_TORQUES, float)[12 * env + lane] = sh.tau[lane];
    }
    if (lane < 3 * LS_NB) LSB(cx, LSIM_BUF_CONTACT_FORCES, float)[3 * LS_NB * env + lane] = sh.cf[lane / 3][lane % 3];
    if (lane == 52) { LSB(cx, LSIM_BUF_CONTACT_COUNT, int32_t)[2 * env] = sh.nact_max; LSB(cx, LSIM_BUF_CONTACT_COUNT, int32_t)[2 * env + 1] = sh.nact; }
}
// A robot whose root / joint state holds a NaN or an infinity after the last sub-step (a solver blow-up; it stays that way until the episode times out):
// +1 in this step's stats row and in the running total of LSIM_BUF_NONFINITE.  Integer-valued float / integer atomics: exact in any order; issued only
// by the waves concerned (none in a healthy run).
LS_FN bool ls_not_finite(float v) { return (ls_float_bits(v) & 0x7f800000u) == 0x7f800000u; }
LS_FN void ls_report_nonfinite(const LsCtx& cx, const LsStepArgs& a) {
    LS_ATOMIC_ADD(LS_G(float, cx.accum) + a.row_out * LSIM_STATS_SIZE + LSIM_STATS_NONFINITE, 1.0f);
    LS_ATOMIC_ADD_I64(LSB(cx, LSIM_BUF_NONFINITE, long long), 1);
    LSB(cx, LSIM_BUF_NONFINITE, long long)[1] = (long long)a.step_counter;       // racing writers store the same value
}
#if defined(LS_EMU)
static inline void emu_count_nonfinite(const LsCtx& cx, WaveShared& sh, const LsStepArgs& a) {
    bool bad = false;
    for (int k = 0; k < 13; ++k) bad |= ls_not_finite(sh.root[k]);
    for (int k = 0; k < 12; ++k) bad |= ls_not_finite(sh.q[k]) || ls_not_finite(sh.qd[k]);
    if (bad) ls_report_nonfinite(cx, a);
}
#else
__device__ __forceinline__ void wc_count_nonfinite(const LsCtx& cx, WaveShared& sh, int lane, const LsStepArgs& a) {
    const float v = lane < 13 ? sh.root[lane] : (lane >= 16 && lane < 28 ? sh.q[lane - 16] : (lane >= 32 && lane < 44 ? sh.qd[lane - 32] : 0.0f));
    if (__ballot(ls_not_finite(v)) != 0ull && lane == 0) ls_report_nonfinite(cx, a);
}
#endif
// the rigid-body state rows into registers (LaneRegs::bs) while the kinematics arrays are still alive -- the height samples overlay them --
// and stored by ph_store_body_states once everything the wave had requested from memory has been consumed: a load whose result is read
// while stores are in flight waits for the stores too (vmcnt retires in order), and at 4096 robots in lockstep a store burst takes ~5 us
LS_FN void ph_body_states_all(const LsCtx& cx, WaveShared& sh, LaneRegs& rg, int lane) {
    const bool at_com = cx.cfg.lin_vel_at_com != 0;
    ph_body_states(sh, lane, rg.bs, at_com);
    if (at_com && lane == 0)   // the root state tensor's linear velocity is row 0's: from here on (stores, post-physics, pushes) the centre of mass's
        for (int k = 0; k < 3; ++k) sh.root[7 + k] = rg.bs[7 + k];
    if (lane < LS_NB)       // the feet rows of the tensor for the reward terms, from registers
        for (int f = 0; f < 4; ++f)
            if (cx.model.feet_bodies[f] == lane) {         // uniform f: scalar loads
                for (int k = 0; k < 3; ++k) { sh.feet[f][k] = rg.bs[k]; sh.feet[f][3 + k] = rg.bs[7 + k]; }
            }
}
LS_FN void ph_store_body_states(const LsCtx& cx, const LaneRegs& rg, int lane, int env) {
    if (lane >= LS_NB) return;
    LS_GLOBAL float* o = LSB(cx, LSIM_BUF_RIGID_BODY_STATES, float) + 13 * (LS_NB * env + lane);
    for (int k = 0; k < 13; ++k) o[k] = rg.bs[k];
}
// LSIM_STEP_SKIP_PHYSICS: take the simulator tensors as injected by the caller
LS_FN void ph_load_injected(const LsCtx& cx, WaveShared& sh, int lane, int env) {
    if (lane < 12) {
        sh.dofs[2 * lane] = sh.q[lane]; sh.dofs[2 * lane + 1] = sh.qd[lane];
        LSB(cx, LSIM_BUF_TORQUES, float)[12 * env + lane] = sh.tau[lane];
    }
    if (lane < 3 * LS_NB) sh.cf[lane / 3][lane % 3] = LSB(cx, LSIM_BUF_CONTACT_FORCES, float)[3 * LS_NB * env + lane];
    if (lane < 24) {
        int f = lane / 6, k = lane % 6;
        LS_GLOBAL const float* bs = LSB(cx, LSIM_BUF_RIGID_BODY_STATES, float) + 13 * (LS_NB * env + ls_foot_body(cx, f));
        sh.feet[f][k] = k < 3 ? bs[k] : bs[7 + k - 3];
    }
}

// the per-step reductions over the resetting envs that the command curriculum and the finish of the step read (lane 0 of a resetting env)
LS_FN void ls_count_reset(const LsCtx& cx, WaveShared& sh, const LsStepArgs& a) {
    LS_GLOBAL float* acc = LS_G(float, cx.accum) + a.row_out * LSIM_STATS_SIZE;
    LS_ATOMIC_ADD(acc + LSIM_STATS_RESET_COUNT, 1.0f);                                   // integer-valued: exact in any order
    LS_ATOMIC_ADD_I64(ls_fix_row(cx, a.row_out) + LSIM_STATS_FIX_TRACK, ls_to_fix(sh.pre_es[LSIM_R_TRACKING_LIN_VEL]));   // updated by ph_reward_terms
}
// ---- termination observations / terminal AMP states of the pre-reset state + per-step reductions (LR:227-228)
LS_FN void ph_term_outputs(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a) {
    if (lane < 13) LSB(cx, LSIM_BUF_ROOT_STATES, float)[13 * env + lane] = sh.root[lane];   // push may have changed the velocity
    if (!LS_UNIFORM(sh.reset)) return;
    LS_GLOBAL float* tp = LSB(cx, LSIM_BUF_TERM_PRIV_OBS, float) + LSIM_NUM_PRIV_OBS * env;
    LS_STRIDED(k, lane, LSIM_NUM_PRIV_OBS) tp[k] = sh.cur[k];
    if (lane < LSIM_NUM_AMP_OBS) {
        float v;
        if (lane < 12) v = sh.dofs[2 * lane];
        else if (lane < 15) v = sh.blv[lane - 12];
        else if (lane < 18) v = sh.bav[lane - 15];
        else v = sh.dofs[2 * (lane - 18) + 1];
        LSB(cx, LSIM_BUF_TERM_AMP_OBS, float)[LSIM_NUM_AMP_OBS * env + lane] = v;
    }
    // with the fused tail the two reductions wait for ph_tail_episode_stats: the resetting waves of a step queue up on these two words, an
    // atomic is acknowledged when it has been performed, and the loads of the reset path (the terrain under the new pose) would retire behind
    // them -- 40 k of a resetting wave's 325 k ticks (round 4)
    if (lane == 0 && !(a.flags & LSIM_STEP_NO_RESET) && !a.fuse_tail) ls_count_reset(cx, sh, a);
}

// =============================================================================================== kernel B
LS_FN void ph_load_b(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a) {
    const lsim_config& c = cx.cfg;
    // ---- loads (see ph_load_a): everything this kernel reads, including what its store phase needs, before any of its stores is in flight
    const float v_root = LSB(cx, LSIM_BUF_ROOT_STATES, float)[13 * env + (lane < 13 ? lane : 0)];
    const float v_dof = LSB(cx, LSIM_BUF_DOF_STATE, float)[24 * env + (lane < 24 ? lane : 0)];
    const int l12 = lane < 12 ? lane : 0, l3 = lane < 3 ? lane : 0;
    const float v_act = LSB(cx, LSIM_BUF_ACTIONS, float)[12 * env + l12];
    const float v_blv = LSB(cx, LSIM_BUF_BASE_LIN_VEL, float)[3 * env + l3];
    const float v_bav = LSB(cx, LSIM_BUF_BASE_ANG_VEL, float)[3 * env + l3];
    const float v_grav = LSB(cx, LSIM_BUF_PROJECTED_GRAVITY, float)[3 * env + l3];
    const float v_pend = LSB(cx, LSIM_BUF_PENDING_FORCE, float)[3 * env + l3];
    const float v_cmd = LSB(cx, LSIM_BUF_COMMANDS, float)[4 * env + (lane & 3)];
    const float v_la = LSB(cx, LSIM_BUF_LAST_ACTIONS, float)[12 * env + l12];
    const float v_tq = LSB(cx, LSIM_BUF_TORQUES, float)[12 * env + l12];
    float v_h[(LS_NHP + 63) / 64], v_o[(LSIM_NUM_OBS - LSIM_ONE_STEP_OBS + 63) / 64];
    for (int it = 0; it < (LS_NHP + 63) / 64; ++it) {
        const int k = lane + 64 * it;
        v_h[it] = LSB(cx, LSIM_BUF_MEASURED_HEIGHTS, float)[LS_NHP * env + (k < LS_NHP ? k : 0)];
    }
    for (int it = 0; it < (LSIM_NUM_OBS - LSIM_ONE_STEP_OBS + 63) / 64; ++it) {   // observation history: the 225 values that shift by one frame (LR:403)
        const int k = lane + 64 * it;
        v_o[it] = LSB(cx, LSIM_BUF_OBS, float)[LSIM_NUM_OBS * env + (k < LSIM_NUM_OBS - LSIM_ONE_STEP_OBS ? k : 0)];
    }
    // per-env / per-step scalars: the same address in every lane
    LS_GLOBAL const float* acc_out = LS_G(const float, cx.accum) + a.row_out * LSIM_STATS_SIZE;
    LS_GLOBAL const float* acc_in = LS_G(const float, cx.accum) + a.row_in * LSIM_STATS_SIZE;
    const float v_nreset = acc_out[LSIM_STATS_RESET_COUNT];
    const float v_track = ls_from_fix(LS_G(const long long, ls_fix_row(cx, a.row_out))[LSIM_STATS_FIX_TRACK]);
    float r[8];
    for (int k = 0; k < 8; ++k) r[k] = acc_in[LSIM_STATS_CMD_RANGES + k];
    const int v_reset = a.reset_all == 2 ? (LS_G(const uint8_t, a.reset_mask)[env] != 0) : LSB(cx, LSIM_BUF_RESET, uint8_t)[env];
    const int v_tout = LSB(cx, LSIM_BUF_TIME_OUT, uint8_t)[env];
    const int v_eplen = (int)LSB(cx, LSIM_BUF_EPISODE_LENGTH, int64_t)[env];
    const int v_level = (int)LSB(cx, LSIM_BUF_TERRAIN_LEVELS, int64_t)[env], v_type = (int)LSB(cx, LSIM_BUF_TERRAIN_TYPES, int64_t)[env];
    const float v_org = LSB(cx, LSIM_BUF_ENV_ORIGINS, float)[3 * env + l3];
    // ---- LDS writes
    if (lane < 13) sh.root[lane] = v_root;
    if (lane < 24) sh.dofs[lane] = v_dof;
    if (lane < 12) { sh.act[lane] = v_act; sh.pre_lla[lane] = v_la; sh.pre_ltau[lane] = v_tq; }
    if (lane < 3) {
        sh.blv[lane] = v_blv;
        sh.bav[lane] = v_bav;
        sh.grav[lane] = v_grav;
        sh.pre_org[lane] = v_org;
        const bool disturbed = !a.reset_all && c.disturbance && (a.step_counter % c.disturbance_interval == 0);
        sh.disturbance[lane] = disturbed ? v_pend : 0.0f;
    }
    if (lane < 4) sh.cmd[lane] = v_cmd;
    for (int it = 0; it < (LS_NHP + 63) / 64; ++it) {
        const int k = lane + 64 * it;
        if (k < LS_NHP) sh.heights[k] = v_h[it];
    }
    {
        float* scratch = &sh.u.I6[0][0];
        for (int it = 0; it < (LSIM_NUM_OBS - LSIM_ONE_STEP_OBS + 63) / 64; ++it) {
            const int k = lane + 64 * it;
            if (k < LSIM_NUM_OBS - LSIM_ONE_STEP_OBS) scratch[k] = v_o[it];
        }
    }
    if (lane == 32) {
        sh.reset = v_reset;
        sh.pre_lc = (unsigned int)v_tout;
        sh.eplen = v_eplen;
        sh.pre_level = v_level; sh.pre_type = v_type;
        float nreset = a.reset_all == 1 ? (float)c.num_envs : v_nreset;     // reset_all == 2: counted by lsim_k_track_sum
        const bool no_reset = (a.flags & LSIM_STEP_NO_RESET) != 0;
        sh.do_reset = a.reset_all == 1 || (v_reset && !no_reset);
        sh.any_reset = a.reset_all == 1 || (!no_reset && nreset > 0.5f);
        sh.flags64[1] = (unsigned int)nreset;          // number of envs that reset this step (the ticket of the last one is this minus 1)
        // command curriculum (LR:307-308, LR:868-880): every wave derives the same new ranges from the reduced sums
        if (sh.any_reset && c.commands_curriculum && (a.step_counter % c.max_episode_length == 0)) {
            float mean = v_track / nreset;
            if (mean / (float)c.max_episode_length > 0.8f * c.reward_scales[LSIM_R_TRACKING_LIN_VEL]) {
                r[0] = fmaxf(fminf(r[0] - 0.1f, 0.0f), -c.max_backward_curriculum);
                r[1] = fmaxf(fminf(r[1] + 0.1f, c.max_forward_curriculum), 0.0f);
                r[2] = fmaxf(fminf(r[2] - 0.1f, 0.0f), -c.max_lat_curriculum);
                r[3] = fmaxf(fminf(r[3] + 0.1f, c.max_lat_curriculum), 0.0f);
            }
        }
        for (int k = 0; k < 8; ++k) sh.ranges[k] = r[k];
    }
}

// env 0 publishes the ranges of this step and clears the accumulator row the NEXT step will use
LS_FN void ph_b_housekeeping(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a) {
    if (env != 0) return;
    LS_GLOBAL float* out = LS_G(float, cx.accum) + a.row_out * LSIM_STATS_SIZE;
    if (lane < 8) out[LSIM_STATS_CMD_RANGES + lane] = sh.ranges[lane];
    if (a.reset_all == 1 && lane == 8) out[LSIM_STATS_RESET_COUNT] = (float)cx.cfg.num_envs;
    LS_GLOBAL float* nxt = LS_G(float, cx.accum) + a.row_in * LSIM_STATS_SIZE;   // the next call accumulates into the row this call read
    if (lane == 9) { nxt[LSIM_STATS_RESET_COUNT] = 0.0f; nxt[LSIM_STATS_NONFINITE] = 0.0f; }
    LS_GLOBAL long long* fnxt = LS_G(long long, ls_fix_row(cx, a.row_in));
    LS_STRIDED(k, lane, LSIM_STATS_FIX_WORDS) fnxt[k] = 0;
    LS_STRIDED(k, lane, LSIM_NUM_REWARD_TERMS) nxt[LSIM_STATS_EPISODE_SUMS + k] = 0.0f;
}

// LeggedRobot._update_terrain_curriculum (LR:846-866), lane 0 of a resetting env
LS_FN void ph_b_terrain_curriculum(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a) {
    const lsim_config& c = cx.cfg;
    if (!LS_UNIFORM(sh.do_reset) || !c.terrain_curriculum || c.mesh_type == 0 || !a.init_done) return;      // wave-uniform: a scalar branch
    // Level, type and origin row come from the load phase (sh.pre_*); the new origin is one row of a table no kernel writes, fetched through
    // the scalar cache (ls_uniform_load: not queued behind the vector stores in flight) and handed to ph_b_reset in sh.pre_org.  Every lane
    // evaluates the same arithmetic on the same LDS values (a scalar load may only stand in wave-uniform control flow); lane 0 stores.
    float dx = sh.root[0] - sh.pre_org[0], dy = sh.root[1] - sh.pre_org[1];
    float dist = sqrtf(dx * dx + dy * dy);
    int up = dist > c.terrain_length / 2.0f;
    int down = (dist < sqrtf(sh.cmd[0] * sh.cmd[0] + sh.cmd[1] * sh.cmd[1]) * c.episode_length_s * 0.5f) && !up;
    int64_t lvl = (int64_t)sh.pre_level + (int64_t)up - (int64_t)down;
    if (lvl >= c.terrain_num_rows) lvl = (int64_t)(ls_draw(cx, env, (uint32_t)a.step_counter ^ a.rng_salt, LSIM_RNG_RESET_LEVEL, 0) * (float)c.terrain_num_rows);
    else if (lvl < 0) lvl = 0;
    LS_GLOBAL const float* to = LSB(cx, LSIM_BUF_TERRAIN_ORIGINS, float) + (lvl * c.terrain_num_cols + (int64_t)sh.pre_type) * 3;
    float o[3];
    ls_uniform_load3(to, o);
    LS_LDS_FENCE();          // every lane has read the old origin before lane 0 replaces it
    if (lane == 0) {
        LSB(cx, LSIM_BUF_TERRAIN_LEVELS, int64_t)[env] = lvl;
        for (int k = 0; k < 3; ++k) { LSB(cx, LSIM_BUF_ENV_ORIGINS, float)[3 * env + k] = o[k]; sh.pre_org[k] = o[k]; }
    }
}

// reset_idx body for a resetting env (LR:316-361), in two phases: the new state into LDS (ph_b_reset_state), everything that goes to global
// memory afterwards (ph_b_reset_store).  Kernel A's fused tail samples the terrain under the new pose BETWEEN the two: those loads then
// queue behind none of reset_idx's stores and atomics (vmcnt retires in order; a resetting wave is the slowest kind of wave of a launch, and a
// launch of one round of waves lasts as long as its slowest wave).  Lane roles in the comments.
// Every uniform reset_idx consumes for this env, one Philox block per lane and all of them at once: lanes 0-5 the six blocks of the joint
// stream (24 draws), 6-8 the three of the root state, 9 the commands', 10-11 the domain randomisation's -- the values and keys of the
// draws the roles used to make one after the other inside their divergent branches (nine blocks in a row on a resetting wave, which is
// the slowest kind of wave of a launch).  `dr`: 48 floats of dead LDS -- kernel A: the reward-part array (dead since ph_reward_terms; its
// observation history sits in Mbl .. Sinv); kernel B: Mbl .. (it has no dynamics, and ITS history is parked where the reward parts are).
LS_FN float* ls_reset_draws_a(WaveShared& sh) { return &sh.u.r.rj[0][0]; }
LS_FN float* ls_reset_draws_b(WaveShared& sh) { return &sh.Mbl[0][0]; }
static_assert(sizeof(((WaveShared*)0)->u.r.rj) >= 48 * sizeof(float) && sizeof(((WaveShared*)0)->Mbl) >= 48 * sizeof(float), "reset draws parked in dead arrays");
LS_FN void ph_b_reset_draws(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a, float* dr) {
    if (!LS_UNIFORM(sh.do_reset)) return;
    if (lane >= 12) return;
    const lsim_config& c = cx.cfg;
    const uint32_t stepw = (uint32_t)a.step_counter ^ a.rng_salt;
    const uint32_t tag = lane < 6 ? LSIM_RNG_RESET_DOF : (lane < 9 ? LSIM_RNG_RESET_ROOT : (lane < 10 ? LSIM_RNG_RESET_CMD : LSIM_RNG_RESET_DR));
    const uint32_t block = (uint32_t)(lane < 6 ? lane : (lane < 9 ? lane - 6 : (lane < 10 ? 0 : lane - 10)));
    float u[4];
    ls_u01x4(c.seed, c.rank, (uint32_t)env, stepw, tag, block, u);
    for (int k = 0; k < 4; ++k) dr[4 * lane + k] = u[k];
}
LS_FN void ph_b_reset_state(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a, const float* q0 /* default_dof_pos: LDS copy or the config's */,
                            const float* dr /* ph_b_reset_draws */) {
    if (!LS_UNIFORM(sh.do_reset)) return;          // wave-uniform: a scalar branch, nothing of the body is issued for the other waves
    const lsim_config& c = cx.cfg;
    // dr: [0..23] joints (stream index = position), [24..35] root, [36..39] commands, [40..47] domain randomisation
    if (lane < 12) {  // _reset_dofs (LR:690-716)
        float pos = q0[lane];
        if (c.has_dof_init_pos_ratio)
            pos = pos * rand_range(dr[lane], c.dof_init_pos_ratio_range[0], c.dof_init_pos_ratio_range[1]);
        float vel = 0.0f;
        if (c.randomize_dof_vel) {
            float lo = c.dof_init_vel_range[0], hi = c.dof_init_vel_range[1];
            vel = dr[12 + lane] * fabsf(hi - lo) + fminf(lo, hi);
        }
        sh.dofs[2 * lane] = pos; sh.dofs[2 * lane + 1] = vel;
    } else if (lane == 12) {  // _reset_root_states (LR:718-820)
        const float* org = sh.pre_org;          // this env's origin row, as ph_b_terrain_curriculum left it
        float u[12];
        for (int k = 0; k < 12; ++k) u[k] = dr[24 + k];
        float r[13];
        for (int k = 0; k < 13; ++k) r[k] = c.base_init_state[k];
        for (int k = 0; k < 3; ++k) r[k] += org[k];
        if (c.mesh_type != 0) {
            if (c.has_base_init_pos_range) for (int k = 0; k < 3; ++k) r[k] += rand_range(u[k], c.base_init_pos_range[k][0], c.base_init_pos_range[k][1]);
            else for (int k = 0; k < 2; ++k) r[k] += rand_range(u[k], -1.0f, 1.0f);
        }
        if (c.has_base_init_rot_range) {
            float rpy[3];
            for (int k = 0; k < 3; ++k) rpy[k] = rand_range(u[3 + k], c.base_init_rot_range[k][0], c.base_init_rot_range[k][1]);
            quat_from_euler_xyz(rpy[0], rpy[1], rpy[2], r + 3);
        }
        for (int k = 0; k < 6; ++k) r[7 + k] = rand_range(u[6 + k], c.base_init_vel_range[k][0], c.base_init_vel_range[k][1]);
        for (int k = 0; k < 13; ++k) sh.root[k] = r[k];
    } else if (lane == 13) {  // _resample_commands (LR:320)
        float cm[4] = {sh.cmd[0], sh.cmd[1], sh.cmd[2], sh.cmd[3]};
        ls_resample_commands_u(cx, env, dr + 36, sh.ranges, cm);
        // sh.cmd is refreshed by the episode-statistics phase (lane 13 owns it here, other lanes may still read the old value)
        sh.rewv[0] = cm[0]; sh.rewv[1] = cm[1]; sh.rewv[2] = cm[2]; sh.rewv[3] = cm[3];
    }
}
LS_FN void ph_b_reset_store(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a, const float* dr) {
    if (!LS_UNIFORM(sh.do_reset)) return;          // wave-uniform: a scalar branch, nothing of the body is issued for the other waves
    const lsim_config& c = cx.cfg;
    if (lane < 12) {
        LS_GLOBAL float* dof = LSB(cx, LSIM_BUF_DOF_STATE, float) + 24 * env;
        dof[2 * lane] = sh.dofs[2 * lane]; dof[2 * lane + 1] = sh.dofs[2 * lane + 1];
        LSB(cx, LSIM_BUF_LAST_ACTIONS, float)[12 * env + lane] = 0.0f;       // LR:323-327
        LSB(cx, LSIM_BUF_LAST_LAST_ACTIONS, float)[12 * env + lane] = 0.0f;
        LSB(cx, LSIM_BUF_LAST_DOF_POS, float)[12 * env + lane] = 0.0f;
        LSB(cx, LSIM_BUF_LAST_DOF_VEL, float)[12 * env + lane] = 0.0f;
        LSB(cx, LSIM_BUF_LAST_TORQUES, float)[12 * env + lane] = 0.0f;
    } else if (lane == 12) {
        for (int k = 0; k < 4; ++k) LSB(cx, LSIM_BUF_COMMANDS, float)[4 * env + k] = sh.rewv[k];
    } else if (lane == 14) {  // domain-randomisation redraw (LR:336-343, LR:533-537)
        const float* u = dr + 40;
        if (c.randomize_kp) LSB(cx, LSIM_BUF_KP_FACTORS, float)[env] = rand_range(u[0], c.kp_range[0], c.kp_range[1]);
        if (c.randomize_kd) LSB(cx, LSIM_BUF_KD_FACTORS, float)[env] = rand_range(u[1], c.kd_range[0], c.kd_range[1]);
        if (c.randomize_motor_strength) LSB(cx, LSIM_BUF_MOTOR_STRENGTH_FACTORS, float)[env] = rand_range(u[2], c.motor_strength_range[0], c.motor_strength_range[1]);
        if (c.randomize_friction) LSB(cx, LSIM_BUF_FRICTION, float)[env] = rand_range(u[3], c.friction_range[0], c.friction_range[1]);
        if (c.randomize_restitution) LSB(cx, LSIM_BUF_RESTITUTION, float)[env] = rand_range(u[4], c.restitution_range[0], c.restitution_range[1]);
    } else if (lane == 15) {
        for (int f = 0; f < 4; ++f) LSB(cx, LSIM_BUF_FEET_AIR_TIME, float)[4 * env + f] = 0.0f;   // LR:328
        LSB(cx, LSIM_BUF_RESET, uint8_t)[env] = 1;                                                 // LR:329
    } else if (lane >= 16 && lane < 29) {
        LSB(cx, LSIM_BUF_ROOT_STATES, float)[13 * env + lane - 16] = sh.root[lane - 16];
    }
}
// extras["episode"] sums (LR:346-350); lane = reward term.  Three steps so that the result is the same whatever order the waves run in:
//   1. every resetting env adds its terms to the fixed-point accumulators
//   2. and then takes a ticket; the env that draws the last ticket of the step (count known from kernel A) knows every add has landed
//   3. and converts the sums to the fp32 row the host reads (ph_b_store)
LS_FN void ph_b_episode_stats(const LsCtx& cx, WaveShared& sh, LaneRegs& rg, int lane, int env, const LsStepArgs& a) {
    rg.ticket = -1;
    if (!sh.do_reset) return;
    if (lane == 13) for (int k = 0; k < 4; ++k) sh.cmd[k] = sh.rewv[k];
    float den = (float)(sh.eplen < 1 ? 1 : sh.eplen);
    long long* fix = ls_fix_row(cx, a.row_out);
    LS_STRIDED(k, lane, LSIM_NUM_REWARD_TERMS) {
        LS_GLOBAL float* es = LSB(cx, LSIM_BUF_EPISODE_SUMS, float) + env * LSIM_NUM_REWARD_TERMS + k;
        float v = *es;
        if (v != 0.0f) LS_ATOMIC_ADD_I64(fix + k, ls_to_fix(v / den));
        *es = 0.0f;
    }
    LS_THREADFENCE();                                  // the whole wave waits for its adds: they are ordered before its ticket
    // the ticket is only looked at in the last phase (ph_b_stats_publish): the round trip of this returning atomic overlaps the reset / observation work
    if (lane == 0) rg.ticket = (int)LS_ATOMIC_FETCH_ADD_I64(fix + LSIM_STATS_FIX_TICKET, 1);
}
LS_FN void ph_b_stats_publish(const LsCtx& cx, WaveShared& sh, LaneRegs& rg, int lane, const LsStepArgs& a) {
    if (lane == 0) sh.flags64[0] = (rg.ticket >= 0 && rg.ticket + 1 == (int)sh.flags64[1]) ? 1u : 0u;
}
LS_FN void ph_b_stats_convert(const LsCtx& cx, WaveShared& sh, int lane, const LsStepArgs& a) {
    if (!sh.flags64[0]) return;
    long long* fix = ls_fix_row(cx, a.row_out);
    LS_GLOBAL float* out = LS_G(float, cx.accum) + a.row_out * LSIM_STATS_SIZE;
    LS_STRIDED(k, lane, LSIM_NUM_REWARD_TERMS) out[LSIM_STATS_EPISODE_SUMS + k] = ls_from_fix(LS_ATOMIC_READ_I64(fix + k));
}

// publish observations (LR:403-404 + clip LR:167-171), AMP features (LR:406-416), last_* roll (LR:235-241)
LS_FN void ph_b_store(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a, const float* scratch /* the 225 history values */) {
    const lsim_config& c = cx.cfg;
    const float clipv = c.clip_observations;
    LS_GLOBAL float* obs = LSB(cx, LSIM_BUF_OBS, float) + LSIM_NUM_OBS * env;
    // rows of 270 / 238 floats start on 8-byte boundaries for every env: two floats per lane and store (half the store instructions)
    static_assert(LSIM_NUM_OBS % 2 == 0 && LSIM_NUM_PRIV_OBS % 2 == 0, "float2 rows");
    LS_STRIDED(m, lane, LSIM_NUM_OBS / 2) {
        LsF2 v;
        const int k0 = 2 * m, k1 = 2 * m + 1;
        v.x = clampf(k0 < LSIM_ONE_STEP_OBS ? sh.cur[k0] : scratch[k0 - LSIM_ONE_STEP_OBS], -clipv, clipv);
        v.y = clampf(k1 < LSIM_ONE_STEP_OBS ? sh.cur[k1] : scratch[k1 - LSIM_ONE_STEP_OBS], -clipv, clipv);
        ((LS_GLOBAL LsF2*)obs)[m] = v;
    }
    LS_GLOBAL float* priv = LSB(cx, LSIM_BUF_PRIV_OBS, float) + LSIM_NUM_PRIV_OBS * env;
    LS_STRIDED(m, lane, LSIM_NUM_PRIV_OBS / 2) {
        LsF2 v;
        v.x = clampf(sh.cur[2 * m], -clipv, clipv);
        v.y = clampf(sh.cur[2 * m + 1], -clipv, clipv);
        ((LS_GLOBAL LsF2*)priv)[m] = v;
    }
    if (lane < LSIM_NUM_AMP_OBS) {
        float v;
        if (lane < 12) v = sh.dofs[2 * lane];
        else if (lane < 15) v = sh.blv[lane - 12];
        else if (lane < 18) v = sh.bav[lane - 15];
        else v = sh.dofs[2 * (lane - 18) + 1];
        LSB(cx, LSIM_BUF_AMP_OBS, float)[LSIM_NUM_AMP_OBS * env + lane] = v;
    }
    if (lane >= 48 && lane < 60) {
        int j = lane - 48;
        LSB(cx, LSIM_BUF_LAST_LAST_ACTIONS, float)[12 * env + j] = sh.do_reset ? 0.0f : sh.pre_lla[j];   // last_actions as of LR:236: fetched by ph_load_b, zeroed by reset_idx (LR:323)
        LSB(cx, LSIM_BUF_LAST_ACTIONS, float)[12 * env + j] = sh.act[j];
        LSB(cx, LSIM_BUF_LAST_DOF_POS, float)[12 * env + j] = sh.dofs[2 * j];
        LSB(cx, LSIM_BUF_LAST_DOF_VEL, float)[12 * env + j] = sh.dofs[2 * j + 1];
        LSB(cx, LSIM_BUF_LAST_TORQUES, float)[12 * env + j] = sh.pre_ltau[j];
    }
    if (lane >= 42 && lane < 48) LSB(cx, LSIM_BUF_LAST_ROOT_VEL, float)[6 * env + lane - 42] = sh.root[7 + lane - 42];
    if (lane == 40 && sh.do_reset) LSB(cx, LSIM_BUF_EPISODE_LENGTH, int64_t)[env] = 0;                       // LR:361
    if (lane == 41 && sh.any_reset && c.send_timeouts)                                                        // LR:358-359
        LSB(cx, LSIM_BUF_EXTRAS_TIME_OUTS, uint8_t)[env] = (uint8_t)sh.pre_lc;
}
// tail of a bare reset_idx (BT:113 on all envs, LR:290 on a subset): no observation / last_* roll
LS_FN void ph_b_store_reset_all(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a) {
    if (lane == 40 && sh.do_reset) LSB(cx, LSIM_BUF_EPISODE_LENGTH, int64_t)[env] = 0;                                             // LR:361
    if (lane == 41 && sh.any_reset && cx.cfg.send_timeouts) LSB(cx, LSIM_BUF_EXTRAS_TIME_OUTS, uint8_t)[env] = (uint8_t)sh.pre_lc;  // LR:358-359
}

// =============================================================================================== kernel A
// ---- the fused tail of kernel A (LsStepArgs::fuse_tail): kernel B's per-env phases run by the same wave, on what it already holds in LDS.
// Kernel B exists because the step has one global dependency -- whether ANY env reset (the stale extras["time_outs"], LR:358) and the
// command-curriculum mean over the reset set (LR:307-308, LR:875) -- but the second only matters on the steps where
// common_step_counter % max_episode_length == 0 (one in a thousand), which the host knows before the launch, and the first only decides
// whether an N-byte mask is copied.  On every other step the per-env work needs nothing from other envs: this wave does it here, and
// lsim_k_step_finish (a few blocks) copies the mask if the reset count says so, converts the fixed-point episode sums and swaps the rows.
// The 225 history values of the observation (LR:403) are the only input the wave does not hold: they are fetched BEFORE the first state store
// of the kernel (a load behind stores waits for all of them) and parked in the dynamics arrays, which are dead by then.
LS_FN float* ls_obs_hist(WaveShared& sh) { return &sh.Mbl[0][0]; }
static_assert(offsetof(WaveShared, Sinv) + sizeof(((WaveShared*)0)->Sinv) - offsetof(WaveShared, Mbl) >= (LSIM_NUM_OBS - LSIM_ONE_STEP_OBS) * sizeof(float),
              "observation history parked in Mbl .. Sinv");
static_assert((LSIM_NUM_OBS - LSIM_ONE_STEP_OBS + 63) / 64 <= 4, "LaneRegs::hist");
// the reward part items (host-built table, ls_api_impl.h) live behind the history: only the post-physics stack reads them
LS_FN uint16_t* ls_part_items(WaveShared& sh) { return (uint16_t*)(ls_obs_hist(sh) + (LSIM_NUM_OBS - LSIM_ONE_STEP_OBS + 1)); }
static_assert(offsetof(WaveShared, nc) - offsetof(WaveShared, Mbl) >= (LSIM_NUM_OBS - LSIM_ONE_STEP_OBS + 1) * sizeof(float) + LS_MAX_PART_ITEMS * sizeof(uint16_t),
              "reward part items parked behind the observation history (Mbl .. vnew)");
// everything the post-physics stack needs from global memory that is not per-step state: issued in the last phase before the kernel's
// first store, written to LDS (dynamics arrays, dead by then) one phase later
LS_FN void ph_late_load(const LsCtx& cx, LaneRegs& rg, int lane, int env, bool fuse) {
    if (fuse)
        for (int it = 0; it < (LSIM_NUM_OBS - LSIM_ONE_STEP_OBS + 63) / 64; ++it) {
            const int k = lane + 64 * it;
            rg.hist[it] = LSB(cx, LSIM_BUF_OBS, float)[LSIM_NUM_OBS * env + (k < LSIM_NUM_OBS - LSIM_ONE_STEP_OBS ? k : 0)];
        }
    for (int it = 0; it < LS_MAX_PART_ITEMS / 64; ++it) rg.items[it] = cx.part_items[lane + 64 * it];
    const int ai = lane < LSIM_NUM_REWARD_TERMS ? lane : 0;
    rg.term_id = cx.active_terms[ai];
    rg.term_scale = cx.active_scales[ai];          // cfg.reward_scales[active_terms[ai]], tabulated by the host: no dependent load
}
LS_FN void ph_late_stage(WaveShared& sh, const LaneRegs& rg, int lane, bool fuse) {
    if (fuse) {
        float* hist = ls_obs_hist(sh);
        for (int it = 0; it < (LSIM_NUM_OBS - LSIM_ONE_STEP_OBS + 63) / 64; ++it) {
            const int k = lane + 64 * it;
            if (k < LSIM_NUM_OBS - LSIM_ONE_STEP_OBS) hist[k] = rg.hist[it];
        }
    }
    uint16_t* items = ls_part_items(sh);
    for (int it = 0; it < LS_MAX_PART_ITEMS / 64; ++it) items[lane + 64 * it] = (uint16_t)rg.items[it];
}
// what ph_load_b derives for kernel B, from kernel A's own LDS state
LS_FN void ph_tail_setup(const LsCtx& cx, WaveShared& sh, int lane, const LsStepArgs& a) {
    if (lane == 32) {
        sh.do_reset = sh.reset && !(a.flags & LSIM_STEP_NO_RESET);
        sh.any_reset = 0;                               // extras["time_outs"] is lsim_k_step_finish's
        sh.pre_lc = (unsigned int)sh.timeout;
    }
    if (lane < 12) { sh.pre_lla[lane] = sh.last_act[lane]; sh.pre_ltau[lane] = sh.tau[lane]; }   // what ph_b_store rolls into last_last_actions / last_torques
}
// the episode sums of a resetting env into the fixed-point accumulators (LR:346-350), from LDS; no ticket: lsim_k_step_finish converts
LS_FN void ph_tail_episode_stats(const LsCtx& cx, WaveShared& sh, int lane, int env, const LsStepArgs& a) {
    if (!LS_UNIFORM(sh.do_reset)) return;
    if (lane == 13) for (int k = 0; k < 4; ++k) sh.cmd[k] = sh.rewv[k];
    if (lane == 63) ls_count_reset(cx, sh, a);           // (ph_term_outputs left them to this phase)
    const float den = (float)(sh.eplen < 1 ? 1 : sh.eplen);
    long long* fix = ls_fix_row(cx, a.row_out);
    LS_STRIDED(k, lane, LSIM_NUM_REWARD_TERMS) {
        const float v = sh.pre_es[k];
        if (v != 0.0f) LS_ATOMIC_ADD_I64(fix + k, ls_to_fix(v / den));
        LSB(cx, LSIM_BUF_EPISODE_SUMS, float)[env * LSIM_NUM_REWARD_TERMS + k] = 0.0f;
    }
}

// SOLVER: LSIM_SOLVER_PGS / LSIM_SOLVER_TGS (lsim_config.solver_type), a template parameter so that each kernel carries one solver's code
template <int SOLVER> LS_WAVE_FN void ls_wave_step_a(const LsCtx& cx, const LsStepArgs& a, const int env, WaveShared& sh, LS_LANES_PARAM) {
    const lsim_config& c = cx.cfg;
    const float dt = c.sim_dt;
    constexpr bool TGS = SOLVER == LSIM_SOLVER_TGS;
    const bool skip = (a.flags & LSIM_STEP_SKIP_PHYSICS) != 0;
    [[maybe_unused]] constexpr int ls_line0 = __LINE__;   // phase-site ids (LS_PHASE_TIMING builds) count lines from here
    LS_TICK_INIT();
#if defined(LS_WAVE_TIMES)    // diagnostics build only (tools/wave_times.py): when each wave of the latest step started and ended
    const unsigned long long ls_wt0 = wall_clock64(), ls_wc0 = clock64();
    unsigned int ls_cp[16] = {};
    [[maybe_unused]] int ls_sub = -1, ls_k = 0;
#endif
    LS_PHASE(ph_load_a(cx, sh, rg, lane, env, a));
    LS_CP(0);
    for (int sub = 0; sub < c.decimation; ++sub) {
#if defined(LS_WAVE_TIMES) && LS_WAVE_TIMES == 2
        ls_sub = sub;
        if (sub == 1) { ls_cp[0] = (unsigned int)(clock64() - ls_wc0); ls_k = 1; }
#endif
        if (skip) { LS_PHASE(ph_torques(cx, sh, lane, env, sub, a.flags)); continue; }
        // phases that do not depend on each other share a barrier: (torques, kinematics), (free velocity, narrow phase),
        // (contact compaction, joint-limit rows), (apply impulses, contact forces)
        LS_TORQUES_KINEMATICS();
        LS_PHASE(ph_body_inertia(cx, sh, lane, sub == 0));
        LS_PHASE(ph_leg_composite(sh, lane));
        LS_PHASE(ph_leg_block(sh, lane));
        LS_PHASE(ph_leg_schur(sh, lane));
        LS_PHASE(ph_base_assemble(sh, lane));
        LS_PHASE(ph_base_factor(sh, lane));
        LS_PHASE(ph_free_leg(sh, lane));
        LS_PHASE(ph_free_base(sh, lane));
        bool walls = false;       // a point of this robot is in the wall path of the narrow phase (stair risers)
        LS_PHASE(ph_free_finish(sh, lane, dt); ph_collide_prefetch(cx, rg, lane); walls |= ph_collide(cx, sh, rg, lane));
        LS_COLLECTIVE(wc_compact_contacts(sh, rg, lane); wc_limits(cx, sh, lane, dt), wc_compact_contacts(sh, L); LS_PHASE(ph_limits(cx, sh, lane, dt)));
        // A launch of <= 4096 robots is ONE round of waves, over when its slowest wave is: the median wave needs 83 us, one with 8 contacts 97
        // (2.3 us per contact: rows, Delassus entries, relaxations), the kernel 117 (tools/wave_times.py).  The more contacts a robot has in
        // this sub-step, the higher its wave's issue priority over the three it shares a SIMD with, which have the slack: kernel A 0.1176 ->
        // 0.1104 ms on the flat task, 0.1405 -> 0.1298 on stairs (LSIM_STEP_FLAT_PRIORITY switches it off).  Results do not change.
        if (!(a.flags & LSIM_STEP_FLAT_PRIORITY)) {
            const int ncu = LS_UNIFORM(sh.nc) + (walls ? 2 : 0);      // a point in the wall path of the narrow phase counts like two contacts (stairs: -1.2 %)
            if (ncu >= 5) LS_SETPRIO(3); else if (ncu >= 3) LS_SETPRIO(2); else if (ncu >= 1) LS_SETPRIO(1); else LS_SETPRIO(0);
        }
        LS_PHASE(ph_rows<TGS>(cx, sh, rg, lane, dt));
#if defined(LS_EMU)
        LS_PHASE(ph_delassus(sh, rg, lane));
        if (TGS) { wc_tgs(cx, sh, L, c.num_position_iterations, dt); LS_PHASE(ph_contact_forces(sh, lane, dt)); }
        else { wc_pgs(sh, L, c.solver_iterations); LS_PHASE(ph_apply_impulses(sh, lane); ph_contact_forces(sh, lane, dt)); }
#else
        // rows, sweep(s), constrained velocity, contact forces
        if constexpr (TGS) LS_PHASE(wc_delassus_tgs(cx, sh, rg, lane, c.num_position_iterations, dt));
        else LS_PHASE(wc_delassus_pgs(sh, rg, lane, c.solver_iterations, dt));
#endif
        if (TGS) LS_PHASE(ph_integrate_tgs(cx, sh, lane, dt, c.num_position_iterations));
        else LS_PHASE(ph_integrate(cx, sh, lane, dt));
        if (sub < 4) LS_CP(1 + sub);
#if defined(LS_WAVE_TIMES)
        ls_sub = -1;
#endif
#if defined(LS_EXP_TWICE) && LS_EXP_TWICE == 9001      // cost probe: the integrator again with a zero step (leaves the state where it is)
        LS_PHASE(ph_integrate(cx, sh, lane, 0.0f));
#endif
    }
    const bool fuse = a.fuse_tail != 0;
    if (!skip) {
#if defined(LS_EMU)
        LS_PHASE(ph_kinematics(sh, lane); ph_late_load(cx, rg, lane, env, fuse); ph_heights_issue(cx, sh, rg, lane));
#else
        LS_COLLECTIVE(wc_kinematics(sh, lane); ph_late_load(cx, rg, lane, env, fuse); ph_heights_issue(cx, sh, rg, lane), (void)0);
#endif
        LS_CP(5);
        LS_PHASE(ph_body_states_all(cx, sh, rg, lane));
        LS_PHASE(ph_heights_finish(cx, sh, rg, lane, env); ph_late_stage(sh, rg, lane, fuse));   // the loads' results, before the first store; Mbl .. vnew are dead from here on
        LS_CP(6);
        LS_COLLECTIVE(ph_store_body_states(cx, rg, lane, env); ph_store_sim_state(cx, sh, lane, env); wc_count_nonfinite(cx, sh, lane, a),
                      LS_PHASE(ph_store_body_states(cx, rg, lane, env); ph_store_sim_state(cx, sh, lane, env)); emu_count_nonfinite(cx, sh, a));
        LS_CP(7);
    } else {
        LS_PHASE(ph_late_load(cx, rg, lane, env, fuse); ph_heights_issue(cx, sh, rg, lane));
        LS_PHASE(ph_heights_finish(cx, sh, rg, lane, env); ph_late_stage(sh, rg, lane, fuse));
        LS_PHASE(ph_load_injected(cx, sh, lane, env));
    }
    // ---- post_physics_step (LR:178-228)
    LS_PHASE(ph_post_state(cx, sh, lane, env));
    LS_PHASE(ph_callback(cx, sh, lane, env, a, sh.ranges));
    LS_CP(8);
#if defined(LS_EXP_TWICE) && LS_EXP_TWICE == 9003      // cost probe: counter-based draws, so a second pass writes the same values
    LS_PHASE(ph_callback(cx, sh, lane, env, a, sh.ranges));
#endif
    LS_PHASE(ph_termination(cx, sh, lane, env); ph_reward_parts(cx, sh, ls_part_items(sh), lane, env));
    LS_PHASE(ph_reward_terms(cx, sh, rg, lane, env));
#if defined(LS_EXP_TWICE) && LS_EXP_TWICE == 9002      // cost probe (adds the step's rewards to the episode sums twice: statistics only)
    LS_PHASE(ph_reward_terms(cx, sh, rg, lane, env));
#endif
    LS_PHASE(ph_reward_total(cx, sh, lane, env));
    LS_CP(9);
    LS_PHASE(if (LS_UNIFORM(sh.reset)) ph_build_obs(cx, sh, lane, env, (uint32_t)a.step_counter, LSIM_RNG_TERM_NOISE, sh.cur, sh.jc_q0));
    LS_PHASE(ph_term_outputs(cx, sh, lane, env, a));
    LS_CP(13);
    if (fuse) {     // LR:229-241 + LR:167-171 for this robot (kernel B's phases; its cross-env part is lsim_k_step_finish)
        LS_PHASE(ph_tail_setup(cx, sh, lane, a));
        LS_PHASE(ph_b_terrain_curriculum(cx, sh, lane, env, a); ph_b_reset_draws(cx, sh, lane, env, a, ls_reset_draws_a(sh)));
        LS_PHASE(ph_b_reset_state(cx, sh, lane, env, a, sh.jc_q0, ls_reset_draws_a(sh)));
        LS_CP(14);
        LS_PHASE(if (LS_UNIFORM(sh.do_reset) && c.measure_heights) ph_heights(cx, sh, lane, env, true, sh.mpx, sh.mpy));     // loads: ahead of reset_idx's stores and atomics
        LS_CP(15);
        LS_PHASE(ph_b_reset_store(cx, sh, lane, env, a, ls_reset_draws_a(sh)); ph_tail_episode_stats(cx, sh, lane, env, a));
        LS_CP(10);
        LS_PHASE(ph_build_obs(cx, sh, lane, env, (uint32_t)a.step_counter, LSIM_RNG_OBS_NOISE, sh.cur, sh.jc_q0));
        LS_CP(11);
        LS_PHASE(ph_b_store(cx, sh, lane, env, a, ls_obs_hist(sh)));
        LS_CP(12);
    }
    LS_TICK_FLUSH();
#if defined(LS_WAVE_TIMES)
    if (lane0 == 0 && env < 65536) {
        g_ls_wave_times[4 * env + 0] = ls_wt0; g_ls_wave_times[4 * env + 1] = wall_clock64();
        for (int i = 0; i < 16; ++i) g_ls_wave_cp[16 * env + i] = ls_cp[i];
        g_ls_wave_times[4 * env + 2] = clock64() - ls_wc0; g_ls_wave_times[4 * env + 3] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 16);
    }
#endif
}

LS_WAVE_FN void ls_wave_step_b(const LsCtx& cx, const LsStepArgs& a, const int env, WaveShared& sh, LS_LANES_PARAM) {
    const lsim_config& c = cx.cfg;
    [[maybe_unused]] constexpr int ls_line0 = __LINE__ - 96;   // kernel B's sites land above kernel A's (A uses 0..95)
#if defined(LS_WAVE_TIMES)
    [[maybe_unused]] int ls_sub = -1, ls_k = 0; [[maybe_unused]] unsigned int ls_cp[16]; [[maybe_unused]] const unsigned long long ls_wc0 = 0;   // (LS_PHASE's checkpoint hook: kernel A only)
#endif
    LS_TICK_INIT();
    LS_PHASE(ph_load_b(cx, sh, lane, env, a));
    LS_PHASE(ph_b_housekeeping(cx, sh, lane, env, a); ph_b_terrain_curriculum(cx, sh, lane, env, a); ph_b_reset_draws(cx, sh, lane, env, a, ls_reset_draws_b(sh)));
    LS_PHASE(ph_b_reset_state(cx, sh, lane, env, a, cx.cfg.default_dof_pos, ls_reset_draws_b(sh)));
    LS_PHASE(ph_b_reset_store(cx, sh, lane, env, a, ls_reset_draws_b(sh)));
    LS_PHASE(ph_b_episode_stats(cx, sh, rg, lane, env, a));
    LS_PHASE(if (LS_UNIFORM(sh.do_reset) && c.measure_heights) ph_heights(cx, sh, lane, env, true, c.measured_points_x, c.measured_points_y));
    if (a.reset_all) {   // reset_idx only: the observation roll belongs to the step that follows (BT:114)
        LS_PHASE(ph_b_store_reset_all(cx, sh, lane, env, a); ph_b_stats_publish(cx, sh, rg, lane, a));
        LS_PHASE(ph_b_stats_convert(cx, sh, lane, a));
        return;
    }
    LS_PHASE(ph_build_obs(cx, sh, lane, env, (uint32_t)a.step_counter, LSIM_RNG_OBS_NOISE, sh.cur, cx.cfg.default_dof_pos));
    LS_PHASE(ph_b_store(cx, sh, lane, env, a, &sh.u.I6[0][0]); ph_b_stats_publish(cx, sh, rg, lane, a));
    LS_PHASE(ph_b_stats_convert(cx, sh, lane, a));
    LS_TICK_FLUSH();
}

// =============================================================================================== lsim_k_step_finish
// The cross-env leftovers of a step whose kernel A ran the fused tail (thread t of a small grid; no LDS):
//   every env:  extras["time_outs"] = time_out_buf if ANY env reset in this step (LR:358-359: only re-assigned inside reset_idx, quirk 5)
//   t < ...  :  the live command ranges carried to this step's row, the fixed-point episode sums of the reset set converted to the fp32 row
//               the host reads (LR:346-350), and the other row cleared for the next step's accumulation (kernel B's housekeeping)
LS_FN void ls_step_finish_env(const LsCtx& cx, const LsStepArgs& a, int env) {
    LS_GLOBAL const float* out = LS_G(const float, cx.accum) + a.row_out * LSIM_STATS_SIZE;
    const bool any = out[LSIM_STATS_RESET_COUNT] > 0.5f && !(a.flags & LSIM_STEP_NO_RESET);
    if (any && cx.cfg.send_timeouts) LSB(cx, LSIM_BUF_EXTRAS_TIME_OUTS, uint8_t)[env] = LSB(cx, LSIM_BUF_TIME_OUT, uint8_t)[env];
}
LS_FN void ls_step_finish_rows(const LsCtx& cx, const LsStepArgs& a, int t) {
    LS_GLOBAL float* out = LS_G(float, cx.accum) + a.row_out * LSIM_STATS_SIZE;
    LS_GLOBAL float* nxt = LS_G(float, cx.accum) + a.row_in * LSIM_STATS_SIZE;
    const bool any = out[LSIM_STATS_RESET_COUNT] > 0.5f;
    if (t < 8) out[LSIM_STATS_CMD_RANGES + t] = nxt[LSIM_STATS_CMD_RANGES + t];
    if (t < LSIM_NUM_REWARD_TERMS) {
        if (any) out[LSIM_STATS_EPISODE_SUMS + t] = ls_from_fix(LS_G(const long long, ls_fix_row(cx, a.row_out))[t]);
        nxt[LSIM_STATS_EPISODE_SUMS + t] = 0.0f;
    }
    if (t < LSIM_STATS_FIX_WORDS) LS_G(long long, ls_fix_row(cx, a.row_in))[t] = 0;
    if (t == 0) { nxt[LSIM_STATS_RESET_COUNT] = 0.0f; nxt[LSIM_STATS_NONFINITE] = 0.0f; }
}
