// ls_shared.h -- per-robot on-chip state of the leggedsim kernels.
//
// Execution model (DESIGN.md "Kernel structure"): ONE WAVEFRONT PER ROBOT (blockDim = 64).
// A kernel is a sequence of *phases*; inside a phase every lane does independent work on its own
// item (a leg, a body, a collision point, a constraint row, an observation slot ...) and lanes exchange
// data only through `WaveShared` (LDS) at phase boundaries.  The few true wave collectives (ballot
// compaction, the Gauss-Seidel sweep) are written against wave intrinsics in ls_physics.h.
// Because phases are plain functions of (shared, lane-private state, lane index), the same source is
// also compiled by g++ for tests/emu, which runs the 64 lanes of a phase in a loop -- a development aid to
// check the lane orchestration on CPU; it is not a product path.
#pragma once
#include "../../include/lsim.h"
#include "ls_math.h"

#define LS_NB LSIM_NUM_BODIES
// row stride of the spatial-inertia array (lane = body).  36 floats put bodies b, b + 8 and b + 16 on the same LDS banks (36 b mod 32 = 4 b); 38 keeps the
// 8-byte alignment and leaves only bodies 0 and 16 on one bank.  Free: the array lives in a union with the larger constraint rows Y.  Round 5 measured it
// (0.1101 -> 0.1097 ms) and got NaNs in the GPU suite: the two padding floats of a row were never written, and they lie inside Y rows that the solver reads as
// "holds no row yet" slots (times a zero impulse) -- uninitialised LDS is not finite.  Round 6: ph_body_inertia zeroes them with the row (two LDS stores per
// body lane and sub-step); interleaved A/B on one lease: flat 0.1103 -> 0.1097 ms, stairs 0.1209 -> 0.1197 (tools/gpu_ab_kernel_a.sh), GPU suite green.
#ifndef LS_I6_STRIDE
#define LS_I6_STRIDE 38
#endif
#define LS_NV 18
#define LS_MAXC LSIM_MAX_CONTACTS
#define LS_MAXR (3 * LSIM_MAX_CONTACTS + LSIM_NUM_DOF)  // 60 <= 64 lanes

#define LS_MAX_PART_ITEMS 192          // three passes of 64 lanes
#define LS_REW_MAX_PARTS 12
#define LS_BH_PARTS 9                // the 63 base-height samples: 9 parts of 7
// number of parts of a term; 0 = a scalar term (ls_reward_scalar)
LS_FN int ls_reward_num_parts(int id) {
    switch (id) {
        case LSIM_R_DOF_VEL: case LSIM_R_DOF_ACC: case LSIM_R_DOF_VEL_LIMITS: case LSIM_R_DOF_POS_DIF: case LSIM_R_DOF_POS_LIMITS:
        case LSIM_R_ACTION_RATE: case LSIM_R_SMOOTHNESS: case LSIM_R_TORQUES: case LSIM_R_TORQUES_DISTRIBUTION: case LSIM_R_TORQUES_DIF:
        case LSIM_R_TORQUE_LIMITS: case LSIM_R_JOINT_POWER: case LSIM_R_POWER: case LSIM_R_POWER_DISTRIBUTION:
        case LSIM_R_STAND_STILL: case LSIM_R_STAND_NICE:
            return 12;
        case LSIM_R_BASE_HEIGHT: case LSIM_R_BASE_HEIGHT_UP:
            return LS_BH_PARTS;
        case LSIM_R_HIP_ACTION_MAGNITUDE:
        case LSIM_R_HIP_POS: case LSIM_R_HIP_POS_UP: case LSIM_R_THIGH_POSE: case LSIM_R_THIGH_POSE_UP: case LSIM_R_CALF_POSE: case LSIM_R_CALF_POSE_UP:
        case LSIM_R_FEET_AIR_TIME: case LSIM_R_FEET_CONTACT_FORCES: case LSIM_R_FEET_SLIDE: case LSIM_R_FEET_SLIDE_UP:
        case LSIM_R_FOOT_CLEARANCE_BASE: case LSIM_R_FOOT_CLEARANCE_BASE_UP: case LSIM_R_FOOT_CLEARANCE_TERRAIN: case LSIM_R_FOOT_CLEARANCE_TERRAIN_UP:
            return 4;
        default: return 0;
    }
}

// constant per simulator instance; lives in device global memory, read through uniform (scalar) loads
struct LsCtx {
    lsim_config cfg;
    lsim_robot_model model;
    void* buf[LSIM_NUM_BUFFERS];
    float* accum;                     // [2][LSIM_STATS_SIZE] ping-pong per-step reductions (== buf[LSIM_BUF_STATS])
    int32_t active_terms[LSIM_NUM_REWARD_TERMS];
    float active_scales[LSIM_NUM_REWARD_TERMS];      // cfg.reward_scales[active_terms[i]]
    int32_t num_active;
    // (term, part) work items of the reward terms that are sums (ls_post.h: ph_reward_parts): (id << 10) | (active index << 4) | part
    uint16_t part_items[LS_MAX_PART_ITEMS];
    int32_t num_part_items;
    uint64_t parted_mask;             // bit i: the parts of active term i are in the item table
    float cmd_span_init[4];
};

// per-launch arguments (by value)
struct LsStepArgs {
    const float* actions;     // [N,12]
    int64_t step_counter;     // common_step_counter AFTER this step's increment (LR:194)
    uint32_t flags;           // LSIM_STEP_*
    int32_t init_done;        // LR:853
    int32_t row_in;           // accumulator row holding the previous step's command ranges
    int32_t row_out;          // accumulator row this step writes (step_counter & 1)
    int32_t reset_all;        // kernel B only.  0: the reset tail of a step; 1: a bare reset_idx(all) (BaseTask.reset, BT:113);
                              // 2: a bare reset_idx(env_ids) (LR:290) on the envs flagged in reset_mask
    const uint8_t* reset_mask; // reset_all == 2: u8 [num_envs] on the device, nonzero = reset this env
    uint32_t rng_salt;        // xor-ed into the step word of reset_idx's draws: 0 inside a step and for lsim_reset_all; a multiple of the golden ratio
                              // per lsim_reset_envs call, so that by-hand resets never repeat the adjacent step's draws or each other's
    int32_t fuse_tail;        // kernel A only.  1: this wave also runs kernel B's per-env work for its robot (reset_idx, observations, last_* roll)
                              // and the step's cross-env leftovers go to lsim_k_step_finish; 0: kernel B follows (command-curriculum steps)
};

// body model entries staged in LDS once per kernel
struct LsBodyLds {
    float mass, com[3], inertia[6], jpos[3], axis[3];
};

struct WaveShared {
    // ---- carried across the step
    float root[13];
    float q[12], qd[12];
    float act[12], last_act[12], ms[12];
    float tau[12];
    float kpf, kdf, mu, payload;
    float comd[3], pend[3];
    int delay;
    LsBodyLds body[LS_NB];
    // ---- inputs of the post-physics stack, fetched by ph_load_a before the kernel has any store in flight: a global load issued after
    //      stores waits for every one of them (vmcnt is in order), so nothing after the physics loop reads global state buffers
    union {
        struct { float pre_lla[12], pre_ldp[12], pre_ldv[12], pre_ltau[12]; };   // last_last_actions, last_dof_pos, last_dof_vel, last_torques
        float pre4[48];                                                           // the same 48 floats as one array (filled by lanes 0..47 of the load phase)
    };
    float pre_cmd[4], pre_air[4];                                // commands, feet_air_time
    int pre_eplen, pre_level;                                    // episode_length_buf, terrain_levels (low words)
    int pre_type;                                                // terrain_types (low word)
    float pre_org[3];                                            // env_origins row: reset_idx's inputs (LR:846-866, LR:718-731) fetched with everything
                                                                 // else, because a load issued once the kernel's stores are in flight waits for all of
                                                                 // them (a resetting wave spent 54 k of its 331 k ticks in two such waits, round 4)
    unsigned int pre_lc;                                         // last_contacts: 4 bytes
    float pre_es[LSIM_NUM_REWARD_TERMS];                         // episode_sums row
    unsigned char filt[4];                                       // contact_filt of this step (LR:207-209)
    // ---- per-joint constants of the model / config (ph_load_a): the torque, limit-row, reward and observation code reads them by a per-lane
    //      joint index, which from the context in global memory is a VECTOR load with its own round trip at every use (and behind the
    //      kernel's stores once the post-physics stack has begun: vmcnt retires in order) -- round 4: ph_build_obs of the fused tail cost
    //      22 k of kernel A's 256 k ticks for four such loads
    float jc_q0[12], jc_kp[12], jc_kd[12], jc_taumax[12];       // default_dof_pos, p_gains, d_gains, torque_limits
    float jc_lo[12], jc_hi[12], jc_vmax[12];                     // dof_pos_lower / upper, dof_vel_limit
    float mpx[LSIM_MAX_HEIGHT_PTS_X], mpy[LSIM_MAX_HEIGHT_PTS_Y];  // measured_points_x / y (AGC:79-80): the height samplers index them per lane
    // ---- kinematics / dynamics of the current sub-step (world axes, positions relative to the base origin), overlaid
    //      with the post-physics scratch that is only used once the last sub-step is over (keeps the block <= 10 KB so
    //      that 16 robots per CU -- all 4096 of a 256-CU launch -- are resident at once)
    union {
        struct {
            float R[LS_NB][9];
            float p[LS_NB][3];
            float S[12][6];          // motion subspace per dof
            float V[LS_NB][6];       // body twists
            float Ab[LS_NB][6];      // bias accelerations
            float Fb[LS_NB][6];      // bias forces (per body, then leg totals in legF)
        };
        struct {
            float heights[LSIM_NUM_HEIGHT_PTS];
            float bh[LSIM_NUM_BASE_HEIGHT_PTS];
            float cur[LSIM_NUM_PRIV_OBS];
            float rewv[LSIM_NUM_REWARD_TERMS];
        };
    };
    float com0[3];           // base COM (world axes)
    union {
        float I6[LS_NB][LS_I6_STRIDE];       // spatial inertias, 6 x 6 row-major in the first 36 floats of a row (dead after the composite pass)
        struct {
            // Per constraint row i, M^-1 J_i^T WITHOUT its base-coupling leg terms: [0..5] z = Sb^-1 (Jb - Mbl y), [6 + 3l + k] = y[k] = (Mll^-1 Jl)[k]
            // on the row's own leg and 0 on the others.  The full vector has - G_l z on every leg; the solver never needs it:
            //   W_ij = J_i M^-1 J_j^T = a_i . z_j + Jl_i . y_j[leg_i]   with a_i = Jb_i - Mbl y_i (registers),   v+ = vfree + sum_r Y_r lam_r - G (sum_r z_r lam_r)
            // which saves 72 FMAs and as many LDS reads per row against forming - G_l z for all four legs.
            float Y[LS_MAXR][LS_NV];
            float dirs[3 * LS_MAXC][3];      // contact rows only
        } c;
        struct { float rj[LSIM_NUM_REWARD_TERMS][12]; } r;   // post-physics: parts of the reward terms (ls_post.h)
    } u;
    float Mbl[4][18];        // 6x3: columns F_hip, F_thigh, F_calf
    float G[4][18];          // 3x6: Mll^-1 Mlb^T
    float Lll[4][9];         // Cholesky of the 3x3 leg block (l00,l10,l11,l20,l21,l22) and the reciprocals of its diagonal
    float hl[4][3];
    float legF[4][6];
    float yl[4][3];
    float Sb[36];            // Schur complement on the base
    float Sinv[36];          // its inverse (symmetric), from the Cholesky factor: every later solve is a 6x6 mat-vec
                             // (TGS: both are dead once the rows are built; the solver parks the base twist change of the impulses after
                             //  each sub-iteration in these 72 floats for the integrator -- ls_tgs_base_twist)
    float hb[6], rb[6], ab[6];
    float vfree[LS_NV], vnew[LS_NV];
    // ---- contacts
    int nc, nrows;
    short nact, nact_max;    // collision points in contact before the cap (diagnostic, LSIM_BUF_CONTACT_COUNT)
    int cbody[LS_MAXC];
    float cpos[LS_MAXC][3], cn[LS_MAXC][3], cdist[LS_MAXC];
    int limdof[12];
    float limvt[12], limrng[12];     // lower velocity bound and width of the admissible interval of each joint-limit row
    int nlim;
    float lam[LS_MAXR];
    float cf[LS_NB][3];
    unsigned int flags64[2];
    // ---- post-physics scratch
    float blv[3], bav[3], grav[3];
    float cmd[4];
    float dofs[24];          // interleaved (pos, vel) copy of the final joint state (layout of the gym dof tensor)
    float feet[4][6];        // world position / linear velocity of the feet (rows of rigid_body_states)
    float ranges[8];         // live command ranges [4][2]
    int eplen, do_reset, any_reset;
    float disturbance[3];
    int reset, timeout;
};

// lane-private state that survives phase boundaries (VGPRs on the GPU, an array element in tests/emu)
struct LaneRegs {
    // collision point owned by this lane
    int cp_body;
    int cp_active;
    float cp_dist, cp_n[3], cp_x[3];
    float cp_r, cp_pos[3];   // the point's constants, fetched from the model at the top of the sub-step (ph_collide_prefetch)
    // constraint row owned by this lane
    int row_kind;            // 0 normal, 1/2 friction, 3 joint limit, -1 none
    int row_leg;             // leg whose dofs the row touches, -1 for the base body
    float Jb[6], Jl[3];
    float brow, wdiag;       // PGS: right-hand side J vfree - target;  TGS: J vfree (the targets move with the sub-iterations)
    float row_rng;           // PGS: width of a two-sided row's interval (+inf: one-sided);  TGS: the joint's velocity limit (limit rows)
    float tg_a, tg_b;        // TGS: normal row -- the contact's gap;  limit row -- the joint's distances to its lower / upper stop
    int ticket;              // kernel B, lane 0: this env's ticket among the envs that reset in this step (-1: none)
    float hist[4];           // kernel A with the fused tail: the 225 observation-history values on their way from global memory to LDS
    uint32_t items[LS_MAX_PART_ITEMS / 64];   // kernel A: the reward part items on the same way (a register each: see hraw) (ph_late_load -> ph_late_stage)
    int hraw[2 * ((LSIM_NUM_HEIGHT_PTS + 63) / 64 + 1)];   // kernel A (one register per sample: packing two would wait for the loads): raw grid samples of this lane's height points (ph_heights_issue -> ph_heights_finish)
    float bs[13];            // kernel A, lane = body: its row of the rigid-body state tensor between ph_body_states_all and ph_store_body_states
    int term_id;             // kernel A: active reward term owned by this lane in ph_reward_terms and its scale (fetched before the first store)
    float term_scale;
#if defined(LS_EMU)
    float W[LS_MAXR];        // Delassus row (the GPU path keeps it local to wc_delassus_pgs)
#endif
};
