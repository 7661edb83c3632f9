// ls_api_impl.h -- implementation of the C-ABI declared in include/lsim.h, written against a small backend
// interface so that the HIP library (lsim_hip.hip) and the tests-only lane emulator (tests/emu/emu_lsim.cpp)
// share the host logic.  The including file must define, before including this header:
//   LS_API(name)                     symbol name (lsim_##name for the product)
//   lsbk_malloc / lsbk_free / lsbk_h2d / lsbk_memset       device-memory primitives, return 0 on success
//   lsbk_launch_a / lsbk_launch_b / lsbk_launch_reduce / lsbk_launch_finish (lsim_sim*, const LsStepArgs&, void* stream)
//   lsbk_set_device(int), lsbk_prof_mark(lsim_sim*, int which, void* stream), lsbk_prof_free(lsim_sim*)
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/lsim_layout.h"
#include "ls_shared.h"

struct lsim_sim {
    lsim_config cfg;
    lsim_robot_model model;
    LsCtx host_ctx;
    LsCtx* dev_ctx;
    char* arena;
    bool owns_arena;
    size_t offsets[LSIM_NUM_BUFFERS];
    int device_id;
    int64_t step_counter;
    uint32_t reset_calls;   // lsim_reset_envs calls so far (salt of their random draws)
    int priority_max_envs;  // contact-count wave priorities up to this many robots (LSIM_PRIORITY_MAX_ENVS overrides: a measurement hook)
    int init_done;
    int stats_row;
    void* prof;        // backend-owned profiling state (HIP events), may be null
    char err[256];
};

static const size_t LS_ALIGN = 256;

static int ls_check_cfg(const lsim_config* c) {
    if (c->abi_version != LSIM_ABI_VERSION) return LSIM_E_ABI;
    if (c->num_envs <= 0 || c->decimation <= 0 || c->decimation > 16) return LSIM_E_INVALID;
    if (c->mesh_type != 0 && (c->grid_rows < 2 || c->grid_cols < 2)) return LSIM_E_INVALID;
    if (c->measure_heights && c->num_points_x * c->num_points_y != LSIM_NUM_HEIGHT_PTS) return LSIM_E_INVALID;
    if (c->resampling_steps <= 0 || c->max_episode_length <= 0) return LSIM_E_INVALID;
    if (c->terrain_num_rows > LSIM_TERRAIN_LEVELS_MAX || c->terrain_num_cols > LSIM_TERRAIN_TYPES_MAX) return LSIM_E_INVALID;
    if (c->solver_type != LSIM_SOLVER_PGS && c->solver_type != LSIM_SOLVER_TGS) return LSIM_E_INVALID;
    if (c->solver_type == LSIM_SOLVER_TGS && (c->num_position_iterations < 1 || c->num_position_iterations > LSIM_MAX_POSITION_ITERATIONS)) return LSIM_E_INVALID;
    // (a direct C-ABI caller with a large value would get an unbounded per-sub-step loop in the limit pass)
    if (c->tgs_limit_passes < 0 || c->tgs_limit_passes > LSIM_MAX_POSITION_ITERATIONS || (c->lin_vel_at_com != 0 && c->lin_vel_at_com != 1)) return LSIM_E_INVALID;
    return LSIM_OK;
}

static size_t ls_layout(const lsim_config* c, size_t* offsets) {
    size_t off = 0;
    for (int id = 0; id < LSIM_NUM_BUFFERS; ++id) {
        if (offsets) offsets[id] = off;
        size_t b = lsim_buffer_bytes(c, id);
        off += (b + LS_ALIGN - 1) / LS_ALIGN * LS_ALIGN;
    }
    return off;
}

extern "C" int LS_API(sizeof_config)(void) { return (int)sizeof(lsim_config); }
extern "C" int LS_API(sizeof_model)(void) { return (int)sizeof(lsim_robot_model); }
extern "C" int LS_API(abi_version)(void) { return LSIM_ABI_VERSION; }

extern "C" int LS_API(query_arena)(const lsim_config* cfg, size_t* bytes_out) {
    if (!cfg || !bytes_out) return LSIM_E_INVALID;
    int rc = ls_check_cfg(cfg);
    if (rc != LSIM_OK) return rc;
    *bytes_out = ls_layout(cfg, nullptr);
    return LSIM_OK;
}

#define LS_FAIL(s, code, ...) do { snprintf((s)->err, sizeof((s)->err), __VA_ARGS__); return (code); } while (0)

template <class T>
static int ls_upload(lsim_sim* s, int id, const std::vector<T>& v) {
    return lsbk_h2d(s->arena + s->offsets[id], v.data(), v.size() * sizeof(T));
}

// Vertex displacement of the reference's triangle mesh (isaacgym.terrain_utils.convert_heightfield_to_trimesh as called at
// TER:72-75 with slope_treshold): where the height difference to a neighbour exceeds the threshold the LOWER vertex is moved
// one cell towards the higher one, which turns the steep face into a vertical wall.  mesh_type heightfield: no correction.
// Bits 24-31 of a word (round 4): how far the highest vertex of the 4 x 4 block around the cell -- every vertex a contact query at this cell
// can meet -- lies above this vertex, in units of LSIM_MESH_DZ_UNIT height steps, rounded UP (255 = too far to say).  A collision point higher
// than that by more than its reach cannot touch anything: the query returns "no contact" after ONE load (ls_terrain_contact) -- most of a
// standing robot's 64 points, on every terrain, and on staircases it spares them the nine-cell closest-triangle loop.
#define LSIM_MESH_DZ_UNIT 4
static void ls_terrain_mesh_dzmax(const lsim_config& c, const int16_t* hf, std::vector<int32_t>& out) {
    const int R = c.grid_rows, C = c.grid_cols;
    std::vector<int16_t> rowmax((size_t)R * C);
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            int16_t m = hf[(size_t)i * C + j];
            for (int b = j - 1; b <= j + 2; ++b) if (b >= 0 && b < C && hf[(size_t)i * C + b] > m) m = hf[(size_t)i * C + b];
            rowmax[(size_t)i * C + j] = m;
        }
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            int m = rowmax[(size_t)i * C + j];
            for (int a = i - 1; a <= i + 2; ++a) if (a >= 0 && a < R && rowmax[(size_t)a * C + j] > m) m = rowmax[(size_t)a * C + j];
            int dz = (m - (int)hf[(size_t)i * C + j] + LSIM_MESH_DZ_UNIT - 1) / LSIM_MESH_DZ_UNIT;
            if (dz > 255) dz = 255;
            out[(size_t)i * C + j] = (int32_t)(((uint32_t)out[(size_t)i * C + j] & 0x00FFFFFFu) | ((uint32_t)dz << 24));
        }
}

static std::vector<int32_t> ls_terrain_mesh_flags(const lsim_config& c, const int16_t* hf);
static std::vector<int32_t> ls_terrain_mesh(const lsim_config& c, const int16_t* hf) {
    std::vector<int32_t> out = ls_terrain_mesh_flags(c, hf);
    ls_terrain_mesh_dzmax(c, hf, out);
    return out;
}
static std::vector<int32_t> ls_terrain_mesh_flags(const lsim_config& c, const int16_t* hf) {
    const int R = c.grid_rows, C = c.grid_cols;
    std::vector<int32_t> out((size_t)R * C);
    for (size_t k = 0; k < out.size(); ++k) out[k] = (int32_t)((uint32_t)(uint16_t)hf[k] | (5u << 16));
    if (c.mesh_type != 2 || c.slope_threshold <= 0.0f) return out;
    const float thr = c.slope_threshold * c.horizontal_scale / c.vertical_scale;
    auto H = [&](int i, int j) { return (float)hf[(size_t)i * C + j]; };
    std::vector<int8_t> dx((size_t)R * C, 0), dy((size_t)R * C, 0);
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            int mx = 0, my = 0, mc = 0;
            if (i + 1 < R && H(i + 1, j) - H(i, j) > thr) mx += 1;
            if (i - 1 >= 0 && H(i - 1, j) - H(i, j) > thr) mx -= 1;
            if (j + 1 < C && H(i, j + 1) - H(i, j) > thr) my += 1;
            if (j - 1 >= 0 && H(i, j - 1) - H(i, j) > thr) my -= 1;
            if (i + 1 < R && j + 1 < C && H(i + 1, j + 1) - H(i, j) > thr) mc += 1;
            if (i - 1 >= 0 && j - 1 >= 0 && H(i - 1, j - 1) - H(i, j) > thr) mc -= 1;
            dx[(size_t)i * C + j] = (int8_t)(mx + (mx == 0 ? mc : 0));
            dy[(size_t)i * C + j] = (int8_t)(my + (my == 0 ? mc : 0));
        }
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            uint8_t f = (uint8_t)((dx[(size_t)i * C + j] + 1) | ((dy[(size_t)i * C + j] + 1) << 2));
            bool any = false;
            for (int a = i - 1; a <= i + 2 && !any; ++a)
                for (int b = j - 1; b <= j + 2; ++b)
                    if (a >= 0 && a < R && b >= 0 && b < C && (dx[(size_t)a * C + b] != 0 || dy[(size_t)a * C + b] != 0)) { any = true; break; }
            if (any) f |= 16;
            out[(size_t)i * C + j] = (int32_t)((uint32_t)(uint16_t)hf[(size_t)i * C + j] | ((uint32_t)f << 16));
        }
    return out;
}

static int ls_create_impl(const lsim_config* cfg, const lsim_robot_model* model, const int16_t* height_grid,
                          const float* terrain_origins, void* arena_dev, int device_id, lsim_sim** out, lsim_sim** partial);
// include/lsim.h promises that no exception crosses the ABI: the set-up below fills std::vectors (the init-time draws, the terrain mesh words),
// whose allocation failure is a C++ exception -- caught here, what had been allocated is released, LSIM_E_NOMEM returned
extern "C" int LS_API(create)(const lsim_config* cfg, const lsim_robot_model* model, const int16_t* height_grid,
                              const float* terrain_origins, void* arena_dev, int device_id, lsim_sim** out) {
    lsim_sim* partial = nullptr;
    try {
        return ls_create_impl(cfg, model, height_grid, terrain_origins, arena_dev, device_id, out, &partial);
    } catch (...) {
        if (partial) {
            if (partial->owns_arena && partial->arena) lsbk_free(partial->arena);
            if (partial->dev_ctx) lsbk_free(partial->dev_ctx);
            free(partial);
        }
        return LSIM_E_NOMEM;
    }
}
static int ls_create_impl(const lsim_config* cfg, const lsim_robot_model* model, const int16_t* height_grid,
                          const float* terrain_origins, void* arena_dev, int device_id, lsim_sim** out, lsim_sim** partial) {
    if (!cfg || !model || !out) return LSIM_E_INVALID;
    int rc = ls_check_cfg(cfg);
    if (rc != LSIM_OK) return rc;
    if (cfg->mesh_type != 0 && (!height_grid || !terrain_origins)) return LSIM_E_INVALID;
    if (model->num_collision_points > LSIM_MAX_COLLISION_POINTS || model->num_collision_points < 0) return LSIM_E_INVALID;
    lsim_sim* s = (lsim_sim*)calloc(1, sizeof(lsim_sim));
    if (!s) return LSIM_E_NOMEM;
    s->cfg = *cfg; s->model = *model; s->device_id = device_id;
    const lsim_config& c = s->cfg;
    const int N = c.num_envs;
    if (lsbk_set_device(device_id) != 0) { free(s); return LSIM_E_HIP; }
    size_t total = ls_layout(&c, s->offsets);
    if (arena_dev) { s->arena = (char*)arena_dev; s->owns_arena = false; }
    else {
        if (lsbk_malloc((void**)&s->arena, total) != 0) { free(s); return LSIM_E_NOMEM; }
        s->owns_arena = true;
    }
    if (lsbk_memset(s->arena, 0, total) != 0) { if (s->owns_arena) lsbk_free(s->arena); free(s); return LSIM_E_HIP; }
    *partial = s;        // from here on an exception (std::bad_alloc of the vectors below) is cleaned up by the caller

    // ---- init-time draws (LR:999-1032, LR:1172-1179, LR:506-513, LR:1232-1239), identical to the oracle's
    const uint32_t W = 0xFFFFFFFFu;
    auto u = [&](int env, uint32_t tag, uint32_t idx) { return ls_u01(c.seed, c.rank, (uint32_t)env, W, tag, idx); };
    std::vector<float> ms(12 * N), kp(N), kd(N), msf(N), pay(N), com(3 * N), fr(N), root(13 * N), org(3 * N, 0.0f);
    std::vector<int64_t> lvl(N, 0), typ(N, 0);
    std::vector<uint8_t> rst(N, 1);
    for (int e = 0; e < N; ++e) {
        for (int j = 0; j < 12; ++j)
            ms[12 * e + j] = c.randomize_motor_strength ? rand_range(u(e, LSIM_RNG_INIT, j), c.motor_strength_range[0], c.motor_strength_range[1]) : 1.0f;
        kp[e] = c.randomize_kp ? rand_range(u(e, LSIM_RNG_INIT, 12), c.kp_range[0], c.kp_range[1]) : 1.0f;
        kd[e] = c.randomize_kd ? rand_range(u(e, LSIM_RNG_INIT, 13), c.kd_range[0], c.kd_range[1]) : 1.0f;
        msf[e] = c.randomize_motor_strength ? rand_range(u(e, LSIM_RNG_INIT, 14), c.motor_strength_range[0], c.motor_strength_range[1]) : 1.0f;
        pay[e] = c.randomize_payload_mass ? rand_range(u(e, LSIM_RNG_INIT, 15), c.payload_mass_range[0], c.payload_mass_range[1]) : 0.0f;
        for (int k = 0; k < 3; ++k)
            com[3 * e + k] = c.randomize_com_displacement ? rand_range(u(e, LSIM_RNG_INIT, 16 + k), c.com_displacement_range[0], c.com_displacement_range[1]) : 0.0f;
        if (c.randomize_friction) {
            int bucket = (int)(u(e, LSIM_RNG_INIT, 19) * 64.0f);
            fr[e] = rand_range(u(bucket, LSIM_RNG_INIT_BUCKET, 0), c.friction_range[0], c.friction_range[1]);
        } else fr[e] = 1.0f;
        if (c.mesh_type != 0) {
            int max_init = c.terrain_curriculum ? c.max_init_terrain_level : c.terrain_num_rows - 1;
            lvl[e] = (int64_t)(u(e, LSIM_RNG_INIT, 20) * (float)(max_init + 1));
            typ[e] = lsim_terrain_type_of_env(e, N, c.terrain_num_cols);     // LR:1234 (torch's floor division, see lsim_layout.h)
            for (int k = 0; k < 3; ++k) org[3 * e + k] = terrain_origins[(lvl[e] * c.terrain_num_cols + typ[e]) * 3 + k];
        }
        for (int k = 0; k < 13; ++k) root[13 * e + k] = c.base_init_state[k];
        for (int k = 0; k < 3; ++k) root[13 * e + k] += org[3 * e + k];
    }
    std::vector<float> stats(2 * LSIM_STATS_SIZE, 0.0f);
    for (int r = 0; r < 2; ++r) for (int i = 0; i < 4; ++i) for (int k = 0; k < 2; ++k) stats[r * LSIM_STATS_SIZE + LSIM_STATS_CMD_RANGES + 2 * i + k] = c.command_ranges[i][k];
    int bad = 0;
    bad |= ls_upload(s, LSIM_BUF_MOTOR_STRENGTH, ms); bad |= ls_upload(s, LSIM_BUF_KP_FACTORS, kp); bad |= ls_upload(s, LSIM_BUF_KD_FACTORS, kd);
    bad |= ls_upload(s, LSIM_BUF_MOTOR_STRENGTH_FACTORS, msf); bad |= ls_upload(s, LSIM_BUF_PAYLOAD, pay); bad |= ls_upload(s, LSIM_BUF_COM_DISPLACEMENT, com);
    bad |= ls_upload(s, LSIM_BUF_FRICTION, fr); bad |= ls_upload(s, LSIM_BUF_ROOT_STATES, root); bad |= ls_upload(s, LSIM_BUF_ENV_ORIGINS, org);
    bad |= ls_upload(s, LSIM_BUF_TERRAIN_LEVELS, lvl); bad |= ls_upload(s, LSIM_BUF_TERRAIN_TYPES, typ); bad |= ls_upload(s, LSIM_BUF_RESET, rst);
    bad |= ls_upload(s, LSIM_BUF_STATS, stats);
    if (c.mesh_type != 0) {
        std::vector<int32_t> mesh = ls_terrain_mesh(c, height_grid);
        bad |= ls_upload(s, LSIM_BUF_TERRAIN_MESH, mesh);
        bad |= lsbk_h2d(s->arena + s->offsets[LSIM_BUF_HEIGHT_GRID], height_grid, lsim_buffer_bytes(&c, LSIM_BUF_HEIGHT_GRID));
        bad |= lsbk_h2d(s->arena + s->offsets[LSIM_BUF_TERRAIN_ORIGINS], terrain_origins, lsim_buffer_bytes(&c, LSIM_BUF_TERRAIN_ORIGINS));
    }
    // ---- device context
    LsCtx& h = s->host_ctx;
    memset(&h, 0, sizeof(h));
    h.cfg = c; h.model = s->model;
    for (int id = 0; id < LSIM_NUM_BUFFERS; ++id) h.buf[id] = s->arena + s->offsets[id];
    h.accum = (float*)h.buf[LSIM_BUF_STATS];
    h.num_active = 0;
    for (int id = 0; id < LSIM_NUM_REWARD_TERMS; ++id)
        if (id != LSIM_R_TERMINATION && c.reward_scales[id] != 0.0f) { h.active_scales[h.num_active] = c.reward_scales[id]; h.active_terms[h.num_active++] = id; }
    for (int ai = 0; ai < h.num_active; ++ai) {       // the (term, part) items of ph_reward_parts, whole terms while the table has room
        const int id = h.active_terms[ai], n = ls_reward_num_parts(id);
        if (n == 0 || h.num_part_items + n > LS_MAX_PART_ITEMS) continue;
        for (int j = 0; j < n; ++j) h.part_items[h.num_part_items++] = (uint16_t)((id << 10) | (ai << 4) | j);
        h.parted_mask |= 1ull << ai;
    }
    if (lsbk_malloc((void**)&s->dev_ctx, sizeof(LsCtx)) != 0) bad = 1;
    else bad |= lsbk_h2d(s->dev_ctx, &h, sizeof(LsCtx));
    if (bad) { *partial = nullptr; if (s->owns_arena) lsbk_free(s->arena); if (s->dev_ctx) lsbk_free(s->dev_ctx); free(s); return LSIM_E_HIP; }
    s->step_counter = 0;
    s->priority_max_envs = 32768;
    if (const char* e = getenv("LSIM_PRIORITY_MAX_ENVS")) s->priority_max_envs = atoi(e);
    s->init_done = 1;   // construction completes before the runner's first reset (LR:116, HIMR:84)
    *partial = nullptr;
    *out = s;
    return LSIM_OK;
}

extern "C" int LS_API(get_buffer)(lsim_sim* s, int id, void** dev_ptr, int64_t shape[4], int* ndim, int* dtype) {
    if (!s || !dev_ptr || id < 0 || id >= LSIM_NUM_BUFFERS) return LSIM_E_INVALID;
    *dev_ptr = s->arena + s->offsets[id];
    return lsim_buffer_desc(&s->cfg, id, shape, ndim, dtype);
}

extern "C" int LS_API(step_ex)(lsim_sim* s, const float* actions_dev, uint32_t flags, void* stream) {
    if (!s || !actions_dev) return LSIM_E_INVALID;
    // the host-side state (step counter, stats row) advances only AFTER every launch of the step was accepted: a failed launch (invalid stream,
    // device lost) leaves the handle where it was -- nothing ran, the call can be repeated
    const int64_t step = s->step_counter + 1;   // LR:194
    LsStepArgs a;
    memset(&a, 0, sizeof(a));
    a.actions = actions_dev; a.step_counter = step; a.flags = flags; a.init_done = s->init_done;
    a.row_in = s->stats_row; a.row_out = s->stats_row ^ 1; a.reset_all = 0;
    // Kernel B exists for the step's one global dependency; it only bites when the command curriculum evaluates (LR:307: one step in
    // max_episode_length).  On every other step kernel A runs B's per-env work itself and a few blocks finish the step (ls_kernels.h).
    const bool curriculum_step = s->cfg.commands_curriculum && (step % s->cfg.max_episode_length == 0);
    a.fuse_tail = (!curriculum_step && !(flags & LSIM_STEP_TWO_KERNELS)) ? 1 : 0;
    // contact-count wave priorities pay while a launch is a few rounds of waves (4096 resident at a time): +6.5 % at N = 4096, +3.7 % at 8192,
    // +2.1 % at 12 288, +1.2 % at 16 384, nothing at 32 768; with many rounds in flight the slowest wave of a round hides behind the next
    // and the priorities only perturb the arbiter (-0.4 % at N = 65 536 / 262 144)
    if (s->cfg.num_envs > s->priority_max_envs) a.flags |= LSIM_STEP_FLAT_PRIORITY;
    lsbk_prof_mark(s, 0, stream);
    if (lsbk_launch_a(s, a, stream) != 0) LS_FAIL(s, LSIM_E_HIP, "kernel A launch failed");
    lsbk_prof_mark(s, 1, stream);
    // (kernel A was accepted: from here on the step HAS begun on the device, so the counters advance even if the second launch fails --
    // the error is reported, and the handle's state matches what the device will have done)
    s->step_counter = step;
    s->stats_row = a.row_out;
#if defined(LS_EXP_NO_FINISH)      // timing probe only (wrong statistics): what the finish launch costs on the rollout's critical path
    if (a.fuse_tail) { }
#else
    if (a.fuse_tail) { if (lsbk_launch_finish(s, a, stream) != 0) LS_FAIL(s, LSIM_E_HIP, "finish kernel launch failed"); }
#endif
    else if (lsbk_launch_b(s, a, stream) != 0) LS_FAIL(s, LSIM_E_HIP, "kernel B launch failed");
    lsbk_prof_mark(s, 2, stream);
    return LSIM_OK;
}
extern "C" int LS_API(step)(lsim_sim* s, const float* actions_dev, void* stream) { return LS_API(step_ex)(s, actions_dev, LSIM_STEP_DEFAULT, stream); }

extern "C" int LS_API(reset_all)(lsim_sim* s, void* stream) {
    if (!s) return LSIM_E_INVALID;
    LsStepArgs a;
    memset(&a, 0, sizeof(a));
    a.actions = nullptr; a.step_counter = s->step_counter; a.flags = 0; a.init_done = s->init_done;
    a.row_in = s->stats_row; a.row_out = s->stats_row ^ 1; a.reset_all = 1;
    if (lsbk_launch_reduce(s, a, stream) != 0) LS_FAIL(s, LSIM_E_HIP, "reduction kernel launch failed");
    s->stats_row = a.row_out;      // (after the first accepted launch, as in lsim_step_ex)
    if (lsbk_launch_b(s, a, stream) != 0) LS_FAIL(s, LSIM_E_HIP, "reset kernel launch failed");
    return LSIM_OK;
}

extern "C" int LS_API(reset_envs)(lsim_sim* s, const uint8_t* mask_dev, void* stream) {
    if (!s || !mask_dev) return LSIM_E_INVALID;
    LsStepArgs a;
    memset(&a, 0, sizeof(a));
    a.actions = nullptr; a.step_counter = s->step_counter; a.flags = 0; a.init_done = s->init_done;
    a.row_in = s->stats_row; a.row_out = s->stats_row ^ 1; a.reset_all = 2; a.reset_mask = mask_dev;
    a.rng_salt = (s->reset_calls + 1) * 0x9E3779B9u;
    if (lsbk_launch_reduce(s, a, stream) != 0) LS_FAIL(s, LSIM_E_HIP, "reduction kernel launch failed");
    s->reset_calls += 1;
    s->stats_row = a.row_out;
    if (lsbk_launch_b(s, a, stream) != 0) LS_FAIL(s, LSIM_E_HIP, "reset kernel launch failed");
    return LSIM_OK;
}

extern "C" int LS_API(get_step_counter)(lsim_sim* s, int64_t* out) { if (!s || !out) return LSIM_E_INVALID; *out = s->step_counter; return LSIM_OK; }
extern "C" int LS_API(set_step_counter)(lsim_sim* s, int64_t v) { if (!s) return LSIM_E_INVALID; s->step_counter = v; return LSIM_OK; }
extern "C" int LS_API(get_reset_calls)(lsim_sim* s, uint32_t* out) { if (!s || !out) return LSIM_E_INVALID; *out = s->reset_calls; return LSIM_OK; }
extern "C" int LS_API(set_reset_calls)(lsim_sim* s, uint32_t v) { if (!s) return LSIM_E_INVALID; s->reset_calls = v; return LSIM_OK; }
extern "C" int LS_API(get_stats_row)(lsim_sim* s, int* row) { if (!s || !row) return LSIM_E_INVALID; *row = s->stats_row; return LSIM_OK; }
extern "C" const char* LS_API(reward_name)(int id) { return (id >= 0 && id < LSIM_NUM_REWARD_TERMS) ? lsim_reward_names[id] : nullptr; }
extern "C" const char* LS_API(buffer_name)(int id) { return (id >= 0 && id < LSIM_NUM_BUFFERS) ? lsim_buffer_names[id] : nullptr; }
extern "C" const char* LS_API(last_error)(lsim_sim* s) { return s ? s->err : "null handle"; }
extern "C" void LS_API(destroy)(lsim_sim* s) {
    if (!s) return;
    lsbk_prof_free(s);
    if (s->owns_arena && s->arena) lsbk_free(s->arena);
    if (s->dev_ctx) lsbk_free(s->dev_ctx);
    free(s);
}
