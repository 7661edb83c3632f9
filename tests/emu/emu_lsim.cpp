// TEST INFRASTRUCTURE -- CPU lane emulator for the kernel sources in isaacgymloco_amd/csrc/.
// Compiles the very same phase functions that run on the GPU (ls_kernels.h) with g++, executing the 64 lanes
// of each phase in a loop (LS_EMU).  Purpose: validate the lane orchestration / LDS hand-offs of the HIP
// kernels against the oracle in the CPU-only build container.  It is NOT a product path: nothing under
// isaacgymloco_amd/ loads this library, and the Python env refuses to run without the HIP library.
#define LS_EMU 1
#define LS_API(name) emu_##name
#include <stdlib.h>
#include <string.h>
struct lsim_sim;
struct LsStepArgs;
static int lsbk_set_device(int) { return 0; }
static int lsbk_malloc(void** p, size_t n) { *p = malloc(n); return *p ? 0 : 1; }
static void lsbk_free(void* p) { free(p); }
static int lsbk_h2d(void* d, const void* s, size_t n) { memcpy(d, s, n); return 0; }
static int lsbk_memset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static int lsbk_launch_a(lsim_sim* s, const LsStepArgs& a, void* stream);
static int lsbk_launch_b(lsim_sim* s, const LsStepArgs& a, void* stream);
static int lsbk_launch_reduce(lsim_sim* s, const LsStepArgs& a, void* stream);
static int lsbk_launch_finish(lsim_sim* s, const LsStepArgs& a, void* stream);
static void lsbk_prof_mark(lsim_sim*, int, void*) {}
static void lsbk_prof_free(lsim_sim*) {}
#include "../../isaacgymloco_amd/csrc/ls_api_impl.h"
#include "../../isaacgymloco_amd/csrc/ls_kernels.h"

static int lsbk_launch_a(lsim_sim* s, const LsStepArgs& a, void*) {
    static WaveShared sh;
    static LaneRegs L[64];
    for (int env = 0; env < s->cfg.num_envs; ++env) {
        memset(&sh, 0xFF, sizeof(sh));   // poison with NaNs (0xFFFFFFFF): phases must not rely on stale LDS -- not even "times zero" (round 5: a finite poison hid exactly that)
        memset(L, 0xFF, sizeof(L));
        if (s->cfg.solver_type == LSIM_SOLVER_TGS) ls_wave_step_a<LSIM_SOLVER_TGS>(*s->dev_ctx, a, env, sh, L);
        else ls_wave_step_a<LSIM_SOLVER_PGS>(*s->dev_ctx, a, env, sh, L);
    }
    return 0;
}
static int lsbk_launch_b(lsim_sim* s, const LsStepArgs& a, void*) {
    static WaveShared sh;
    static LaneRegs L[64];
    for (int env = 0; env < s->cfg.num_envs; ++env) {
        memset(&sh, 0xFF, sizeof(sh));
        memset(L, 0xFF, sizeof(L));
        ls_wave_step_b(*s->dev_ctx, a, env, sh, L);
    }
    return 0;
}
static int lsbk_launch_finish(lsim_sim* s, const LsStepArgs& a, void*) {
    const LsCtx& cx = *s->dev_ctx;
    for (int env = 0; env < s->cfg.num_envs; ++env) ls_step_finish_env(cx, a, env);
    for (int t = 0; t < 256; ++t) ls_step_finish_rows(cx, a, t);
    return 0;
}
static int lsbk_launch_reduce(lsim_sim* s, const LsStepArgs& a, void*) {   // bare reset_idx: track sum (and count) over the resetting envs (LR:875)
    const LsCtx& cx = *s->dev_ctx;
    long long acc = 0;
    int n = 0;
    for (int env = 0; env < s->cfg.num_envs; ++env) {
        if (a.reset_all == 2 && !a.reset_mask[env]) continue;
        acc += ls_to_fix(LSB(cx, LSIM_BUF_EPISODE_SUMS, float)[env * LSIM_NUM_REWARD_TERMS + LSIM_R_TRACKING_LIN_VEL]);
        n += 1;
    }
    ls_fix_row(cx, a.row_out)[LSIM_STATS_FIX_TRACK] = acc;
    if (a.reset_all == 2) cx.accum[a.row_out * LSIM_STATS_SIZE + LSIM_STATS_RESET_COUNT] = (float)n;
    return 0;
}
extern "C" int emu_sizeof_shared(void) { return (int)sizeof(WaveShared); }
