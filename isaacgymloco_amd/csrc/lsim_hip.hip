// lsim_hip.hip -- the MI355X (gfx950) library behind include/lsim.h.
// One wavefront (64-thread workgroup) per robot; per-robot scratch in LDS (WaveShared); XCD-aware block -> env map.
// This translation unit is the SIMULATOR (kernels A / B, the C-ABI of lsim_create .. lsim_destroy); the rollout / learner / policy kernels
// are lsim_learn.hip, compiled with IEEE division / square root and denormals kept (build.py states both flag sets and why).
#include <hip/hip_runtime.h>

#define LS_API(name) lsim_##name
struct lsim_sim;
struct LsStepArgs;
static int lsbk_set_device(int d) { return hipSetDevice(d) == hipSuccess ? 0 : 1; }
static int lsbk_malloc(void** p, size_t n) { return hipMalloc(p, n) == hipSuccess ? 0 : 1; }
static void lsbk_free(void* p) { (void)hipFree(p); }
static int lsbk_h2d(void* d, const void* s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1; }
static int lsbk_memset(void* d, int v, size_t n) { return hipMemset(d, v, n) == hipSuccess ? 0 : 1; }
static int lsbk_launch_a(lsim_sim* s, const LsStepArgs& a, void* stream);
static int lsbk_launch_b(lsim_sim* s, const LsStepArgs& a, void* stream);
static int lsbk_launch_reduce(lsim_sim* s, const LsStepArgs& a, void* stream);
static int lsbk_launch_finish(lsim_sim* s, const LsStepArgs& a, void* stream);
static void lsbk_prof_mark(lsim_sim* s, int which, void* stream);
static void lsbk_prof_free(lsim_sim* s);

#include "ls_api_impl.h"
#include "ls_kernels.h"

// Each XCD (8 per chip, block b is dispatched to XCD b % 8) works on its own slices of the env range, so a
// robot's state lines stay in one XCD's L2 and neighbouring robots do not false-share lines across XCD L2s.
// Round 4: the slices are interleaved in GROUPS OF 32 envs (XCD x owns groups x, x + 8, x + 16, ...) instead of one contiguous eighth each.
// A group of 32 keeps the lines of the per-env 4-byte buffers (128 B = 32 envs) inside one XCD; interleaving spreads the terrain types -- which
// the reference assigns by env index, LR:1234, ~205 consecutive envs per type -- over all XCDs.  With contiguous eighths one XCD held only
// staircase robots (whose narrow phase is the slow one) and another only flat-ground robots, and at N = 4096, where every wave is resident at
// once, the kernel ends with the slowest XCD.
__device__ __forceinline__ int ls_env_of_block(int b, int num_envs) {
    const int l = b >> 3;
    return ((((l >> 5) << 3) + (b & 7)) << 5) + (l & 31);
    (void)num_envs;
}

// Kernel A is bound by per-wave latency (dependent VALU chains, LDS round trips at ~19 phase boundaries per sub-step):
// 4 waves per SIMD (<= 128 VGPRs, no scratch) beat 1-2 waves with more registers by 1.35x at N = 4096, where 4 waves/SIMD is also exactly
// the whole batch resident at once (4096 waves / 1024 SIMDs; the 9.6 KB LDS struct allows 16 blocks per CU).
// One kernel per solver (lsim_config.solver_type: the host picks at launch), so that neither carries the other's code: lsim_k_step_a_tgs is
// what every reference config runs (LRC:245), lsim_k_step_a_pgs the build's earlier velocity-level solver.
#define LS_KERNEL_A(name, SOLVER) \
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void name(const LsCtx* __restrict__ ctx, LsStepArgs a) { \
    __shared__ WaveShared sh; \
    const int env = ls_env_of_block((int)blockIdx.x, ctx->cfg.num_envs); \
    if (env >= ctx->cfg.num_envs) return; \
    LaneRegs rg; \
    ls_wave_step_a<SOLVER>(*ctx, a, env, sh, rg, (int)threadIdx.x); \
}
LS_KERNEL_A(lsim_k_step_a_pgs, LSIM_SOLVER_PGS)
LS_KERNEL_A(lsim_k_step_a_tgs, LSIM_SOLVER_TGS)

__global__ __launch_bounds__(64) void lsim_k_step_b(const LsCtx* __restrict__ ctx, LsStepArgs a) {
    __shared__ WaveShared sh;
    const int env = ls_env_of_block((int)blockIdx.x, ctx->cfg.num_envs);
    if (env >= ctx->cfg.num_envs) return;
    LaneRegs rg;
    ls_wave_step_b(*ctx, a, env, sh, rg, (int)threadIdx.x);
}

// after a kernel A with the fused tail: the mask copy that depends on whether any env reset, and the stats rows (ls_kernels.h)
__global__ __launch_bounds__(256) void lsim_k_step_finish(const LsCtx* __restrict__ ctx, LsStepArgs a) {
    const LsCtx& cx = *ctx;
    static_assert(LSIM_STATS_FIX_WORDS <= 256 && LSIM_NUM_REWARD_TERMS <= 256, "one block covers the stats rows");
    for (int env = (int)(blockIdx.x * blockDim.x + threadIdx.x); env < cx.cfg.num_envs; env += (int)(gridDim.x * blockDim.x)) ls_step_finish_env(cx, a, env);
    if (blockIdx.x == 0) ls_step_finish_rows(cx, a, (int)threadIdx.x);
}

// reset_idx outside a step: sum of episode_sums["tracking_lin_vel"] over the resetting envs for the command curriculum (LR:875) and, for a
// subset (reset_all == 2), their number
__global__ __launch_bounds__(256) void lsim_k_track_sum(const LsCtx* __restrict__ ctx, LsStepArgs a) {
    __shared__ long long part[256];
    __shared__ int count[256];
    const LsCtx& cx = *ctx;
    long long acc = 0;
    int n = 0;
    // every env's value goes to fixed point on its own (ls_to_fix clamps ONE addend to +-2^20: converting a block partial instead would clamp
    // the sum of N / 64 envs silently beyond ~3 M envs); the int64 partials add exactly, in any order
    for (int env = (int)(blockIdx.x * blockDim.x + threadIdx.x); env < cx.cfg.num_envs; env += (int)(gridDim.x * blockDim.x)) {
        if (a.reset_all == 2 && !LS_G(const uint8_t, a.reset_mask)[env]) continue;
        acc += ls_to_fix(LSB(cx, LSIM_BUF_EPISODE_SUMS, float)[env * LSIM_NUM_REWARD_TERMS + LSIM_R_TRACKING_LIN_VEL]);
        n += 1;
    }
    part[threadIdx.x] = acc;
    count[threadIdx.x] = n;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { part[threadIdx.x] += part[threadIdx.x + s]; count[threadIdx.x] += count[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        LS_ATOMIC_ADD_I64(ls_fix_row(cx, a.row_out) + LSIM_STATS_FIX_TRACK, part[0]);
        if (a.reset_all == 2 && count[0] > 0)     // integer-valued: exact in any order
            LS_ATOMIC_ADD(LS_G(float, cx.accum) + a.row_out * LSIM_STATS_SIZE + LSIM_STATS_RESET_COUNT, (float)count[0]);
    }
}

static int ls_grid(const lsim_sim* s) { return 8 * 32 * ((((s->cfg.num_envs + 31) / 32) + 7) / 8); }     // whole groups of 32 envs for each of the 8 XCDs

static int lsbk_launch_a(lsim_sim* s, const LsStepArgs& a, void* stream) {
    if (s->cfg.solver_type == LSIM_SOLVER_TGS) hipLaunchKernelGGL(lsim_k_step_a_tgs, dim3(ls_grid(s)), dim3(64), 0, (hipStream_t)stream, (const LsCtx*)s->dev_ctx, a);
    else hipLaunchKernelGGL(lsim_k_step_a_pgs, dim3(ls_grid(s)), dim3(64), 0, (hipStream_t)stream, (const LsCtx*)s->dev_ctx, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
static int lsbk_launch_b(lsim_sim* s, const LsStepArgs& a, void* stream) {
    hipLaunchKernelGGL(lsim_k_step_b, dim3(ls_grid(s)), dim3(64), 0, (hipStream_t)stream, (const LsCtx*)s->dev_ctx, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
static int lsbk_launch_finish(lsim_sim* s, const LsStepArgs& a, void* stream) {
    int blocks = (s->cfg.num_envs + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(lsim_k_step_finish, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const LsCtx*)s->dev_ctx, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
static int lsbk_launch_reduce(lsim_sim* s, const LsStepArgs& a, void* stream) {
    int blocks = (s->cfg.num_envs + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(lsim_k_track_sum, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const LsCtx*)s->dev_ctx, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- measurement aid: HIP events around the two kernels of each step, on the caller's stream (lsim_set_profiling)
struct LsProf {
    int capacity;
    long long count;          // steps recorded so far
    hipEvent_t* ev;           // [capacity][3]
};
static void lsbk_prof_mark(lsim_sim* s, int which, void* stream) {
    LsProf* p = (LsProf*)s->prof;
    if (!p) return;
    int slot = (int)(p->count % p->capacity);
    (void)hipEventRecord(p->ev[3 * slot + which], (hipStream_t)stream);
    if (which == 2) p->count += 1;
}
static void lsbk_prof_free(lsim_sim* s) {
    LsProf* p = (LsProf*)s->prof;
    if (!p) return;
    for (int i = 0; i < 3 * p->capacity; ++i) (void)hipEventDestroy(p->ev[i]);
    free(p->ev);
    free(p);
    s->prof = nullptr;
}
extern "C" int lsim_set_profiling(lsim_sim* s, int capacity) {
    if (!s || capacity < 0) return LSIM_E_INVALID;
    lsbk_prof_free(s);
    if (capacity == 0) return LSIM_OK;
    LsProf* p = (LsProf*)calloc(1, sizeof(LsProf));
    p->capacity = capacity;
    p->ev = (hipEvent_t*)calloc((size_t)3 * capacity, sizeof(hipEvent_t));
    for (int i = 0; i < 3 * capacity; ++i)
        if (hipEventCreate(&p->ev[i]) != hipSuccess) return LSIM_E_HIP;
    s->prof = p;
    return LSIM_OK;
}
extern "C" int lsim_read_profile(lsim_sim* s, float* ms_a, float* ms_b, int* n_inout) {
    if (!s || !ms_a || !ms_b || !n_inout) return LSIM_E_INVALID;
    LsProf* p = (LsProf*)s->prof;
    if (!p) { *n_inout = 0; return LSIM_OK; }
    long long avail = p->count < p->capacity ? p->count : p->capacity;
    int n = (int)(avail < *n_inout ? avail : *n_inout);
    for (int i = 0; i < n; ++i) {
        long long step = p->count - n + i;
        int slot = (int)(step % p->capacity);
        if (hipEventSynchronize(p->ev[3 * slot + 2]) != hipSuccess) return LSIM_E_HIP;
        if (hipEventElapsedTime(&ms_a[i], p->ev[3 * slot], p->ev[3 * slot + 1]) != hipSuccess) return LSIM_E_HIP;
        if (hipEventElapsedTime(&ms_b[i], p->ev[3 * slot + 1], p->ev[3 * slot + 2]) != hipSuccess) return LSIM_E_HIP;
    }
    *n_inout = n;
    return LSIM_OK;
}

#if defined(LS_DEBUG_KIN)
// diagnostics build only (tools/kin_check.py): run one of the two kinematics forms on a given state and return every array it fills
__global__ __launch_bounds__(64) void lsim_k_debug_kin(const LsCtx* __restrict__ ctx, const float* __restrict__ in, float* __restrict__ out, int variant) {
    __shared__ WaveShared sh;
    const int lane = (int)threadIdx.x;
    const LsCtx& cx = *ctx;
    if (lane < 13) sh.root[lane] = in[lane];
    if (lane < 12) { sh.q[lane] = in[13 + lane]; sh.qd[lane] = in[25 + lane]; }
    if (lane >= 16 && lane < 16 + LS_NB) {
        const lsim_body& b = cx.model.bodies[lane - 16];
        LsBodyLds& o = sh.body[lane - 16];
        o.mass = b.mass;
        for (int k = 0; k < 3; ++k) { o.com[k] = b.com[k]; o.jpos[k] = b.joint_pos[k]; o.axis[k] = b.joint_axis[k]; }
        for (int k = 0; k < 6; ++k) o.inertia[k] = b.inertia[k];
    }
    __syncthreads();
    if (variant == 0) ph_kinematics(sh, lane); else wc_kinematics(sh, lane);
    __syncthreads();
    for (int k = lane; k < 153; k += 64) out[k] = (&sh.R[0][0])[k];
    for (int k = lane; k < 51; k += 64) out[153 + k] = (&sh.p[0][0])[k];
    for (int k = lane; k < 72; k += 64) out[204 + k] = (&sh.S[0][0])[k];
    for (int k = lane; k < 102; k += 64) out[276 + k] = (&sh.V[0][0])[k];
    for (int k = lane; k < 102; k += 64) out[378 + k] = (&sh.Ab[0][0])[k];
}
extern "C" int lsim_debug_kinematics(lsim_sim* s, const float* in_host, float* out_host, int variant) {
    float *din, *dout;
    if (hipMalloc(&din, 37 * 4) != hipSuccess || hipMalloc(&dout, 480 * 4) != hipSuccess) return 1;
    (void)hipMemcpy(din, in_host, 37 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(lsim_k_debug_kin, dim3(1), dim3(64), 0, 0, (const LsCtx*)s->dev_ctx, din, dout, variant);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    (void)hipMemcpy(out_host, dout, 480 * 4, hipMemcpyDeviceToHost);
    (void)hipFree(din); (void)hipFree(dout);
    return 0;
}
#endif

#if defined(LS_PHASE_TIMING)
// diagnostics build only: read and clear the per-site tick / call accumulators (tools/phase_profile.py)
extern "C" int lsim_debug_read_phase_ticks(unsigned long long* ticks, unsigned long long* calls) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(ticks, HIP_SYMBOL(g_ls_phase_ticks), 128 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(calls, HIP_SYMBOL(g_ls_phase_calls), 128 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[128] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ls_phase_ticks), z, sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ls_phase_calls), z, sizeof(z));
    return 0;
}
#endif

#if defined(LS_PHASE_TIMING)
// the per-site ticks again by kind of wave ([3][129]: few contacts / many contacts / resetting; last column = waves), read and cleared
extern "C" int lsim_debug_read_phase_ticks_by(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ls_phase_ticks_by), 3 * 129 * sizeof(unsigned long long)) != hipSuccess) return 1;
    static unsigned long long z[3 * 129];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ls_phase_ticks_by), z, sizeof(z));
    return 0;
}
#endif

#if defined(LS_WAVE_TIMES)
// diagnostics build only: per-env (start, end, ticks, hardware id) of the latest kernel A (tools/wave_times.py)
extern "C" int lsim_debug_read_wave_times(unsigned long long* out, int n) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ls_wave_times), (size_t)4 * n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int lsim_debug_read_wave_checkpoints(unsigned int* out, int n) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ls_wave_cp), (size_t)16 * n * sizeof(unsigned int)) == hipSuccess ? 0 : 1;
}
#endif
