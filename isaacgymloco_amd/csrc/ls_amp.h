// ls_amp.h -- the AMP rollout step in ONE launch (include/lsim.h, lsim_amp_step; SURVEY.md 8(f)4 "discriminator reward fused after the step").
//
// Per rollout step the reference's HybridPolicyRunner (rsl_rl/runners/hybrid_runner.py:183-200) patches the terminal AMP states of resetting
// envs into the next AMP observation, asks the discriminator for the style reward of every (state, next state) pair
// (rsl_rl/algorithms/amp_discriminator.py:55-72: running-moment normalisation rsl_rl/utils/utils.py:124-130, trunk Linear + ReLU x 2, linear
// head, clamp(1 - (d - 1)^2 / 4, 0) * coef, lerp with the task reward) and inserts the pair into the policy replay ring
// (rsl_rl/storage/replay_buffer.py:52-68) -- ~25 torch launches at N = 4096.  Here a workgroup of 16 waves owns 32 environments:
//   stage     thread (row, c) loads column c of its env's previous AMP observation and of the next one (terminal row where done), writes the
//             raw pair to the replay ring slot (cursor + env) % capacity and the un-patched next observation to the carry buffer, normalises
//             both with the fp64 running moments and puts them into LDS as the 2 D-wide input row
//   trunk[0]  2 D -> H1 (1024) on v_mfma_f32_16x16x4_f32, bias + ReLU, activations stay in LDS (ls_policy.h's layer, row stride 1040)
//   trunk[1]  H1 -> H2 (512); its epilogue applies bias + ReLU and multiplies by the head's weights at once: the layer's output is never
//             stored, each wave leaves one partial dot product per row
//   finish    d = sum of the partials + head bias; the style reward; rewards_out / disc_out
// gridDim.y = NSPLIT blocks share one row group: each evaluates trunk[0] (11 % of the work) and 1 / NSPLIT of trunk[1]'s output tiles, so
// that 4096 envs fill 256 CUs instead of 128; the last block of a group to arrive (device-scope ticket; the partial sums travel as write-through
// sc1 stores acknowledged before the ticket is drawn, see the exchange below -- no L2 write-back fence) adds the partial sums in a fixed order --
// the result does not depend on which block that is.
#pragma once
#include <hip/hip_runtime.h>

#define LS_AMP_ROWS 32
#define LS_AMP_WAVES 16
#define LS_AMP_MAX_DIM 32                /* AMP observation width D (30, LR:416): the input row 2 D <= 64 */
#define LS_AMP_STRIDE_IN 80
#define LS_AMP_STRIDE_H 1040
#define LS_AMP_MAX_H1 1024
static_assert(LS_AMP_STRIDE_IN % 64 == 16 && LS_AMP_STRIDE_H % 64 == 16, "row strides of 4 slots (mod 16 slots): ls_policy.h's conflict-free swizzle");
static_assert(LS_AMP_STRIDE_IN >= 2 * LS_AMP_MAX_DIM && LS_AMP_STRIDE_H >= LS_AMP_MAX_H1, "strides hold the widest row");
#define LS_AMP_O_IN 0
#define LS_AMP_O_H (LS_AMP_O_IN + LS_AMP_ROWS * LS_AMP_STRIDE_IN)
#define LS_AMP_O_PART (LS_AMP_O_H + LS_AMP_ROWS * LS_AMP_STRIDE_H)
#define LS_AMP_LDS_FLOATS (LS_AMP_O_PART + LS_AMP_WAVES * LS_AMP_ROWS)

struct LsAmpArgs {
    lsim_amp_disc d;
    const float* prev; const float* next; const uint8_t* dones; const float* term; const float* task_rewards;
    long num_envs;
    float* rewards_out; float* disc_out; float* carry_out;
    float* replay_s; float* replay_ns;
    long replay_cap, replay_cursor;
    float* part; unsigned int* counters;
};

// trunk[1] for this wave's NTW output tiles: relu(acc + b) . head_w summed over the wave's tiles and over the four k groups of a row; lanes 0-15
// leave part[wave][row + 16 h]
template <int NTW, int RH, bool FULL>
__device__ __noinline__ void ls_amp_layer_head(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ head_w, int k_pad_in,
                                               int x_off, int x_stride, int tile0_in, int valid_in, int part_off, int lane) {
    const int k_pad = __builtin_amdgcn_readfirstlane(k_pad_in), tile0 = __builtin_amdgcn_readfirstlane(tile0_in);
    const int valid = FULL ? NTW : __builtin_amdgcn_readfirstlane(valid_in);
    const int i = lane & 15, q = lane >> 4;
    ls_v4f acc[NTW][RH];
    float4 b[NTW];
    ls_pol_accumulate<NTW, RH, FULL>(W, bias, k_pad, x_off, x_stride, tile0, valid, lane, acc, b);
    float s[RH];
#pragma unroll
    for (int h = 0; h < RH; ++h) s[h] = 0.0f;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        if (FULL || t < valid) {
            const float4 hw = ls_pol_ld4((ls_pol_gptr)head_w + (tile0 + t) * 16 + 4 * q);
#pragma unroll
            for (int h = 0; h < RH; ++h) {
                s[h] += fmaxf(acc[t][h][0] + b[t].x, 0.0f) * hw.x;
                s[h] += fmaxf(acc[t][h][1] + b[t].y, 0.0f) * hw.y;
                s[h] += fmaxf(acc[t][h][2] + b[t].z, 0.0f) * hw.z;
                s[h] += fmaxf(acc[t][h][3] + b[t].w, 0.0f) * hw.w;
            }
        }
    }
#pragma unroll
    for (int h = 0; h < RH; ++h) {
        s[h] += __shfl_xor(s[h], 16);
        s[h] += __shfl_xor(s[h], 32);
        if (lane < 16) ls_pol_lds[part_off + i + 16 * h] = s[h];
    }
}

template <int NSPLIT>
__global__ __launch_bounds__(64 * LS_AMP_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void lsim_k_amp_step(LsAmpArgs a) {
    constexpr int ROWS = LS_AMP_ROWS, WAVES = LS_AMP_WAVES, RH = ROWS / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r0 = (long)blockIdx.x * ROWS;
    const int split = NSPLIT > 1 ? (int)blockIdx.y : 0;
    const int D = a.d.amp_dim;
    const lsim_mlp_layer& L0 = a.d.hidden[0];
    const lsim_mlp_layer& L1 = a.d.hidden[1];
    {   // ---- stage: 32 rows x 32 column slots = the block's 1024 threads
        const int r = tid >> 5, c = tid & 31;
        const long env = r0 + r;
        float xs = 0.0f, xn = 0.0f;
        if (c < D && env < a.num_envs) {
            const float p = a.prev[env * D + c], nraw = a.next[env * D + c];
            const float n = (a.dones && a.dones[env]) ? a.term[env * D + c] : nraw;             // HYBR:191-192
            if (split == 0) {
                if (a.carry_out) a.carry_out[env * D + c] = nraw;                               // HYBR:196: the next step's state is the UN-patched observation
                if (a.replay_s) {                                                               // RB:52-68: rows cursor .. cursor + N - 1 of the ring
                    const long slot = (a.replay_cursor + env) % a.replay_cap;
                    a.replay_s[slot * D + c] = p;
                    a.replay_ns[slot * D + c] = n;
                }
            }
            if (a.d.norm_mean) {                                                                // UT:124-130: float32(mean), sqrt(float32(var + eps)), clamp
                const float m = (float)a.d.norm_mean[c], sd = sqrtf((float)(a.d.norm_var[c] + a.d.norm_eps));
                const float cl = (float)a.d.norm_clip;
                xs = fminf(fmaxf((p - m) / sd, -cl), cl);
                xn = fminf(fmaxf((n - m) / sd, -cl), cl);
            } else { xs = p; xn = n; }
        }
        if (c < D) {
            ls_pol_lds[LS_AMP_O_IN + r * LS_AMP_STRIDE_IN + ls_pol_col(r, c)] = xs;
            ls_pol_lds[LS_AMP_O_IN + r * LS_AMP_STRIDE_IN + ls_pol_col(r, D + c)] = xn;
        }
        for (int cc = 2 * D + c; cc < L0.k_pad; cc += 32) ls_pol_lds[LS_AMP_O_IN + r * LS_AMP_STRIDE_IN + ls_pol_col(r, cc)] = 0.0f;
    }
    __syncthreads();
    {   // ---- trunk[0]: every wave two output tiles per pass, WAVES * 2 tiles per pass
        const int tiles = L0.n_pad >> 4;
        for (int base = 0; base < tiles; base += 2 * WAVES) {
            const int tile0 = base + 2 * wave;
            int valid = tiles - tile0;
            if (valid > 2) valid = 2;
            if (valid == 2) ls_pol_layer<2, RH, true>(L0.weight, L0.bias, L0.k_pad, LS_AMP_O_IN, LS_AMP_STRIDE_IN, LS_AMP_O_H, LS_AMP_STRIDE_H, tile0, 2, LS_POL_ACT_RELU, lane);
            else if (valid > 0) ls_pol_layer<2, RH, false>(L0.weight, L0.bias, L0.k_pad, LS_AMP_O_IN, LS_AMP_STRIDE_IN, LS_AMP_O_H, LS_AMP_STRIDE_H, tile0, valid, LS_POL_ACT_RELU, lane);
        }
    }
    __syncthreads();
    {   // ---- trunk[1] + head: this block's share of the output tiles, dealt to the waves (host: at most 2 per wave)
        const int tiles_all = L1.n_pad >> 4, share = (tiles_all + NSPLIT - 1) / NSPLIT;
        const int first = split * share;
        int mine = tiles_all - first;
        if (mine > share) mine = share;
        const int per = (share + WAVES - 1) / WAVES;          // 1 or 2
        const int tile0 = first + wave * per;
        int valid = first + mine - tile0;
        if (valid > per) valid = per;
        const int part_off = LS_AMP_O_PART + wave * ROWS;
        if (per == 1) {
            if (valid == 1) ls_amp_layer_head<1, RH, true>(L1.weight, L1.bias, a.d.head_weight, L1.k_pad, LS_AMP_O_H, LS_AMP_STRIDE_H, tile0, 1, part_off, lane);
            else if (lane < 16) { ls_pol_lds[part_off + lane] = 0.0f; ls_pol_lds[part_off + lane + 16] = 0.0f; }
        } else {
            if (valid == 2) ls_amp_layer_head<2, RH, true>(L1.weight, L1.bias, a.d.head_weight, L1.k_pad, LS_AMP_O_H, LS_AMP_STRIDE_H, tile0, 2, part_off, lane);
            else if (valid > 0) ls_amp_layer_head<2, RH, false>(L1.weight, L1.bias, a.d.head_weight, L1.k_pad, LS_AMP_O_H, LS_AMP_STRIDE_H, tile0, valid, part_off, lane);
            else if (lane < 16) { ls_pol_lds[part_off + lane] = 0.0f; ls_pol_lds[part_off + lane + 16] = 0.0f; }
        }
    }
    __syncthreads();
    if (tid >= 64) return;                                  // the finish is one wave's work (ROWS <= 64 lanes)
    float dsum = 0.0f;
    const long env = r0 + tid;
    if (tid < ROWS) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) dsum += ls_pol_lds[LS_AMP_O_PART + w * ROWS + tid];      // fixed order
    }
    if constexpr (NSPLIT > 1) {
        // The blocks of a row group exchange their partial sums through device-scope RELAXED atomics: on gfx950 those are stores / loads with
        // sc1 set -- written through to, resp. fetched from, the level all XCDs share -- so neither side needs the release / acquire fences
        // (buffer_wbl2 / buffer_inv: a write-back of every dirty line of the XCD's L2, measured at +47 us per launch at N = 4096 with
        // __threadfence()).  Order: partial sums stored and ACKNOWLEDGED (s_waitcnt vmcnt(0)), then the ticket; the block that draws the last
        // ticket reads the others' sums after it.
        if (tid < ROWS && env < a.num_envs) __hip_atomic_store(a.part + (long)split * a.num_envs + env, dsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);                      // vmcnt(0) expcnt(0) lgkmcnt(0): the stores above have completed
        __builtin_amdgcn_wave_barrier();
        unsigned int ticket = 0;
        if (tid == 0) ticket = __hip_atomic_fetch_add(a.counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
        if (ticket != (unsigned int)(NSPLIT - 1)) return;   // not the last block of this row group
        if (tid == 0) __hip_atomic_store(a.counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        if (tid < ROWS && env < a.num_envs) {
            dsum = 0.0f;
#pragma unroll
            for (int y = 0; y < NSPLIT; ++y)                // fixed order, whichever block does it
                dsum += __hip_atomic_load(a.part + (long)y * a.num_envs + env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid < ROWS && env < a.num_envs) {
        const float d = dsum + a.d.head_bias[0];
        const float e = d - 1.0f;
        float r = (float)a.d.reward_coef * fmaxf(1.0f - 0.25f * (e * e), 0.0f);                 // DISC:63
        if (a.d.task_reward_lerp > 0.0) r = (float)(1.0 - a.d.task_reward_lerp) * r + (float)a.d.task_reward_lerp * a.task_rewards[env];   // DISC:69-71
        a.rewards_out[env] = r;
        if (a.disc_out) a.disc_out[env] = d;
    }
}

static int ls_amp_check(const lsim_amp_disc* d) {
    if (!d || !d->head_weight || !d->head_bias) return LSIM_E_INVALID;
    if (d->amp_dim <= 0 || d->amp_dim > LS_AMP_MAX_DIM) return LSIM_E_UNSUPPORTED;
    if ((d->norm_mean == nullptr) != (d->norm_var == nullptr)) return LSIM_E_INVALID;
    const lsim_mlp_layer* L = d->hidden;
    for (int l = 0; l < 2; ++l) {
        if (!L[l].weight || !L[l].bias || L[l].k_pad <= 0 || L[l].n_pad <= 0 || (L[l].k_pad & 15) || (L[l].n_pad & 15)) return LSIM_E_INVALID;
        if (L[l].k_in > L[l].k_pad || L[l].n_out > L[l].n_pad) return LSIM_E_INVALID;
        if (((uintptr_t)L[l].weight & 15) || ((uintptr_t)L[l].bias & 15)) return LSIM_E_INVALID;
    }
    if (((uintptr_t)d->head_weight & 15)) return LSIM_E_INVALID;
    if (L[0].k_in != 2 * d->amp_dim || L[0].k_pad > 2 * LS_AMP_MAX_DIM || L[1].k_in != L[0].n_out || L[1].k_pad != L[0].n_pad) return LSIM_E_INVALID;
    if (L[0].n_pad > LS_AMP_MAX_H1) return LSIM_E_UNSUPPORTED;
    return LSIM_OK;
}

// blocks per row group: two while that is what fills the chip (one block per CU), one beyond; trunk[1]'s tiles per wave must stay <= 2
static int ls_amp_nsplit(const lsim_amp_disc* d, int64_t num_envs) {
    const int tiles = d->hidden[1].n_pad >> 4;
    int ns = (num_envs + LS_AMP_ROWS - 1) / LS_AMP_ROWS <= 192 ? 2 : 1;
    static const char* force = getenv("LSIM_AMP_NSPLIT");        // A/B switch, read once
    if (force && (force[0] == '1' || force[0] == '2')) ns = force[0] - '0';
    if ((tiles + ns - 1) / ns > 2 * LS_AMP_WAVES) ns = 2;
    return ((tiles + ns - 1) / ns > 2 * LS_AMP_WAVES) ? 0 : ns;
}

extern "C" int lsim_amp_step_workspace(int64_t num_envs, size_t* bytes) {
    if (num_envs <= 0 || !bytes) return LSIM_E_INVALID;
    const size_t groups = (size_t)((num_envs + LS_AMP_ROWS - 1) / LS_AMP_ROWS);
    *bytes = ((groups * sizeof(unsigned int) + 255) / 256) * 256 + 2 * (size_t)num_envs * sizeof(float);
    return LSIM_OK;
}

extern "C" int lsim_amp_step(const lsim_amp_disc* d, const float* amp_obs, const float* next_amp_obs, const uint8_t* dones, const float* terminal_amp_states,
                             const float* task_rewards, int64_t num_envs, float* rewards_out, float* disc_out, float* amp_obs_carry,
                             float* replay_states, float* replay_next_states, int64_t replay_capacity, int64_t replay_cursor,
                             void* workspace, size_t workspace_bytes, void* stream) {
    int rc = ls_amp_check(d);
    if (rc != LSIM_OK) return rc;
    if (!amp_obs || !next_amp_obs || !rewards_out || num_envs <= 0) return LSIM_E_INVALID;
    if (dones && !terminal_amp_states) return LSIM_E_INVALID;
    if (d->task_reward_lerp > 0.0 && !task_rewards) return LSIM_E_INVALID;
    if ((replay_states == nullptr) != (replay_next_states == nullptr)) return LSIM_E_INVALID;
    if (replay_states && (replay_capacity < num_envs || replay_cursor < 0 || replay_cursor >= replay_capacity)) return LSIM_E_INVALID;
    size_t need = 0;
    (void)lsim_amp_step_workspace(num_envs, &need);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return LSIM_E_INVALID;
    const int ns = ls_amp_nsplit(d, num_envs);
    if (ns == 0) return LSIM_E_UNSUPPORTED;
    LsAmpArgs a;
    memset(&a, 0, sizeof(a));
    a.d = *d; a.prev = amp_obs; a.next = next_amp_obs; a.dones = dones; a.term = terminal_amp_states; a.task_rewards = task_rewards;
    a.num_envs = (long)num_envs; a.rewards_out = rewards_out; a.disc_out = disc_out; a.carry_out = amp_obs_carry;
    a.replay_s = replay_states; a.replay_ns = replay_next_states; a.replay_cap = (long)replay_capacity; a.replay_cursor = (long)replay_cursor;
    const size_t groups = (size_t)((num_envs + LS_AMP_ROWS - 1) / LS_AMP_ROWS);
    a.counters = (unsigned int*)workspace;
    a.part = (float*)((char*)workspace + ((groups * sizeof(unsigned int) + 255) / 256) * 256);
    const size_t lds = (size_t)LS_AMP_LDS_FLOATS * sizeof(float);
    static size_t configured[2][64] = {{0}};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LSIM_E_HIP;
    const void* fn = ns == 2 ? (const void*)lsim_k_amp_step<2> : (const void*)lsim_k_amp_step<1>;
    if (lds > configured[ns - 1][dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return LSIM_E_HIP;
        configured[ns - 1][dev] = lds;
    }
    if (ns == 2) hipLaunchKernelGGL(lsim_k_amp_step<2>, dim3((unsigned)groups, 2), dim3(64 * LS_AMP_WAVES), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(lsim_k_amp_step<1>, dim3((unsigned)groups, 1), dim3(64 * LS_AMP_WAVES), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// =====================================================================================================================================
// Discriminator UPDATE side (HybridPPO.update, rsl_rl/algorithms/hybrid_ppo.py:236-281): the elementwise passes between the big GEMMs of the
// LSGAN loss and of the gradient penalty, each folded into the one kernel that has to touch the activation anyway.
//
// lsim_relu_head_backward -- everything between d loss / d d and the gradient of the LAST trunk layer's pre-activation, for the head
//   d = a2 . w3 + b3 on a2 = relu(z2) (DISC:27, DISC:61):
//     g2[b, n] = a2[b, n] > 0 ? gd[b] * w3[n] : 0            (outer product + threshold_backward: was a K = 1 "GEMM" of 182 us + a 93 us mask pass)
//     db2[n]   = sum_b g2[b, n]                               (was a 68 us column sum)
//     dw3[n]   = sum_b a2[b, n] * gd[b],  db3 = sum_b gd[b]   (was a 33 us GEMV + a reduction)
//   in one pass over a2 (read) and g2 (written).
// lsim_masked_colsum -- out[n] = sum_b (m[b, n] > 0 ? v[b, n] : 0): the gradient penalty's d w3 (amp.py _GradPenFn: threshold_backward + sum).
// Both: a block owns a slice of rows, thread (row lane, column quad) keeps its column sums in registers, the row lanes of a block add up
// through LDS, one partial row per block, summed over the blocks in a fixed order by lsim_k_wgrad_reduce (deterministic, no atomics).
#define LS_COLK_SLICES 1024
template <int MODE /* 0: head backward, 1: masked column sum */>
__global__ __launch_bounds__(256) void lsim_k_relu_cols(const float* __restrict__ a, long lda, const float* __restrict__ v, long ldv, const float* __restrict__ gd,
                                                       const float* __restrict__ w3, long batch, int n, long rows_per_slice, float* __restrict__ g2,
                                                       float* __restrict__ part1, float* __restrict__ part2) {
    const int quads = n >> 2, rl = 256 / quads;                 // row lanes per block (host: quads divides 256)
    const int q = (int)threadIdx.x % quads, lane_r = (int)threadIdx.x / quads;
    const long s0 = (long)blockIdx.x * rows_per_slice;
    long s1 = s0 + rows_per_slice;
    if (s1 > batch) s1 = batch;
    float4 c1 = make_float4(0, 0, 0, 0), c2 = make_float4(0, 0, 0, 0);
    float cg = 0.0f;
    float4 w = make_float4(0, 0, 0, 0);
    if (MODE == 0) w = *(const float4*)(w3 + 4 * q);
    for (long r = s0 + lane_r; r < s1; r += rl) {
        const float4 av = *(const float4*)(a + r * lda + 4 * q);
        if (MODE == 0) {
            const float g = gd[r];
            const float4 o = make_float4(av.x > 0.0f ? g * w.x : 0.0f, av.y > 0.0f ? g * w.y : 0.0f, av.z > 0.0f ? g * w.z : 0.0f, av.w > 0.0f ? g * w.w : 0.0f);
            *(float4*)(g2 + r * (long)n + 4 * q) = o;
            c1.x += o.x; c1.y += o.y; c1.z += o.z; c1.w += o.w;
            c2.x += av.x * g; c2.y += av.y * g; c2.z += av.z * g; c2.w += av.w * g;
            if (q == 0) cg += g;
        } else {
            const float4 vv = *(const float4*)(v + r * ldv + 4 * q);
            c1.x += av.x > 0.0f ? vv.x : 0.0f; c1.y += av.y > 0.0f ? vv.y : 0.0f; c1.z += av.z > 0.0f ? vv.z : 0.0f; c1.w += av.w > 0.0f ? vv.w : 0.0f;
        }
    }
    __shared__ float4 r1[256], r2[256];
    __shared__ float rg[256];
    r1[threadIdx.x] = c1;
    if (MODE == 0) { r2[threadIdx.x] = c2; rg[threadIdx.x] = cg; }
    __syncthreads();
    if (lane_r != 0) return;
    for (int l = 1; l < rl; ++l) {                               // fixed order
        const float4 o = r1[l * quads + q];
        c1.x += o.x; c1.y += o.y; c1.z += o.z; c1.w += o.w;
        if (MODE == 0) {
            const float4 p = r2[l * quads + q];
            c2.x += p.x; c2.y += p.y; c2.z += p.z; c2.w += p.w;
            if (q == 0) cg += rg[l * quads];
        }
    }
    *(float4*)(part1 + (size_t)blockIdx.x * n + 4 * q) = c1;
    if (MODE == 0) {
        float* p2 = part2 + (size_t)blockIdx.x * (n + 4);        // dw3 [n] | db3 | 3 unused
        *(float4*)(p2 + 4 * q) = c2;
        if (q == 0) *(float4*)(p2 + n) = make_float4(cg, 0.0f, 0.0f, 0.0f);
    }
}

static int ls_colk_plan(int64_t batch, int n, int* slices, long* rows) {
    if (batch <= 0 || n <= 0 || (n & 3) || n > 1024 || 256 % (n >> 2) != 0) return LSIM_E_UNSUPPORTED;
    const int rl = 256 / (n >> 2);
    long rps = (batch + LS_COLK_SLICES - 1) / LS_COLK_SLICES;
    rps = (rps + rl - 1) / rl * rl;
    if (rps < 4L * rl) rps = 4L * rl;
    *rows = rps;
    *slices = (int)((batch + rps - 1) / rps);
    return LSIM_OK;
}
extern "C" int lsim_relu_cols_workspace(int64_t batch, int n, size_t* bytes) {
    int slices; long rows;
    if (!bytes) return LSIM_E_INVALID;
    int rc = ls_colk_plan(batch, n, &slices, &rows);
    if (rc != LSIM_OK) return rc;
    *bytes = (size_t)slices * (2 * (size_t)n + 4) * sizeof(float);
    return LSIM_OK;
}
extern "C" int lsim_relu_head_backward(const float* relu_out, int64_t ld, const float* grad_d, const float* head_weight, int64_t batch, int n,
                                       float* grad_pre, float* grad_bias, float* grad_head /* [n + 4]: d head_weight | d head_bias | 3 zeros */,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!relu_out || !grad_d || !head_weight || !grad_pre || !grad_bias || !grad_head || !workspace || ld < n) return LSIM_E_INVALID;
    int slices; long rows; size_t need;
    int rc = ls_colk_plan(batch, n, &slices, &rows);
    if (rc != LSIM_OK) return rc;
    if ((ld & 3) || ((uintptr_t)relu_out & 15) || ((uintptr_t)grad_pre & 15) || ((uintptr_t)head_weight & 15)) return LSIM_E_UNSUPPORTED;
    (void)lsim_relu_cols_workspace(batch, n, &need);
    if (workspace_bytes < need || ((uintptr_t)workspace & 15)) return LSIM_E_INVALID;
    float* p1 = (float*)workspace;
    float* p2 = p1 + (size_t)slices * n;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lsim_k_relu_cols<0>, dim3(slices), dim3(256), 0, s, relu_out, (long)ld, (const float*)nullptr, 0L, grad_d, head_weight, (long)batch, n, rows,
                       grad_pre, p1, p2);
    // the two partial arrays have different row lengths (n and n + 4): two small fixed-order sums
    hipLaunchKernelGGL(lsim_k_wgrad_reduce, dim3(ls_reduce_blocks(p1, n)), dim3(256), 0, s, (const float*)p1, slices, n, grad_bias, (const float*)nullptr, 0, (float*)nullptr);
    hipLaunchKernelGGL(lsim_k_wgrad_reduce, dim3(ls_reduce_blocks(p2, n + 4)), dim3(256), 0, s, (const float*)p2, slices, n + 4, grad_head, (const float*)nullptr, 0, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}
extern "C" int lsim_masked_colsum(const float* v, int64_t ldv, const float* mask_src, int64_t ldm, int64_t batch, int n, float* out,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!v || !mask_src || !out || !workspace || ldv < n || ldm < n) return LSIM_E_INVALID;
    int slices; long rows; size_t need;
    int rc = ls_colk_plan(batch, n, &slices, &rows);
    if (rc != LSIM_OK) return rc;
    if ((ldv & 3) || (ldm & 3) || ((uintptr_t)v & 15) || ((uintptr_t)mask_src & 15)) return LSIM_E_UNSUPPORTED;
    (void)lsim_relu_cols_workspace(batch, n, &need);
    if (workspace_bytes < need || ((uintptr_t)workspace & 15)) return LSIM_E_INVALID;
    float* p1 = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lsim_k_relu_cols<1>, dim3(slices), dim3(256), 0, s, mask_src, (long)ldm, v, (long)ldv, (const float*)nullptr, (const float*)nullptr, (long)batch, n,
                       rows, (float*)nullptr, p1, (float*)nullptr);
    hipLaunchKernelGGL(lsim_k_wgrad_reduce, dim3(ls_reduce_blocks(p1, n)), dim3(256), 0, s, (const float*)p1, slices, n, out, (const float*)nullptr, 0, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// lsim_running_moments_update -- Normalizer.update (rsl_rl/utils/utils.py:86-106, called twice per minibatch at HYBP:279-281): batch mean /
// variance of x [batch, dim] merged into the float64 running (mean, var, count) with the parallel-variance formula, in two launches and
// without a host round trip (the torch statement was ~25 launches of 1-2 us and one pageable host -> device copy, i.e. a pipeline drain, per call).
// Batch sums are formed in float64 (the reference forms them in float32 with numpy's pairwise summation: agreement to ~1e-7 relative).
#define LS_MOM_SLICES 256
#define LS_MOM_MAX_DIM 64
__global__ __launch_bounds__(256) void lsim_k_moments_partial(const float* __restrict__ x, long ldx, long batch, int dim, long rows_per_slice, double* __restrict__ part) {
    const int c = (int)threadIdx.x & 63, lane_r = (int)threadIdx.x >> 6;           // 4 row lanes x 64 columns
    const long s0 = (long)blockIdx.x * rows_per_slice;
    long s1 = s0 + rows_per_slice;
    if (s1 > batch) s1 = batch;
    double a = 0.0, b = 0.0;
    if (c < dim)
        for (long r = s0 + lane_r; r < s1; r += 4) { const double t = (double)x[r * ldx + c]; a += t; b += t * t; }
    __shared__ double sa[256], sb[256];
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    if (lane_r == 0 && c < dim) {
        for (int l = 1; l < 4; ++l) { a += sa[64 * l + c]; b += sb[64 * l + c]; }
        part[((size_t)blockIdx.x * 2) * LS_MOM_MAX_DIM + c] = a;
        part[((size_t)blockIdx.x * 2 + 1) * LS_MOM_MAX_DIM + c] = b;
    }
}
__global__ __launch_bounds__(256) void lsim_k_moments_merge(const double* __restrict__ part, int slices, long batch, int dim, double* __restrict__ mean,
                                                           double* __restrict__ var, double* __restrict__ count) {
    const int c = (int)threadIdx.x & 63, grp = (int)threadIdx.x >> 6;             // four groups take every fourth slice, then add up in group order
    double a = 0.0, b = 0.0;
    if (c < dim)
        for (int s = grp; s < slices; s += 4) { a += part[((size_t)s * 2) * LS_MOM_MAX_DIM + c]; b += part[((size_t)s * 2 + 1) * LS_MOM_MAX_DIM + c]; }
    __shared__ double sa[256], sb[256];
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    if (grp != 0) return;
    const double cnt = count[0], n = (double)batch, tot = cnt + n;
    if (c < dim) {
        for (int l = 1; l < 4; ++l) { a += sa[64 * l + c]; b += sb[64 * l + c]; }
        const double bmean = a / n, bvar = b / n - bmean * bmean;
        const double delta = bmean - mean[c];                                                           // UT:96-106
        const double m2 = var[c] * cnt + bvar * n + delta * delta * cnt * n / tot;
        mean[c] += delta * n / tot;
        var[c] = m2 / tot;
    }
    __builtin_amdgcn_wave_barrier();
    if (c == 0) count[0] = tot;          // after this wave's reads of count[0] above (one wave: program order)
}
extern "C" int lsim_running_moments_workspace(size_t* bytes) {
    if (!bytes) return LSIM_E_INVALID;
    *bytes = (size_t)LS_MOM_SLICES * 2 * LS_MOM_MAX_DIM * sizeof(double);
    return LSIM_OK;
}
extern "C" int lsim_running_moments_update(const float* x, int64_t ldx, int64_t batch, int dim, double* mean, double* var, double* count,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !mean || !var || !count || !workspace || batch <= 0 || dim <= 0 || ldx < dim) return LSIM_E_INVALID;
    if (dim > LS_MOM_MAX_DIM) return LSIM_E_UNSUPPORTED;
    if (workspace_bytes < (size_t)LS_MOM_SLICES * 2 * LS_MOM_MAX_DIM * sizeof(double) || ((uintptr_t)workspace & 7)) return LSIM_E_INVALID;
    long rps = ((long)batch + LS_MOM_SLICES - 1) / LS_MOM_SLICES;
    rps = (rps + 3) & ~3L;
    if (rps < 16) rps = 16;
    const int slices = (int)(((long)batch + rps - 1) / rps);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lsim_k_moments_partial, dim3(slices), dim3(256), 0, s, x, (long)ldx, (long)batch, dim, rps, (double*)workspace);
    hipLaunchKernelGGL(lsim_k_moments_merge, dim3(1), dim3(256), 0, s, (const double*)workspace, slices, (long)batch, dim, mean, var, count);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// lsim_amp_pair_rows -- the discriminator's input rows of one sampled block (HYBP:247-251 normalize_torch on state and next state, DISC:57 / DISC:37
// torch.cat([state, next_state], dim=-1)): out[b, 0:D] = f(s[b]), out[b, D:2D] = f(ns[b]) with f = clamp((x - float32(mean)) / sqrt(float32(var + eps)),
// +-clip) (UT:124-130; the arithmetic of lsim_k_amp_step's stage) or the identity when mean == NULL (the gradient penalty's un-normalised pair).
// One pass instead of 14 elementwise launches and a cat per pair; the caller points `out` at this block's rows of the stacked evaluation.
__global__ __launch_bounds__(256) void lsim_k_amp_pair_rows(const float* __restrict__ s, long lds, const float* __restrict__ ns, long ldns,
                                                           const double* __restrict__ mean, const double* __restrict__ var, double eps, double clip,
                                                           long batch, int D, float* __restrict__ out, long ldo) {
    const long total = batch * 2 * D;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / (2 * D);
        const int c2 = (int)(e - b * 2 * D), c = c2 < D ? c2 : c2 - D;
        float x = c2 < D ? s[b * lds + c] : ns[b * ldns + c];
        if (mean) {
            const float m = (float)mean[c], sd = sqrtf((float)(var[c] + eps)), cl = (float)clip;
            x = fminf(fmaxf((x - m) / sd, -cl), cl);
        }
        out[b * ldo + c2] = x;
    }
}
extern "C" int lsim_amp_pair_rows(const float* states, int64_t ld_states, const float* next_states, int64_t ld_next, const double* norm_mean,
                                  const double* norm_var, double norm_eps, double norm_clip, int64_t batch, int dim, float* out, int64_t ld_out, void* stream) {
    if (!states || !next_states || !out || batch <= 0 || dim <= 0 || ld_states < dim || ld_next < dim || ld_out < 2 * dim) return LSIM_E_INVALID;
    if ((norm_mean == nullptr) != (norm_var == nullptr)) return LSIM_E_INVALID;
    const long total = (long)batch * 2 * dim;
    long blocks = (total + 4 * 256 - 1) / (4 * 256);        // ~4 elements per thread
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(lsim_k_amp_pair_rows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, states, (long)ld_states, next_states, (long)ld_next,
                       norm_mean, norm_var, norm_eps, norm_clip, (long)batch, dim, out, (long)ld_out);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}
