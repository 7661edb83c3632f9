// ls_rollout.h -- fused elementwise / storage kernels of the rollout step (include/lsim.h, "rollout-side fused kernels").
// One wavefront per env: lanes stream the env's observation rows into the storage (float2), lanes < A sample the action.
#pragma once
#include "ls_math.h"

struct LsRolloutActArgs {
    lsim_rollout_storage st;
    const int64_t* step_idx; const int64_t* draw_counter;      // device-side counters, or NULL: the values below (lsim_rollout_act_at)
    int64_t step_val, draw_val;
    const float* mean; const float* std; const float* values; const float* obs; const float* priv;
    uint32_t seed, rank;
    float* actions_out;
};
struct LsRolloutPostArgs {
    lsim_rollout_storage st;
    int64_t* step_idx; int64_t* draw_counter;                  // device-side counters, or NULL: step_val (lsim_rollout_post_at)
    int64_t step_val;
    const uint8_t* dones; const uint8_t* time_outs; const float* rewards; const float* values;
    const float* priv; const float* term_priv;
    float gamma;
};

__device__ __forceinline__ void ls_copy_row2(float* dst, const float* src, int n, int lane) {   // n even, rows 8-byte aligned
    const float2* s2 = (const float2*)src;
    float2* d2 = (float2*)dst;
    for (int k = lane; k < (n >> 1); k += 64) d2[k] = s2[k];
}

// action j of env: mean + std * z, z ~ N(0,1) by Box-Muller on the Philox block (env, draw counter, LSIM_RNG_POLICY, j / 2)
__device__ __forceinline__ float ls_sample_action(uint32_t seed, uint32_t rank, uint32_t env, uint32_t draw, int j, float mu, float sd) {
    float u[4];
    ls_u01x4(seed, rank, env, draw, LSIM_RNG_POLICY, (uint32_t)(j >> 1), u);
    const float u1 = 1.0f - u[2 * (j & 1)], u2 = u[2 * (j & 1) + 1];   // u1 in (0, 1]
    const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    return fmaf(sd, z, mu);
}
__device__ __forceinline__ float ls_normal_log_prob(float d, float sd) {      // torch.distributions.Normal.log_prob of mean + d
    return -(d * d) / (2.0f * sd * sd) - logf(sd) - 0.9189385332046727f;
}

__global__ __launch_bounds__(256) void lsim_k_rollout_act(LsRolloutActArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = blockIdx.x * 4 + (threadIdx.x >> 6);
    const lsim_rollout_storage& st = a.st;
    if (env >= st.num_envs) return;
    const int64_t t = a.step_idx ? *a.step_idx : a.step_val;
    if (t < 0 || t >= st.num_steps) return;                     // a full storage is the caller's error (HST:93-94 raises)
    const size_t row = (size_t)t * st.num_envs + env;
    ls_copy_row2(st.observations + row * st.num_obs, a.obs + (size_t)env * st.num_obs, st.num_obs, lane);
    ls_copy_row2(st.privileged_observations + row * st.num_priv_obs, a.priv + (size_t)env * st.num_priv_obs, st.num_priv_obs, lane);
    const int A = st.num_actions;
    float lp = 0.0f;
    if (lane < A) {
        const float mu = a.mean[(size_t)env * A + lane], sd = a.std[lane];
        const float act = ls_sample_action(a.seed, a.rank, (uint32_t)env, (uint32_t)(a.draw_counter ? *a.draw_counter : a.draw_val), lane, mu, sd);
        a.actions_out[(size_t)env * A + lane] = act;
        st.actions[row * A + lane] = act;
        st.mu[row * A + lane] = mu;
        st.sigma[row * A + lane] = sd;
        const float d = act - mu;
        lp = ls_normal_log_prob(d, sd);
    }
    for (int off = 16; off > 0; off >>= 1) lp += __shfl_down(lp, off, 64);       // A <= 32
    if (lane == 0) {
        st.actions_log_prob[row] = lp;
        st.values[row] = a.values[env];
    }
}

__global__ __launch_bounds__(256) void lsim_k_rollout_post(LsRolloutPostArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = blockIdx.x * 4 + (threadIdx.x >> 6);
    const lsim_rollout_storage& st = a.st;
    if (env >= st.num_envs) return;
    const int64_t t = a.step_idx ? *a.step_idx : a.step_val;
    if (t < 0 || t >= st.num_steps) return;
    const size_t row = (size_t)t * st.num_envs + env;
    const uint8_t done = a.dones[env];
    const float* src = (done ? a.term_priv : a.priv) + (size_t)env * st.num_priv_obs;
    ls_copy_row2(st.next_privileged_observations + row * st.num_priv_obs, src, st.num_priv_obs, lane);
    if (lane == 0) {
        float r = a.rewards[env];
        if (a.time_outs) r += a.gamma * (a.values[env] * (float)a.time_outs[env]);   // HIMP:110-111
        st.rewards[row] = r;
        st.dones[row] = done;
    }
}

__global__ void lsim_k_rollout_advance(int64_t* step_idx, int64_t* draw_counter) {
    *step_idx += 1;
    *draw_counter += 1;
}

static int ls_rollout_check(const lsim_rollout_storage* st) {
    if (!st || st->num_envs <= 0 || st->num_steps <= 0 || st->num_actions <= 0 || st->num_actions > 32) return LSIM_E_INVALID;
    if ((st->num_obs & 1) || (st->num_priv_obs & 1)) return LSIM_E_UNSUPPORTED;
    return LSIM_OK;
}

extern "C" int lsim_rollout_act(const lsim_rollout_storage* st, const int64_t* step_idx_dev, const int64_t* draw_counter_dev,
                                const float* mean, const float* std, const float* values, const float* obs, const float* priv_obs,
                                uint32_t seed, uint32_t rank, float* actions_out, void* stream) {
    int rc = ls_rollout_check(st);
    if (rc != LSIM_OK) return rc;
    if (!step_idx_dev || !draw_counter_dev || !mean || !std || !values || !obs || !priv_obs || !actions_out) return LSIM_E_INVALID;
    LsRolloutActArgs a;
    a.st = *st; a.step_idx = step_idx_dev; a.draw_counter = draw_counter_dev; a.step_val = 0; a.draw_val = 0; a.mean = mean; a.std = std; a.values = values;
    a.obs = obs; a.priv = priv_obs; a.seed = seed; a.rank = rank; a.actions_out = actions_out;
    hipLaunchKernelGGL(lsim_k_rollout_act, dim3((st->num_envs + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// the same two kernels with the storage row and the sampler's counter passed by value: a host-driven rollout loop knows both, and the
// one-thread kernel that advances the device-side counters (a whole launch slot per step) disappears
extern "C" int lsim_rollout_act_at(const lsim_rollout_storage* st, int64_t step_idx, int64_t draw_counter,
                                   const float* mean, const float* std, const float* values, const float* obs, const float* priv_obs,
                                   uint32_t seed, uint32_t rank, float* actions_out, void* stream) {
    int rc = ls_rollout_check(st);
    if (rc != LSIM_OK) return rc;
    if (!mean || !std || !values || !obs || !priv_obs || !actions_out || step_idx < 0 || step_idx >= st->num_steps) return LSIM_E_INVALID;
    LsRolloutActArgs a;
    a.st = *st; a.step_idx = nullptr; a.draw_counter = nullptr; a.step_val = step_idx; a.draw_val = draw_counter; a.mean = mean; a.std = std; a.values = values;
    a.obs = obs; a.priv = priv_obs; a.seed = seed; a.rank = rank; a.actions_out = actions_out;
    hipLaunchKernelGGL(lsim_k_rollout_act, dim3((st->num_envs + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

extern "C" int lsim_rollout_post_at(const lsim_rollout_storage* st, int64_t step_idx, const uint8_t* dones, const uint8_t* time_outs,
                                    const float* rewards, const float* values, const float* priv_obs, const float* term_priv_obs, float gamma,
                                    void* stream) {
    int rc = ls_rollout_check(st);
    if (rc != LSIM_OK) return rc;
    if (!dones || !rewards || !values || !priv_obs || !term_priv_obs || step_idx < 0 || step_idx >= st->num_steps) return LSIM_E_INVALID;
    LsRolloutPostArgs a;
    a.st = *st; a.step_idx = nullptr; a.draw_counter = nullptr; a.step_val = step_idx; a.dones = dones; a.time_outs = time_outs;
    a.rewards = rewards; a.values = values; a.priv = priv_obs; a.term_priv = term_priv_obs; a.gamma = gamma;
    hipLaunchKernelGGL(lsim_k_rollout_post, dim3((st->num_envs + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

extern "C" int lsim_rollout_post(const lsim_rollout_storage* st, int64_t* step_idx_dev, int64_t* draw_counter_dev,
                                 const uint8_t* dones, const uint8_t* time_outs, const float* rewards, const float* values,
                                 const float* priv_obs, const float* term_priv_obs, float gamma, void* stream) {
    int rc = ls_rollout_check(st);
    if (rc != LSIM_OK) return rc;
    if (!step_idx_dev || !draw_counter_dev || !dones || !rewards || !values || !priv_obs || !term_priv_obs) return LSIM_E_INVALID;
    LsRolloutPostArgs a;
    a.st = *st; a.step_idx = step_idx_dev; a.draw_counter = draw_counter_dev; a.step_val = 0; a.dones = dones; a.time_outs = time_outs;
    a.rewards = rewards; a.values = values; a.priv = priv_obs; a.term_priv = term_priv_obs; a.gamma = gamma;
    hipLaunchKernelGGL(lsim_k_rollout_post, dim3((st->num_envs + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(lsim_k_rollout_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, step_idx_dev, draw_counter_dev);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

__global__ __launch_bounds__(256) void lsim_k_rollout_gae(lsim_rollout_storage st, const float* last_values, float gamma, float lam,
                                                           float* returns, float* advantages) {
    const int env = blockIdx.x * 256 + threadIdx.x;
    if (env >= st.num_envs) return;
    float nxt = last_values[env], adv = 0.0f;
    for (int t = st.num_steps - 1; t >= 0; --t) {
        const size_t row = (size_t)t * st.num_envs + env;
        const float nd = 1.0f - (float)st.dones[row], v = st.values[row];
        const float delta = st.rewards[row] + nd * gamma * nxt - v;
        adv = delta + nd * gamma * lam * adv;
        returns[row] = adv + v;
        advantages[row] = (adv + v) - v;      // HST:126: returns - values, as the reference forms it
        nxt = v;
    }
}

extern "C" int lsim_rollout_gae(const lsim_rollout_storage* st, const float* last_values, float gamma, float lam,
                                float* returns, float* advantages, void* stream) {
    int rc = ls_rollout_check(st);
    if (rc != LSIM_OK) return rc;
    if (!last_values || !returns || !advantages) return LSIM_E_INVALID;
    hipLaunchKernelGGL(lsim_k_rollout_gae, dim3((st->num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, *st, last_values, gamma, lam,
                       returns, advantages);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}
