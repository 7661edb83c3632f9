// ls_policy.h -- fused rollout-time forward of the HIM policy (include/lsim.h, lsim_policy_forward): estimator encoder
// (history -> velocity + latent), L2-normalise, actor, critic -- eleven Linear layers, eight ELUs and the glue between them in ONE
// kernel instead of 29 launches (HAC:136-163, HES:64-68).  A workgroup owns 32 environments (16 for small batches); activations never leave LDS.
//
// Layer = Y[16 x N] = act(X[16 x K] W^T + b) on v_mfma_f32_16x16x4_f32 with A = W tile (16 outputs x 4 k) read straight from
// global memory (the weights, 2.2 MB, stay L2 resident; stored as 16 x 16 blocks) and B = X (4 k x 16 rows) from LDS.  Both operands are fetched as 16-byte
// vectors: lane (i, q) takes k = 16 kc + 4 q .. + 3 of its weight row / activation row and feeds component c to MFMA step c, so a
// 16-wide k chunk costs one global and one LDS vector load per 4 MFMAs.  The result tile lands as D[n = 4 q + r][row = lane % 16],
// i.e. four consecutive outputs of one row per lane: bias + ELU + one ds_write_b128.  The four waves of the block split the output
// tiles of a layer.  Weights are passed padded to multiples of 16 in both dimensions (zero fill), packed by the caller.
#pragma once
#include <hip/hip_runtime.h>

#define LS_POL_MAX_HIDDEN 512
#define LS_POL_MAX_IN 272
// LDS layout of an activation buffer: row stride = 16 (mod 64) floats -- 272 (inputs, buffer B) and 528 (buffer A) -- and inside every
// 16-float chunk the four 4-float groups of row r sit at group index q ^ g[(r >> 2) & 3], g = {0, 3, 2, 1}.  A layer's B-operand read is a
// ds_read_b128 of lane (i = row, q = k group); the hardware serves it in four 16-lane groups {q = 0: rows 0-3, 12-15; q = 1: rows 4-11} etc.
// (MI355X_MICROARCH.md, LDS), 16 slots of 16 bytes per cycle.  With a stride of 4 slots, rows r, r + 4, r + 8, r + 12 share a block of 4
// slots and the XOR by g spreads them over it: every lane group reads 16 distinct slots.  [Rounds 2-4 padded rows by 4 floats (stride 516 /
// 276): slot = (row + q) mod 16, two lanes of every group on one slot -- SQ_LDS_BANK_CONFLICT = 53 % of the kernel's LDS cycles, VERDICT r4.]
#define LS_POL_STRIDE_IN 272
#define LS_POL_STRIDE_A 528
#define LS_POL_STRIDE_B 272
static_assert(LS_POL_STRIDE_IN % 64 == 16 && LS_POL_STRIDE_A % 64 == 16 && LS_POL_STRIDE_B % 64 == 16, "row strides of 4 slots (mod 16 slots)");
static_assert(LS_POL_STRIDE_IN >= LS_POL_MAX_IN && LS_POL_STRIDE_A >= LS_POL_MAX_HIDDEN && LS_POL_STRIDE_B >= LS_POL_MAX_IN, "strides hold the widest row");
__device__ __forceinline__ int ls_pol_swz(int row) { return (0x1230 >> (4 * ((row >> 2) & 3))) & 3; }       // g = {0, 3, 2, 1}
__device__ __forceinline__ int ls_pol_col(int row, int c) { return c ^ (ls_pol_swz(row) << 2); }            // column c of row `row` -> its float index in the row

extern __shared__ float ls_pol_lds[];
// The weight / bias pointers arrive inside a struct passed by value, so the compiler cannot tell their address space and emits FLAT loads --
// which count in lgkmcnt as well as vmcnt: the `s_waitcnt lgkmcnt(0)` in front of every chunk's MFMAs (for the LDS operand reads) then also
// waits for the weight vectors just issued for two chunks ahead, i.e. every chunk pays a full L2 round trip (MFMA pipe 42 % busy, round 3
// PMC).  Explicit global address space = global_load, vmcnt only.
#define LS_POL_GLOBAL __attribute__((address_space(1)))
typedef const LS_POL_GLOBAL float* ls_pol_gptr;
typedef float ls_pol_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ls_pol_ld4(ls_pol_gptr p) {
    const ls_pol_f4v v = *(const LS_POL_GLOBAL ls_pol_f4v*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

#if defined(LS_POL_FAST_ELU)      // timing probe: what the library expm1f costs in the layer epilogues
__device__ __forceinline__ float ls_elu(float x) { return x > 0.0f ? x : __builtin_amdgcn_exp2f(x * 1.44269504f) - 1.0f; }
#else
__device__ __forceinline__ float ls_elu(float x) { return x > 0.0f ? x : expm1f(x); }
#endif

// one layer for this wave's NTW output tiles [tile0, tile0 + valid) and all RH * 16 rows of the block: every weight vector fetched from
// L2 feeds RH MFMAs (one per 16-row half), so a block of 32 environments streams half the weight bytes per environment of a block of 16.
// The k loop runs over THREE register stages used round-robin (chunk j computes from stage j % 3 while the weight vectors of chunk j + 2 load
// into stage (j + 2) % 3 and the activation vectors of chunk j + 1 are read from LDS into stage (j + 1) % 3): no register is copied.
// [Rounds 3-4 rotated w0 <- w1 <- w2 at the end of every chunk; the copy of the vector requested in that same chunk put an
// `s_waitcnt vmcnt(0)` there -- the prefetch distance was one chunk's MFMAs, not two chunks, and a deeper ring (measured in round 5: depth 4
// / 5 / 6 = 70.7 / 70.7 / 73.3 us against 63.4) only added copies in front of the same wait; the LDS operand of a chunk was read and waited
// for at its top.  The ISA of the loop showed both.]
// FULL: the wave owns all NTW tiles (every layer of the shipped networks: tile counts are powers of two) -- no per-tile predicate, so the
// loop body is straight-line code and the compiler's s_waitcnt counters see every load (a load inside a branch makes it wait for all of them)
template <int NTW, int RH, bool FULL>
__device__ __forceinline__ void ls_pol_accumulate(const float* __restrict__ W, const float* __restrict__ bias, int k_pad, int x_off, int x_stride,
                                                  int tile0, int valid, int lane, ls_v4f (&acc)[NTW][RH], float4 (&b)[NTW]) {
    const int i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int h = 0; h < RH; ++h) acc[t][h] = (ls_v4f){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < NTW; ++t) b[t] = make_float4(0, 0, 0, 0);
    if (valid > 0) {
        const float* xrow = ls_pol_lds + x_off + i * x_stride + 4 * (q ^ ls_pol_swz(i));   // B operand: row i (+ 16 h: same swizzle) of the block, k group q
        // A operand: output tile0*16 + i, k group q.  The weights are stored as 16 x 16 blocks (include/lsim.h): tile T, chunk KC / 16 at
        // (T * k_pad + KC) * 16 floats, row i, k group q inside it -- a wave's request is one contiguous kilobyte
        ls_pol_gptr wrow = (ls_pol_gptr)W + (size_t)tile0 * 16 * k_pad + 16 * i + 4 * q;
        float4 wa[NTW], wb[NTW], wc[NTW], xa[RH], xb[RH], xc[RH];
        // G = guarded against the end of the row (prologue and the last <= 4 chunks), U = unguarded (steady state: every request is in range;
        // a request inside a branch would make the compiler wait for ALL outstanding loads at the next use)
#define LS_POL_LDW_U(ST, KC) do { _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                            \
            if (FULL || t < valid) ST[t] = ls_pol_ld4(wrow + (size_t)t * 16 * k_pad + 16 * (KC)); } while (0)
#define LS_POL_LDW_G(ST, KC) do { _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                            \
            ST[t] = ((FULL || t < valid) && (KC) < k_pad) ? ls_pol_ld4(wrow + (size_t)t * 16 * k_pad + 16 * (KC)) : make_float4(0, 0, 0, 0); } while (0)
#define LS_POL_LDX_U(ST, KC) do { _Pragma("unroll") for (int h = 0; h < RH; ++h) ST[h] = *(const float4*)(xrow + 16 * h * x_stride + (KC)); } while (0)
#define LS_POL_LDX_G(ST, KC) do { if ((KC) < k_pad) LS_POL_LDX_U(ST, KC); } while (0)
#define LS_POL_MMA(WS, XS) do { _Pragma("unroll") for (int t = 0; t < NTW; ++t) { if (FULL || t < valid) { _Pragma("unroll") for (int h = 0; h < RH; ++h) {   \
            acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(WS[t].x, XS[h].x, acc[t][h], 0, 0, 0);                          \
            acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(WS[t].y, XS[h].y, acc[t][h], 0, 0, 0);                          \
            acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(WS[t].z, XS[h].z, acc[t][h], 0, 0, 0);                          \
            acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(WS[t].w, XS[h].w, acc[t][h], 0, 0, 0); } } } } while (0)
        // chunk KC from (WS, XS); requests: weights of chunk KC + 32 into WL (the stage the previous chunk consumed), activations of chunk
        // KC + 16 into XL  (scheduling barriers: left alone, hipcc sinks the requests behind the MFMAs, to two instructions in front of
        // their first use)
#define LS_POL_CHUNK(V, WS, XS, WL, XL, KC) do { LS_POL_LDW_##V(WL, (KC) + 32); LS_POL_LDX_##V(XL, (KC) + 16); __builtin_amdgcn_sched_barrier(0);    \
                                                 LS_POL_MMA(WS, XS); __builtin_amdgcn_sched_barrier(0); } while (0)
        LS_POL_LDW_G(wa, 0);
        LS_POL_LDW_G(wb, 16);
        LS_POL_LDX_G(xa, 0);
        // the bias vectors now, not behind the loop (one L2 round trip less in front of the epilogue)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
            if (FULL || t < valid) b[t] = ls_pol_ld4((ls_pol_gptr)bias + (tile0 + t) * 16 + 4 * q);
        int kc = 0;
        for (; kc + 80 <= k_pad; kc += 48) {            // every request of these three chunks (up to kc + 64 .. + 79) is inside the row
            LS_POL_CHUNK(U, wa, xa, wc, xb, kc);
            LS_POL_CHUNK(U, wb, xb, wa, xc, kc + 16);
            LS_POL_CHUNK(U, wc, xc, wb, xa, kc + 32);
        }
        if (kc < k_pad) { LS_POL_CHUNK(G, wa, xa, wc, xb, kc); kc += 16; }     // one to four chunks left; the stages are aligned again
        if (kc < k_pad) { LS_POL_CHUNK(G, wb, xb, wa, xc, kc); kc += 16; }
        if (kc < k_pad) { LS_POL_CHUNK(G, wc, xc, wb, xa, kc); kc += 16; }
        if (kc < k_pad) { LS_POL_CHUNK(G, wa, xa, wc, xb, kc); }
#undef LS_POL_CHUNK
#undef LS_POL_MMA
#undef LS_POL_LDX_G
#undef LS_POL_LDX_U
#undef LS_POL_LDW_G
#undef LS_POL_LDW_U
    }
}

// activation codes of a layer's epilogue
#define LS_POL_ACT_NONE 0
#define LS_POL_ACT_ELU 1
#define LS_POL_ACT_RELU 2      /* the AMP discriminator's trunk (ls_amp.h) */
template <int NTW, int RH, bool FULL>
__device__ __noinline__ void ls_pol_layer(const float* __restrict__ W, const float* __restrict__ bias, int k_pad_in, int x_off, int x_stride,
                                          int y_off, int y_stride, int tile0_in, int valid_in, int elu_in, int lane) {
    // arguments of a non-inlined function arrive in vector registers; these are wave-uniform: scalar registers make the loop's bounds tests
    // scalar branches (s_cbranch_scc) instead of exec-mask manipulation around every guarded load
    const int k_pad = __builtin_amdgcn_readfirstlane(k_pad_in), tile0 = __builtin_amdgcn_readfirstlane(tile0_in), elu = __builtin_amdgcn_readfirstlane(elu_in);
    const int valid = FULL ? NTW : __builtin_amdgcn_readfirstlane(valid_in);
    const int i = lane & 15, q = lane >> 4;
    ls_v4f acc[NTW][RH];
    float4 b[NTW];
    ls_pol_accumulate<NTW, RH, FULL>(W, bias, k_pad, x_off, x_stride, tile0, valid, lane, acc, b);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        if (FULL || t < valid) {
            const int n_lds = (tile0 + t) * 16 + 4 * (q ^ ls_pol_swz(i));         // D[n .. n+3][row i + 16 h], n = (tile0 + t) * 16 + 4 q: where that group of four lives in the row
#pragma unroll
            for (int h = 0; h < RH; ++h) {
                float4 y = make_float4(acc[t][h][0] + b[t].x, acc[t][h][1] + b[t].y, acc[t][h][2] + b[t].z, acc[t][h][3] + b[t].w);
                if (elu == LS_POL_ACT_ELU) { y.x = ls_elu(y.x); y.y = ls_elu(y.y); y.z = ls_elu(y.z); y.w = ls_elu(y.w); }
                else if (elu == LS_POL_ACT_RELU) { y.x = fmaxf(y.x, 0.0f); y.y = fmaxf(y.y, 0.0f); y.z = fmaxf(y.z, 0.0f); y.w = fmaxf(y.w, 0.0f); }
                *(float4*)(ls_pol_lds + y_off + (i + 16 * h) * y_stride + n_lds) = y;
            }
        }
    }
}

// this wave's output tiles of a layer: [tile0, tile0 + valid), at most `per` of them
template <int WAVES>
__device__ __forceinline__ void ls_pol_split(const lsim_mlp_layer& L, int wave, int& per, int& tile0, int& valid) {
    const int tiles = L.n_pad >> 4;
    per = (tiles + WAVES - 1) / WAVES;
    tile0 = wave * per;
    valid = tiles - tile0;
    if (valid > per) valid = per;
    if (valid < 0) valid = 0;
}

template <int ROWS, int WAVES>
__device__ __forceinline__ void ls_pol_run_layer(const lsim_mlp_layer& L, int x_off, int x_stride, int y_off, int y_stride, int elu, int wave, int lane) {
    constexpr int RH = ROWS / 16;
    int per, tile0, valid;
    ls_pol_split<WAVES>(L, wave, per, tile0, valid);
#define LS_POL_CALL(NTW) do { if (valid == NTW) ls_pol_layer<NTW, RH, true>(L.weight, L.bias, L.k_pad, x_off, x_stride, y_off, y_stride, tile0, valid, elu, lane); \
                              else if (valid > 0) ls_pol_layer<NTW, RH, false>(L.weight, L.bias, L.k_pad, x_off, x_stride, y_off, y_stride, tile0, valid, elu, lane); } while (0)
    if constexpr (32 / WAVES > 2) { if (per > 2) { LS_POL_CALL(4); __syncthreads(); return; } }
    if (per > 1) LS_POL_CALL(2);
    else LS_POL_CALL(1);
#undef LS_POL_CALL
    __syncthreads();
}

// ROWS environments per workgroup of WAVES waves: (16, 8) -- two blocks per CU, 68 KB of LDS each -- or (32, 16): one block per CU (137 KB),
// the same four waves per SIMD, half the weight traffic per environment
// ACT: the rollout's sampling / storage step (lsim_rollout_act: HIMP:90-103, HST:92-106) done by the same blocks -- the observation rows
// go to the storage while they are staged, the actor block samples the actions from the means it holds in LDS, the critic block stores the
// values: one launch less per rollout step, same Philox draws and the same summation order of the log-probability as lsim_k_rollout_act
struct LsPolActArgs {
    lsim_rollout_storage st;
    int64_t step, draw;
    const float* std;
    uint32_t seed, rank;
    float* actions_out;
    // the PREVIOUS step's post-step store (lsim_rollout_post: HIMP:105-118, HST:92-106), done by the critic blocks while they stage the new
    // privileged observation -- which IS the previous step's next critic observation -- and before they overwrite values_out; prev_step < 0: none
    int64_t prev_step;
    const uint8_t* prev_dones; const uint8_t* prev_time_outs;
    const float* prev_rewards; const float* prev_term_priv;
    float gamma;
};
template <int ROWS, int WAVES, bool ACT>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void lsim_k_policy_forward(lsim_him_policy p, const float* __restrict__ obs, const float* __restrict__ priv,
                                                             long num_envs, float* __restrict__ mean_out, float* __restrict__ values_out, LsPolActArgs act) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r0 = (long)blockIdx.x * ROWS;
    // blockIdx.y = 0: estimator encoder + actor on the observation history; 1: critic on the privileged observation.  The two halves are
    // independent: separate blocks halve the serial layer chain.
    // LDS map (floats): input rows (stride 272) | buffer A (<= 512 wide, stride 528) | buffer B (<= 272 wide, stride 272); columns swizzled: ls_pol_col
    const bool critic = blockIdx.y != 0;
#if defined(LS_POL_EXP) && LS_POL_EXP == 1      // timing probes (tools/archive/policy_ab.sh): one half of the blocks returns at once
    if (critic) return;
#elif defined(LS_POL_EXP) && LS_POL_EXP == 2
    if (!critic) return;
#endif
    const lsim_mlp_layer& first = critic ? p.critic[0] : p.encoder[0];
    const int n_in = critic ? p.num_priv_obs : p.num_obs;
    const float* __restrict__ src = critic ? priv : obs;
    constexpr int s_in = LS_POL_STRIDE_IN, s_a = LS_POL_STRIDE_A, s_b = LS_POL_STRIDE_B;
    constexpr int o_in = 0, o_a = o_in + ROWS * s_in, o_b = o_a + ROWS * s_a;
    // rows of the block are dealt to the waves (ROWS / WAVES each), a row's columns to the lanes: no index division, coalesced rows
    constexpr int RPW = ROWS / WAVES, CPL = (LS_POL_MAX_IN + 63) / 64;
    static_assert(ROWS % WAVES == 0, "whole rows per wave");
    const int kp = first.k_pad;
    {   // stage the block's input rows: every thread's loads first, all in flight together (one memory round trip), then the LDS writes
        // (a load-store-load-store loop paid a round trip per pass: 9 passes)
        float v[RPW][CPL];
#pragma unroll
        for (int a = 0; a < RPW; ++a) {
            const int r = wave * RPW + a;
            const long env = r0 + r;
#pragma unroll
            for (int m = 0; m < CPL; ++m) {
                const int c = lane + 64 * m;
                v[a][m] = (env < num_envs && c < n_in) ? src[env * n_in + c] : 0.0f;
            }
        }
#pragma unroll
        for (int a = 0; a < RPW; ++a) {
            const int r = wave * RPW + a;
#pragma unroll
            for (int m = 0; m < CPL; ++m) {
                const int c = lane + 64 * m;
                if (c < kp) ls_pol_lds[o_in + r * s_in + ls_pol_col(r, c)] = v[a][m];
            }
        }
    }
    // The rollout's copies of the staged rows (storage row `step` of the observations / privileged observations; the previous step's next
    // critic observation with the termination rows patched in, HIMR:119-121) are written at the END of the block, from the input rows that
    // stay in LDS: issued here they sat in front of the staging barrier, whose s_waitcnt vmcnt(0) waits for stores too -- every block paid
    // the store round trip before its first layer
    auto store_rows = [&]() {
        if constexpr (ACT) {
            const bool patch = critic && act.prev_step >= 0;
            float* dst = critic ? act.st.privileged_observations : act.st.observations;
#pragma unroll
            for (int a = 0; a < RPW; ++a) {
                const int r = wave * RPW + a;
                const long env = r0 + r;
                if (env >= num_envs) continue;
                const bool term = patch && act.prev_dones[env];                 // wave-uniform: one row per wave at a time
                float* d0 = dst + ((size_t)act.step * act.st.num_envs + env) * n_in;
                float* d1 = patch ? act.st.next_privileged_observations + ((size_t)act.prev_step * act.st.num_envs + env) * n_in : nullptr;
#pragma unroll
                for (int m = 0; m < CPL; ++m) {
                    const int c = lane + 64 * m;
                    if (c >= n_in) continue;
                    const float v = ls_pol_lds[o_in + r * s_in + ls_pol_col(r, c)];
                    d0[c] = v;
                    if (patch) d1[c] = term ? act.prev_term_priv[env * n_in + c] : v;
                }
            }
        }
    };
    if (ACT && critic && act.prev_step >= 0 && tid < ROWS && r0 + tid < num_envs) {
        const long env = r0 + tid;
        const size_t row = (size_t)act.prev_step * act.st.num_envs + env;
        float r = act.prev_rewards[env];
        if (act.prev_time_outs) r += act.gamma * (values_out[env] * (float)act.prev_time_outs[env]);      // HIMP:110-111, with the previous step's value
        act.st.rewards[row] = r;
        act.st.dones[row] = act.prev_dones[env];
    }
    __syncthreads();
#if defined(LS_POL_STOP_AFTER)      // timing probe (tools/archive/gpu_r5_f.sh): every block returns after that many layers (0: after the staging)
    int ls_layers_left = LS_POL_STOP_AFTER;
#define LS_RUN(L, XO, XS, YO, YS, ELU) do { if (ls_layers_left-- <= 0) return; ls_pol_run_layer<ROWS, WAVES>(L, XO, XS, YO, YS, ELU, wave, lane); } while (0)
#else
#define LS_RUN(L, XO, XS, YO, YS, ELU) ls_pol_run_layer<ROWS, WAVES>(L, XO, XS, YO, YS, ELU, wave, lane)
#endif
    if (critic) {
        // ---- critic (HAC:82-95)
        LS_RUN(p.critic[0], o_in, s_in, o_a, s_a, 1);
        LS_RUN(p.critic[1], o_a, s_a, o_b, s_b, 1);
        LS_RUN(p.critic[2], o_b, s_b, o_a, s_a, 1);
        LS_RUN(p.critic[3], o_a, s_a, o_b, s_b, 0);
        if (tid < ROWS && r0 + tid < num_envs) {
            const float v = ls_pol_lds[o_b + tid * s_b + ls_pol_col(tid, 0)];
            values_out[r0 + tid] = v;
            if (ACT) act.st.values[(size_t)act.step * act.st.num_envs + r0 + tid] = v;
        }
        store_rows();
        return;
    }
    // ---- estimator encoder (HES:64-68): history -> (velocity 3, latent 16)
    LS_RUN(p.encoder[0], o_in, s_in, o_a, s_a, 1);
    LS_RUN(p.encoder[1], o_a, s_a, o_b, s_b, 1);
    LS_RUN(p.encoder[2], o_b, s_b, o_a, s_a, 0);
    // ---- actor input (HAC:136-141): [current one-step observation, velocity estimate, L2-normalised latent] -> buffer B
    {
        const int n1 = p.num_one_step_obs, nl = p.encoder[2].n_out - 3, kin = p.actor[0].k_pad;
        for (int e = tid; e < ROWS * kin; e += 64 * WAVES) {
            const int r = e / kin, c = e - r * kin;
            float v = 0.0f;
            if (c < n1) v = ls_pol_lds[o_in + r * s_in + ls_pol_col(r, c)];
            else if (c < n1 + 3) v = ls_pol_lds[o_a + r * s_a + ls_pol_col(r, c - n1)];
            else if (c < n1 + 3 + nl) {
                float ss = 0.0f;
                for (int k = 0; k < nl; ++k) { const float z = ls_pol_lds[o_a + r * s_a + ls_pol_col(r, 3 + k)]; ss += z * z; }
                v = ls_pol_lds[o_a + r * s_a + ls_pol_col(r, 3 + (c - n1 - 3))] / fmaxf(sqrtf(ss), 1e-12f);      // F.normalize(p=2, eps=1e-12)
            }
            ls_pol_lds[o_b + r * s_b + ls_pol_col(r, c)] = v;
        }
        __syncthreads();
    }
    // ---- actor (HAC:66-80)
    LS_RUN(p.actor[0], o_b, s_b, o_a, s_a, 1);
    LS_RUN(p.actor[1], o_a, s_a, o_b, s_b, 1);
    LS_RUN(p.actor[2], o_b, s_b, o_a, s_a, 1);
    LS_RUN(p.actor[3], o_a, s_a, o_b, s_b, 0);
#undef LS_RUN
    for (int e = tid; e < ROWS * p.num_actions; e += 64 * WAVES) {
        const int r = e / p.num_actions, c = e - r * p.num_actions;
        if (r0 + r < num_envs) mean_out[(r0 + r) * p.num_actions + c] = ls_pol_lds[o_b + r * s_b + ls_pol_col(r, c)];
    }
    if constexpr (ACT) {
        const int A = p.num_actions;                    // <= 16 (checked by the host)
        float* lp_lds = ls_pol_lds + o_a;               // buffer A is free again: 16 log-probability terms per row
        for (int e = tid; e < ROWS * 16; e += 64 * WAVES) {
            const int r = e >> 4, c = e & 15;
            const long env = r0 + r;
            float lp = 0.0f;
            if (c < A && env < num_envs) {
                const float mu = ls_pol_lds[o_b + r * s_b + ls_pol_col(r, c)], sd = act.std[c];
                const float a = ls_sample_action(act.seed, act.rank, (uint32_t)env, (uint32_t)act.draw, c, mu, sd);
                const size_t row = (size_t)act.step * act.st.num_envs + env;
                act.actions_out[env * A + c] = a;
                act.st.actions[row * A + c] = a;
                act.st.mu[row * A + c] = mu;
                act.st.sigma[row * A + c] = sd;
                lp = ls_normal_log_prob(a - mu, sd);
            }
            lp_lds[e] = lp;
        }
        __syncthreads();
        if (tid < ROWS && r0 + tid < num_envs) {        // the pairwise tree of lsim_k_rollout_act's shuffle reduction, so that the sums agree bit for bit
            const float* l = lp_lds + 16 * tid;
            float t8[8], t4[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) t8[i] = l[i] + l[i + 8];
#pragma unroll
            for (int i = 0; i < 4; ++i) t4[i] = t8[i] + t8[i + 4];
            act.st.actions_log_prob[(size_t)act.step * act.st.num_envs + r0 + tid] = (t4[0] + t4[2]) + (t4[1] + t4[3]);
        }
    }
    store_rows();
}

static int ls_pol_check_layer(const lsim_mlp_layer* L, int k_in_expected) {
    if (!L->weight || !L->bias || L->k_pad <= 0 || L->n_pad <= 0 || (L->k_pad & 15) || (L->n_pad & 15)) return 1;
    if (L->k_in > L->k_pad || L->n_out > L->n_pad || L->k_in != k_in_expected) return 1;
    if (L->n_pad > LS_POL_MAX_HIDDEN || L->k_pad > LS_POL_MAX_HIDDEN) return 1;
    if (((uintptr_t)L->weight & 15) || ((uintptr_t)L->bias & 15)) return 1;
    return 0;
}

static int ls_policy_launch(const lsim_him_policy* p, const float* obs, const float* priv_obs, int64_t num_envs, float* mean_out,
                            float* values_out, const LsPolActArgs* act_args, void* stream) {
    if (!p || !obs || !priv_obs || !mean_out || !values_out || num_envs <= 0) return LSIM_E_INVALID;
    const bool act = act_args != nullptr;
    LsPolActArgs aa;
    memset(&aa, 0, sizeof(aa));
    if (act) aa = *act_args;
    if (p->num_obs > LS_POL_MAX_IN || p->num_priv_obs > LS_POL_MAX_IN || p->num_actions <= 0 || p->num_actions > 16) return LSIM_E_UNSUPPORTED;
    int bad = 0, k = p->num_obs;
    for (int l = 0; l < 3; ++l) { bad |= ls_pol_check_layer(&p->encoder[l], k); k = p->encoder[l].n_out; }
    if (p->encoder[2].n_out < 4 || p->encoder[0].k_pad > LS_POL_MAX_IN) bad = 1;
    k = p->num_one_step_obs + p->encoder[2].n_out;
    for (int l = 0; l < 4; ++l) { bad |= ls_pol_check_layer(&p->actor[l], k); k = p->actor[l].n_out; }
    if (p->actor[3].n_out != p->num_actions || p->actor[0].k_pad > LS_POL_MAX_IN) bad = 1;
    k = p->num_priv_obs;
    for (int l = 0; l < 4; ++l) { bad |= ls_pol_check_layer(&p->critic[l], k); k = p->critic[l].n_out; }
    if (p->critic[3].n_out != 1 || p->critic[0].k_pad > LS_POL_MAX_IN) bad = 1;
    // layers whose output lands in LDS buffer B (row stride LS_POL_STRIDE_B): second and last layer of each network
    if (p->encoder[1].n_pad > LS_POL_MAX_IN || p->actor[1].n_pad > LS_POL_MAX_IN || p->actor[3].n_pad > LS_POL_MAX_IN ||
        p->critic[1].n_pad > LS_POL_MAX_IN || p->critic[3].n_pad > LS_POL_MAX_IN) bad = 1;
    if (bad) return LSIM_E_UNSUPPORTED;
    // 32 environments per block once that still fills the chip's 256 CUs with one block each; 16 per block (two blocks per CU) below that
    static const bool force16 = getenv("LSIM_POLICY_ROWS16") != nullptr;      // A/B switch (tools/policy_time.py), read once
    const bool wide = num_envs >= 2048 && !force16;
    const int rows = wide ? 32 : 16;
    const size_t lds = (size_t)rows * (LS_POL_STRIDE_IN + LS_POL_STRIDE_A + LS_POL_STRIDE_B) * sizeof(float);
    static size_t configured[4][64] = {{0}};     // per kernel and device: the attribute belongs to the device's copy of the kernel
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LSIM_E_HIP;
    const void* fn = wide ? (act ? (const void*)lsim_k_policy_forward<32, 16, true> : (const void*)lsim_k_policy_forward<32, 16, false>)
                          : (act ? (const void*)lsim_k_policy_forward<16, 8, true> : (const void*)lsim_k_policy_forward<16, 8, false>);
    const int slot = 2 * (int)wide + (int)act;
    if (lds > configured[slot][dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return LSIM_E_HIP;
        configured[slot][dev] = lds;
    }
    const int blocks = (int)((num_envs + rows - 1) / rows);
#define LS_POL_LAUNCH(R, W, A) hipLaunchKernelGGL((lsim_k_policy_forward<R, W, A>), dim3(blocks, 2), dim3(64 * W), lds, (hipStream_t)stream, *p, obs, priv_obs, \
                                                 (long)num_envs, mean_out, values_out, aa)
    if (wide) { if (act) LS_POL_LAUNCH(32, 16, true); else LS_POL_LAUNCH(32, 16, false); }
    else { if (act) LS_POL_LAUNCH(16, 8, true); else LS_POL_LAUNCH(16, 8, false); }
#undef LS_POL_LAUNCH
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

extern "C" int lsim_policy_forward(const lsim_him_policy* p, const float* obs, const float* priv_obs, int64_t num_envs, float* mean_out,
                                   float* values_out, void* stream) {
    return ls_policy_launch(p, obs, priv_obs, num_envs, mean_out, values_out, nullptr, stream);
}

extern "C" int lsim_policy_act_at(const lsim_him_policy* p, const lsim_rollout_storage* st, int64_t step_idx, int64_t draw_counter,
                                  const float* obs, const float* priv_obs, const float* std, uint32_t seed, uint32_t rank,
                                  float* mean_out, float* values_out, float* actions_out, void* stream) {
    if (!p || !st || !std || !actions_out) return LSIM_E_INVALID;
    int rc = ls_rollout_check(st);
    if (rc != LSIM_OK) return rc;
    if (step_idx < 0 || step_idx >= st->num_steps || st->num_obs != p->num_obs || st->num_priv_obs != p->num_priv_obs ||
        st->num_actions != p->num_actions) return LSIM_E_INVALID;
    LsPolActArgs a;
    memset(&a, 0, sizeof(a));
    a.st = *st; a.step = step_idx; a.draw = draw_counter; a.std = std; a.seed = seed; a.rank = rank; a.actions_out = actions_out;
    a.prev_step = -1;
    return ls_policy_launch(p, obs, priv_obs, st->num_envs, mean_out, values_out, &a, stream);
}

extern "C" int lsim_policy_act_post_at(const lsim_him_policy* p, const lsim_rollout_storage* st, int64_t step_idx, int64_t draw_counter,
                                       const float* obs, const float* priv_obs, const float* std, uint32_t seed, uint32_t rank,
                                       float* mean_out, float* values_out, float* actions_out,
                                       int64_t prev_step, const uint8_t* prev_dones, const uint8_t* prev_time_outs, const float* prev_rewards,
                                       const float* prev_term_priv_obs, float gamma, void* stream) {
    if (!p || !st || !std || !actions_out) return LSIM_E_INVALID;
    int rc = ls_rollout_check(st);
    if (rc != LSIM_OK) return rc;
    if (step_idx < 0 || step_idx >= st->num_steps || st->num_obs != p->num_obs || st->num_priv_obs != p->num_priv_obs ||
        st->num_actions != p->num_actions) return LSIM_E_INVALID;
    if (prev_step >= st->num_steps || (prev_step >= 0 && (!prev_dones || !prev_rewards || !prev_term_priv_obs))) return LSIM_E_INVALID;
    LsPolActArgs a;
    memset(&a, 0, sizeof(a));
    a.st = *st; a.step = step_idx; a.draw = draw_counter; a.std = std; a.seed = seed; a.rank = rank; a.actions_out = actions_out;
    a.prev_step = prev_step < 0 ? -1 : prev_step; a.prev_dones = prev_dones; a.prev_time_outs = prev_time_outs; a.prev_rewards = prev_rewards;
    a.prev_term_priv = prev_term_priv_obs; a.gamma = gamma;
    return ls_policy_launch(p, obs, priv_obs, st->num_envs, mean_out, values_out, &a, stream);
}
