// ls_learn.h -- tall-skinny weight-gradient kernel for the learner's small Linear layers (include/lsim.h, lsim_linear_wgrad).
//
// dW[n, k] = sum_b g[b, n] * x[b, k],  db[n] = sum_b g[b, n]   with b = 102 400 minibatch rows and n_out * k_in <= 8192
// (actor / critic / estimator heads, the 45-wide target input layer, the 32 x 16 prototype layer).  BLAS runs these
// K = 102 400 reductions through 16x32 macro-tiles at 1-5 % of peak (48-270 us each) and torch's column-sum takes 260 us for
// 19 columns; here every wave owns a slice of the batch, keeps the whole n_out x k_in result in MFMA accumulators
// (v_mfma_f32_16x16x4_f32: A = 16 n x 4 rows of g, B = 4 rows x 16 k of x, loaded straight from global memory in the MFMA
// operand layout: 16 consecutive floats of one row per 16 lanes), writes its partial tile, and a second kernel adds the
// partials in a fixed order (deterministic, no atomics).
#pragma once
#include <hip/hip_runtime.h>

typedef float ls_v4f __attribute__((ext_vector_type(4)));

#define LS_WGRAD_WAVES_PER_BLOCK 4

template <int NT, int KT>
__global__ __launch_bounds__(64 * LS_WGRAD_WAVES_PER_BLOCK) void lsim_k_linear_wgrad(const float* __restrict__ x, long ldx, const float* __restrict__ g, long ldg,
                                                                                    long batch, int k_in, int n_out, long rows_per_wave,
                                                                                    float* __restrict__ part_dw, float* __restrict__ part_db) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * LS_WGRAD_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int sub = lane >> 4, col = lane & 15;
    const long b0 = (long)wave * rows_per_wave;
    long b1 = b0 + rows_per_wave;
    if (b1 > batch) b1 = batch;
    ls_v4f acc[NT][KT];
    float dbacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        dbacc[nt] = 0.0f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = (ls_v4f){0.0f, 0.0f, 0.0f, 0.0f};
    }
    for (long b = b0; b < b1; b += 4) {
        const long row = b + sub;
        const bool valid = row < b1;
        float a[NT], bb[KT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + col;
            a[nt] = (valid && n < n_out) ? g[row * ldg + n] : 0.0f;
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k = kt * 16 + col;
            bb[kt] = (valid && k < k_in) ? x[row * ldx + k] : 0.0f;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            dbacc[nt] += a[nt];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt], bb[kt], acc[nt][kt], 0, 0, 0);
        }
    }
    float* pw = part_dw + (size_t)wave * n_out * k_in;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nt * 16 + 4 * sub + r, k = kt * 16 + col;     // C[i = 4 * (lane / 16) + r][j = lane % 16]
                if (n < n_out && k < k_in) pw[(size_t)n * k_in + k] = acc[nt][kt][r];
            }
    if (part_db) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v = dbacc[nt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = nt * 16 + col;
            if (sub == 0 && n < n_out) part_db[(size_t)wave * n_out + n] = v;
        }
    }
}

// out[o] = sum over waves of part[w][o] in a fixed order.  Block = 16 outputs x 16 wave-slices: thread (o, s) adds the partials
// w = s, s + 16, ... with four independent accumulators (64-byte coalesced rows), then the 16 slices meet in LDS.
__global__ __launch_bounds__(256) void lsim_k_wgrad_reduce(const float* __restrict__ part, int num_waves, int count, float* __restrict__ out) {
    __shared__ float red[16][17];
    const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + ol;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (o < count) {
        int w = sl;
        for (; w + 48 < num_waves; w += 64) {
            s0 += part[(size_t)w * count + o]; s1 += part[(size_t)(w + 16) * count + o];
            s2 += part[(size_t)(w + 32) * count + o]; s3 += part[(size_t)(w + 48) * count + o];
        }
        for (; w < num_waves; w += 16) s0 += part[(size_t)w * count + o];
    }
    red[sl][ol] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && o < count) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][ol];
        out[o] = t;
    }
}

#define LS_WGRAD_MAX_TILES 32

typedef void (*ls_wgrad_fn)(const float*, long, const float*, long, long, int, int, long, float*, float*);

template <int NT, int KT> static void ls_wgrad_launch(const float* x, long ldx, const float* g, long ldg, long batch, int k_in, int n_out,
                                                      long rows_per_wave, int blocks, float* pdw, float* pdb, hipStream_t s) {
    hipLaunchKernelGGL((lsim_k_linear_wgrad<NT, KT>), dim3(blocks), dim3(64 * LS_WGRAD_WAVES_PER_BLOCK), 0, s, x, ldx, g, ldg, batch, k_in, n_out,
                       rows_per_wave, pdw, pdb);
}

template <int NT> static int ls_wgrad_dispatch_k(int kt, const float* x, long ldx, const float* g, long ldg, long batch, int k_in, int n_out,
                                                 long rpw, int blocks, float* pdw, float* pdb, hipStream_t s) {
    switch (kt) {
#define LS_K(KT) case KT: if constexpr (NT * KT <= LS_WGRAD_MAX_TILES) { ls_wgrad_launch<NT, KT>(x, ldx, g, ldg, batch, k_in, n_out, rpw, blocks, pdw, pdb, s); return 0; } return 1;
        LS_K(1) LS_K(2) LS_K(3) LS_K(4) LS_K(5) LS_K(6) LS_K(7) LS_K(8)
#undef LS_K
        default: return 1;
    }
}

extern "C" int lsim_linear_wgrad_workspace(long batch, int k_in, int n_out, size_t* bytes, int* num_waves) {
    if (!bytes || !num_waves || batch <= 0 || k_in <= 0 || n_out <= 0) return LSIM_E_INVALID;
    const int nt = (n_out + 15) / 16, kt = (k_in + 15) / 16;
    if (nt > 8 || kt > 8 || nt * kt > LS_WGRAD_MAX_TILES) return LSIM_E_UNSUPPORTED;
    long waves = 1024;                                   // one wave per SIMD of the 256-CU part
    long rpw = ((batch + waves - 1) / waves + 3) & ~3L;  // multiple of the 4 rows one MFMA consumes
    if (rpw < 16) rpw = 16;
    waves = (batch + rpw - 1) / rpw;
    waves = (waves + LS_WGRAD_WAVES_PER_BLOCK - 1) / LS_WGRAD_WAVES_PER_BLOCK * LS_WGRAD_WAVES_PER_BLOCK;
    *num_waves = (int)waves;
    *bytes = (size_t)waves * ((size_t)n_out * k_in + n_out) * sizeof(float);
    return LSIM_OK;
}

extern "C" int lsim_linear_wgrad(const float* x, int64_t ldx, const float* g, int64_t ldg, int64_t batch, int k_in, int n_out,
                                 float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !g || !dw || !workspace) return LSIM_E_INVALID;
    size_t need; int waves;
    int rc = lsim_linear_wgrad_workspace(batch, k_in, n_out, &need, &waves);
    if (rc != LSIM_OK) return rc;
    if (workspace_bytes < need || ldx < k_in || ldg < n_out) return LSIM_E_INVALID;
    const int nt = (n_out + 15) / 16, kt = (k_in + 15) / 16;
    long rpw = ((batch + 1023) / 1024 + 3) & ~3L;
    if (rpw < 16) rpw = 16;
    float* pdw = (float*)workspace;
    float* pdb = db ? pdw + (size_t)waves * n_out * k_in : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = waves / LS_WGRAD_WAVES_PER_BLOCK;
    int bad = 1;
    switch (nt) {
#define LS_N(NT) case NT: bad = ls_wgrad_dispatch_k<NT>(kt, x, ldx, g, ldg, batch, k_in, n_out, rpw, blocks, pdw, pdb, s); break;
        LS_N(1) LS_N(2) LS_N(3) LS_N(4) LS_N(5) LS_N(6) LS_N(7) LS_N(8)
#undef LS_N
        default: break;
    }
    if (bad) return LSIM_E_UNSUPPORTED;
    const int count = n_out * k_in;
    hipLaunchKernelGGL(lsim_k_wgrad_reduce, dim3((count + 15) / 16), dim3(256), 0, s, pdw, waves, count, dw);
    if (db) hipLaunchKernelGGL(lsim_k_wgrad_reduce, dim3((n_out + 15) / 16), dim3(256), 0, s, pdb, waves, n_out, db);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}
