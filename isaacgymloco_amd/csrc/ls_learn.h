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
    if (b0 >= batch) return;                      // padding waves of the last block own no partial slot
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
    long b = b0;
    {   // steady state: 16 rows (four MFMA steps) per iteration, every row inside the slice, all 4 (NT + KT) loads issued before the
        // first MFMA (one wave per SIMD: without this the loop runs at one memory latency per step).  Columns past the matrix are read
        // from a clamped column and not masked: they only reach accumulators of outputs that are never written.
        int gn[NT], xk[KT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { const int n = nt * 16 + col; gn[nt] = n < n_out ? n : n_out - 1; }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) { const int k = kt * 16 + col; xk[kt] = k < k_in ? k : k_in - 1; }
        for (; b + 16 <= b1; b += 16) {
            float a[4][NT], bb[4][KT];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const long row = b + 4 * s + sub;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) a[s][nt] = g[row * ldg + gn[nt]];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) bb[s][kt] = x[row * ldx + xk[kt]];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    dbacc[nt] += a[s][nt];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][nt], bb[s][kt], acc[nt][kt], 0, 0, 0);
                }
        }
    }
    for (; b < b1; b += 4) {
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
// VEC4: four consecutive outputs per thread through 16-byte loads (count % 4 == 0, 16-byte aligned rows): a wave reads 4 partial rows x 256
// contiguous bytes per load instead of 4 x 64 -- the summing launch of a minibatch's 15 layers read its 70 MB at 0.9 TB/s in the scalar form
// (65 us).  Every output is still added up in the same order: (s0 + s1) + (s2 + s3) over the partials w = s, s + 16, ..., then the 16 slices.
#define LS_LEARN_HD static __host__ __device__ __forceinline__
LS_LEARN_HD int ls_reduce_vec4(const float* part, int count) { return (count % 4 == 0) && ((((uintptr_t)part) & 15) == 0); }
#ifndef LS_REDUCE_QUADS
#define LS_REDUCE_QUADS 1          // output quads per thread of the vector form.  Measured in round 5 (train line, three interleaved runs): 1 -> update
#endif                             // 70.3-70.9 ms, 2 -> 71.9-74.5, 4 -> 73.5-75.0: the summing launch wants more blocks, not more loads per thread
LS_LEARN_HD int ls_reduce_blocks(const float* part, int count) { return ls_reduce_vec4(part, count) ? (count + 64 * LS_REDUCE_QUADS - 1) / (64 * LS_REDUCE_QUADS) : (count + 15) / 16; }
__device__ __forceinline__ void ls_wgrad_reduce_body(const float* __restrict__ part, int num_waves, int count, float* __restrict__ out,
                                                     const float* __restrict__ part2, int count2, float* __restrict__ out2, int block) {
    // one launch serves the weight-gradient partials (count outputs) and, in the blocks after them, the bias-gradient partials (count2)
    __shared__ float red[16][17];
    __shared__ float4 red4[16][17];
    const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int nb1 = ls_reduce_blocks(part, count);
    const bool second = block >= nb1;
    if (!second && ls_reduce_vec4(part, count)) {          // block-uniform
        // LS_REDUCE_QUADS quads of outputs per thread, 64 outputs apart; every output is added up exactly as before (the mapping of outputs to
        // threads changed in round 5, the order of additions did not)
        auto add = [](float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
        int oq[LS_REDUCE_QUADS];
        float4 s0[LS_REDUCE_QUADS], s1[LS_REDUCE_QUADS], s2[LS_REDUCE_QUADS], s3[LS_REDUCE_QUADS];
#pragma unroll
        for (int q = 0; q < LS_REDUCE_QUADS; ++q) {
            oq[q] = (block * LS_REDUCE_QUADS + q) * 64 + 4 * ol;
            s0[q] = s1[q] = s2[q] = s3[q] = make_float4(0, 0, 0, 0);
        }
        int w = sl;
        for (; w + 48 < num_waves; w += 64) {
#pragma unroll
            for (int q = 0; q < LS_REDUCE_QUADS; ++q) {
                if (oq[q] >= count) continue;
                const float* p0 = part + (size_t)w * count + oq[q];
                add(s0[q], *(const float4*)p0); add(s1[q], *(const float4*)(p0 + (size_t)16 * count));
                add(s2[q], *(const float4*)(p0 + (size_t)32 * count)); add(s3[q], *(const float4*)(p0 + (size_t)48 * count));
            }
        }
        for (; w < num_waves; w += 16) {
#pragma unroll
            for (int q = 0; q < LS_REDUCE_QUADS; ++q)
                if (oq[q] < count) add(s0[q], *(const float4*)(part + (size_t)w * count + oq[q]));
        }
#pragma unroll
        for (int q = 0; q < LS_REDUCE_QUADS; ++q) {
            if (q > 0) __syncthreads();
            red4[sl][ol] = make_float4((s0[q].x + s1[q].x) + (s2[q].x + s3[q].x), (s0[q].y + s1[q].y) + (s2[q].y + s3[q].y),
                                       (s0[q].z + s1[q].z) + (s2[q].z + s3[q].z), (s0[q].w + s1[q].w) + (s2[q].w + s3[q].w));
            __syncthreads();
            const int o = oq[q];
            if (sl == 0 && o < count) {
                float4 t = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 16; ++k) add(t, red4[k][ol]);
                if ((((uintptr_t)out) & 15) == 0) *(float4*)(out + o) = t;       // a gradient-arena slice starts wherever the previous parameter ended
                else { out[o] = t.x; out[o + 1] = t.y; out[o + 2] = t.z; out[o + 3] = t.w; }
            }
        }
        return;
    }
    if (second) { part = part2; out = out2; count = count2; }
    const int o = (block - (second ? nb1 : 0)) * 16 + ol;
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
__global__ __launch_bounds__(256) void lsim_k_wgrad_reduce(const float* __restrict__ part, int num_waves, int count, float* __restrict__ out,
                                                           const float* __restrict__ part2, int count2, float* __restrict__ out2) {
    ls_wgrad_reduce_body(part, num_waves, count, out, part2, count2, out2, (int)blockIdx.x);
}
// the same for up to LS_REDUCE_BATCH layers in one launch (lsim_wgrad_reduce_batch): block -> (layer, block of that layer) by the table
#define LS_REDUCE_BATCH 24
struct LsReduceBatch {
    lsim_wgrad_pending it[LS_REDUCE_BATCH];
    int first[LS_REDUCE_BATCH + 1];
    int n;
};
__global__ __launch_bounds__(256) void lsim_k_wgrad_reduce_batch(LsReduceBatch b) {
    int e = 0;
    while (e + 1 < b.n && (int)blockIdx.x >= b.first[e + 1]) ++e;           // wave-uniform
    const lsim_wgrad_pending& p = b.it[e];
    ls_wgrad_reduce_body(p.part, p.num_partials, p.count, p.out, p.part2, p.count2, p.out2, (int)blockIdx.x - b.first[e]);
}

// VEC: 2 = 16-byte vector load (rows 16-byte aligned), 1 = two 8-byte loads (rows 8-byte aligned, e.g. ld = 238 or 270), 0 = scalars
// FULL: the four columns are inside the matrix (interior tile): no column tests, no branches -- the row test becomes a select on a
// clamped row, so that all operand loads of a step issue back to back and overlap.
template <int VEC, bool FULL>
__device__ __forceinline__ void ls_wgrad_load4(const float* __restrict__ p, long ld, long row, bool row_ok, long safe_row, int c0, int cmax, float (&v)[4]) {
    if (FULL) {
        const float* q = p + (row_ok ? row : safe_row) * ld + c0;
        float t0, t1, t2, t3;
        if (VEC == 2) { const float4 t = *(const float4*)q; t0 = t.x; t1 = t.y; t2 = t.z; t3 = t.w; }
        else if (VEC == 1) { const float2 ta = *(const float2*)q, tb = *(const float2*)(q + 2); t0 = ta.x; t1 = ta.y; t2 = tb.x; t3 = tb.y; }
        else { t0 = q[0]; t1 = q[1]; t2 = q[2]; t3 = q[3]; }
        v[0] = row_ok ? t0 : 0.0f; v[1] = row_ok ? t1 : 0.0f; v[2] = row_ok ? t2 : 0.0f; v[3] = row_ok ? t3 : 0.0f;
        return;
    }
    // edge tile: which columns exist is a per-lane constant of the whole batch loop, so there are no branches here either: the same
    // vector loads as the interior path from a start column clamped into the row (rows are padded to the vector width by the VEC
    // contract: ld % 4 == 0 resp. ld % 2 == 0).  Components of columns >= cmax are NOT masked: they only reach accumulators of outputs
    // with n >= n_out or k >= k_in, which are never written (each MFMA output element depends on its own operand row / column only)
    const float* r = p + (row_ok ? row : safe_row) * ld;
    float t[4];
    if (VEC == 2) {
        const int last = ((cmax + 3) & ~3) - 4;
        const float4 u = *(const float4*)(r + (c0 < last ? c0 : last));
        t[0] = u.x; t[1] = u.y; t[2] = u.z; t[3] = u.w;
    } else if (VEC == 1) {
        const int last = ((cmax + 1) & ~1) - 2;
        const float2 ua = *(const float2*)(r + (c0 < last ? c0 : last)), ub = *(const float2*)(r + (c0 + 2 < last ? c0 + 2 : last));
        t[0] = ua.x; t[1] = ua.y; t[2] = ub.x; t[3] = ub.y;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = r[c0 + j < cmax ? c0 + j : cmax - 1];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = row_ok ? t[j] : 0.0f;
}

// the batch loop of one (tile, slice): operands of the current step and of the next two are in registers (prefetch distance 2:
// ~2 x 1024 MFMA cycles per wave, two waves per SIMD)
// FZ: the A operand is formed on the fly as g_z * elu'(z) from the incoming gradient g_z and the saved ELU OUTPUT z (torch's
// elu_backward with is_result: z > 0 ? 1 : z + alpha, alpha = 1), and the waves of k-block 0 also write it out (g_y, the gradient of
// the pre-activation, which the caller's input-gradient GEMM needs): the separate elu_backward pass over [B, N] disappears.
template <int VX, int VG, int KG, bool FULL, bool FZ>
__device__ __forceinline__ void ls_wgrad_tile_loop(const float* __restrict__ x, long ldx, const float* __restrict__ g, long ldg,
                                                   const float* __restrict__ z, long ldz, float* __restrict__ gy, bool write_gy, long b0, long b1,
                                                   long safe, int n_base, int k_base, int n_out, int k_in, int sub, int col,
                                                   ls_v4f (&acc)[4][4 * KG], float (&dbacc)[4], float zoff) {
    float a0[4], x0[KG][4], a1[4], x1[KG][4], a2[4], x2[KG][4], z1[4], z2[4];
    // wave-uniform, loop-invariant: does the second 64-column k group of an edge tile exist at all
    const bool live_k1 = FULL || k_base + 64 < k_in;
    if (!FULL && KG > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x1[KG - 1][j] = x2[KG - 1][j] = 0.0f;      // a dead second k group is never loaded
    }
#define LS_LOAD_STEP(ROW, A, Z, X) do { const long row_ = (ROW); const bool ok_ = row_ < b1;                                 \
        ls_wgrad_load4<VG, FULL>(g, ldg, row_, ok_, safe, n_base + 4 * col, n_out, A);                                 \
        if (FZ) ls_wgrad_load4<VG, FULL>(z, ldz, row_, ok_, safe, n_base + 4 * col, n_out, Z);                         \
        ls_wgrad_load4<VX, FULL>(x, ldx, row_, ok_, safe, k_base + 4 * col, k_in, X[0]);                              \
        if (KG > 1 && live_k1) ls_wgrad_load4<VX, FULL>(x, ldx, row_, ok_, safe, k_base + 64 + 4 * col, k_in, X[KG - 1]); } while (0)
#define LS_MFMA_STEP(A, X) do {                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
            dbacc[j] += A[j];                                                                                                \
            _Pragma("unroll") for (int kt = 0; kt < 4; ++kt)                                                                 \
                acc[j][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], X[0][kt], acc[j][kt], 0, 0, 0);                      \
            if (KG > 1 && live_k1) {                                                                                         \
                _Pragma("unroll") for (int kt = 4; kt < 4 * KG; ++kt)                                                        \
                    acc[j][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], X[KG - 1][kt & 3], acc[j][kt], 0, 0, 0);         \
            }                                                                                                                \
        } } while (0)
    // a0/x0 = operands of the current step, a1/x1 of the next, a2/x2 in flight for the one after.  The rotation copies sit at the
    // TOP of the iteration, so the s_waitcnt they need covers loads issued a whole iteration (32 MFMAs) earlier, and the loads
    // issued below stay in flight across this iteration's MFMAs.  Rows past the slice load zeros.
    LS_LOAD_STEP(b0 + sub, a1, z1, x1);
    LS_LOAD_STEP(b0 + 4 + sub, a2, z2, x2);
    for (long b = b0; b < b1; b += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0[j] = FZ ? a1[j] * (z1[j] > 0.0f ? 1.0f : z1[j] + zoff) : a1[j];
            a1[j] = a2[j];
            if (FZ) z1[j] = z2[j];
#pragma unroll
            for (int q = 0; q < KG; ++q) { x0[q][j] = x1[q][j]; x1[q][j] = x2[q][j]; }
        }
        if (FZ && write_gy) {
            const long row = b + sub;
            const int n = n_base + 4 * col;
            if (row < b1 && n < n_out) {
                float* dst = gy + row * (long)n_out + n;
                if (FULL && (n_out & 3) == 0) *(float4*)dst = make_float4(a0[0], a0[1], a0[2], a0[3]);
                else { for (int j = 0; j < 4; ++j) if (n + j < n_out) dst[j] = a0[j]; }
            }
        }
        LS_LOAD_STEP(b + 8 + sub, a2, z2, x2);
        LS_MFMA_STEP(a0, x0);
    }
#undef LS_MFMA_STEP
#undef LS_LOAD_STEP
}

// ---- the steady-state batch loop: three register stages used round-robin (loads for step i + 2 issue before the MFMAs of step i, no
// register rotation), every row known to exist (no row selects), column offsets of the four loads precomputed per lane (clamped into
// the row on edge tiles).  Per 32 MFMAs the wave issues 3-4 loads and, without the ELU factor, 4 VALU instructions (the bias sums);
// the rotating loop above costs ~44 (PMC: 1.4 VALU per MFMA, MFMA pipe 60 % busy).  Rows [b, b1) that do not fill a whole round of
// three steps, and slices shorter than 20 rows, are left to ls_wgrad_tile_loop.
template <int VEC>
__device__ __forceinline__ void ls_wgrad_offsets(int c0, int cmax, bool full, int (&o)[4]) {
    if (full) { o[0] = c0; o[1] = c0 + 1; o[2] = c0 + 2; o[3] = c0 + 3; return; }
    if (VEC == 2) { const int last = ((cmax + 3) & ~3) - 4; o[0] = c0 < last ? c0 : last; o[1] = o[2] = o[3] = o[0]; }
    else if (VEC == 1) { const int last = ((cmax + 1) & ~1) - 2; o[0] = c0 < last ? c0 : last; o[2] = c0 + 2 < last ? c0 + 2 : last; o[1] = o[0]; o[3] = o[2]; }
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = c0 + j < cmax ? c0 + j : cmax - 1;
    }
}
template <int VEC>
__device__ __forceinline__ void ls_wgrad_ld(const float* __restrict__ r, const int (&o)[4], float (&v)[4]) {
    if (VEC == 2) { const float4 t = *(const float4*)(r + o[0]); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else if (VEC == 1) { const float2 ta = *(const float2*)(r + o[0]), tb = *(const float2*)(r + o[2]); v[0] = ta.x; v[1] = ta.y; v[2] = tb.x; v[3] = tb.y; }
    else { v[0] = r[o[0]]; v[1] = r[o[1]]; v[2] = r[o[2]]; v[3] = r[o[3]]; }
}
#ifndef LS_WG_KNOCK
#define LS_WG_KNOCK 0        /* diagnostic builds only (wrong results): 1 = no operand loads in the steady-state loop, 2 = no MFMAs (one FMA per accumulator instead) */
#endif
template <int KG> struct LsWgradStage { float a[4], z[4], x[KG][4]; };

template <int VX, int VG, int KG, bool FZ, bool LIVE1 /* the second 64-column k group exists (compile-time: no branch between MFMAs) */,
          bool WGY /* this wave writes g_y, as whole 16-byte vectors (interior n tile, n_out % 4 == 0) */>
__device__ __forceinline__ long ls_wgrad_tile_loop3(const float* __restrict__ x, long ldx, const float* __restrict__ g, long ldg,
                                                    const float* __restrict__ z, long ldz, float* __restrict__ gy, long b0, long b1,
                                                    bool full, int n_base, int k_base, int n_out, int k_in, int sub, int col,
                                                    ls_v4f (&acc)[4][4 * KG], float (&dbacc)[4], float zoff) {
    if (b1 - b0 < 20) return b0;
    int og[4], ox0[4], ox1[4];
    ls_wgrad_offsets<VG>(n_base + 4 * col, n_out, full, og);
    ls_wgrad_offsets<VX>(k_base + 4 * col, k_in, full, ox0);
    constexpr bool live_k1 = KG > 1 && LIVE1;
    ls_wgrad_offsets<VX>(k_base + 64 + 4 * col, k_in, full, ox1);
    LsWgradStage<KG> s0, s1, s2;
    if (KG > 1 && !live_k1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s0.x[KG - 1][j] = s1.x[KG - 1][j] = s2.x[KG - 1][j] = 0.0f;
    }
#define LS_LD(ST, ROW) do { const long r_ = (ROW);                                                                           \
        ls_wgrad_ld<VG>(g + r_ * ldg, og, ST.a);                                                                              \
        if (FZ) ls_wgrad_ld<VG>(z + r_ * ldz, og, ST.z);                                                                      \
        ls_wgrad_ld<VX>(x + r_ * ldx, ox0, ST.x[0]);                                                                          \
        if (live_k1) ls_wgrad_ld<VX>(x + r_ * ldx, ox1, ST.x[KG - 1]); } while (0)
#define LS_CMP(ST, ROW) do {                                                                                                 \
        if (FZ) {                                                                                                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) ST.a[j] *= ST.z[j] > 0.0f ? 1.0f : ST.z[j] + zoff;                 \
            if (WGY) *(float4*)(gy + (ROW) * (long)n_out + n_base + 4 * col) = make_float4(ST.a[0], ST.a[1], ST.a[2], ST.a[3]);     \
        }                                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
            dbacc[j] += ST.a[j];                                                                                             \
            _Pragma("unroll") for (int kt = 0; kt < 4; ++kt)                                                                 \
                acc[j][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ST.a[j], ST.x[0][kt], acc[j][kt], 0, 0, 0);                \
            if (live_k1) {                                                                                                   \
                _Pragma("unroll") for (int kt = 4; kt < 4 * KG; ++kt)                                                        \
                    acc[j][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ST.a[j], ST.x[KG - 1][kt & 3], acc[j][kt], 0, 0, 0);   \
            }                                                                                                                \
        } } while (0)
    // one step with the next-but-one stage's loads SPREAD between the four MFMA groups (a load every 8 MFMAs) and pinned there with
    // scheduling barriers: clustered at the top of the step the same loads cost a quarter of the MFMA rate (tools/micro/mfma_peak:
    // 102 -> 127 TFLOP/s for 3 loads per 32 MFMAs), and left alone hipcc sinks them to their first use
#define LS_SB() __builtin_amdgcn_sched_barrier(0)
#define LS_GRP(ST, J) do {                                                                                                   \
        dbacc[J] += ST.a[J];                                                                                                 \
        __builtin_amdgcn_s_setprio(1);          /* the wave entering an MFMA group wins arbitration over the one issuing loads: +5 % */  \
        if (LS_WG_KNOCK != 2) {                                                                                              \
        _Pragma("unroll") for (int kt = 0; kt < 4; ++kt)                                                                     \
            acc[J][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ST.a[J], ST.x[0][kt], acc[J][kt], 0, 0, 0);                    \
        if (live_k1) {                                                                                                       \
            _Pragma("unroll") for (int kt = 4; kt < 4 * KG; ++kt)                                                            \
                acc[J][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ST.a[J], ST.x[KG - 1][kt & 3], acc[J][kt], 0, 0, 0);       \
        } } else { _Pragma("unroll") for (int kt = 0; kt < 4 * KG; ++kt) acc[J][kt][0] += ST.a[J] * ST.x[(kt >> 2) ? KG - 1 : 0][kt & 3]; }  \
        __builtin_amdgcn_s_setprio(0); } while (0)
#define LS_STEP(CS, CROW, LS, LROW) do { const long lr_ = (LROW);                                                            \
        if (FZ) {     /* g_z * elu'(z) for the whole step up front: spread between the MFMA groups it costs 10 % (measured) */   \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) CS.a[j] *= CS.z[j] > 0.0f ? 1.0f : CS.z[j] + zoff;                 \
            if (WGY) *(float4*)(gy + (CROW) * (long)n_out + n_base + 4 * col) = make_float4(CS.a[0], CS.a[1], CS.a[2], CS.a[3]);  \
        }                                                                                                                    \
        LS_SB(); LS_GRP(CS, 0); LS_SB();                                                                                     \
        if (LS_WG_KNOCK != 1) ls_wgrad_ld<VG>(g + lr_ * ldg, og, LS.a); LS_SB();                                             \
        LS_GRP(CS, 1); LS_SB();                                                                                              \
        if (LS_WG_KNOCK != 1) { if (FZ) ls_wgrad_ld<VG>(z + lr_ * ldz, og, LS.z); else ls_wgrad_ld<VX>(x + lr_ * ldx, ox0, LS.x[0]); }   \
        LS_SB(); LS_GRP(CS, 2); LS_SB();                                                                                     \
        if (LS_WG_KNOCK != 1) { if (FZ) ls_wgrad_ld<VX>(x + lr_ * ldx, ox0, LS.x[0]); else if (live_k1) ls_wgrad_ld<VX>(x + lr_ * ldx, ox1, LS.x[KG - 1]); }  \
        LS_SB(); LS_GRP(CS, 3); LS_SB();                                                                                     \
        if (LS_WG_KNOCK != 1) { if (FZ && live_k1) ls_wgrad_ld<VX>(x + lr_ * ldx, ox1, LS.x[KG - 1]); }                      \
        LS_SB(); } while (0)
    long b = b0;
    LS_LD(s0, b + sub);
    LS_LD(s1, b + 4 + sub);
    for (; b + 20 <= b1; b += 12) {
        LS_STEP(s0, b + sub, s2, b + 8 + sub);
        LS_STEP(s1, b + 4 + sub, s0, b + 12 + sub);
        LS_STEP(s2, b + 8 + sub, s1, b + 16 + sub);
    }
    LS_CMP(s0, b + sub);             // the two stages still in flight (their rows exist: loaded under the loop condition)
    LS_CMP(s1, b + 4 + sub);
#undef LS_STEP
#undef LS_GRP
#undef LS_SB
#undef LS_CMP
#undef LS_LD
    return b + 8;
}

// KG: 64-wide k groups per tile (1 for k_in <= 64, else 2)
template <int VX, int VG, int KG, bool FZ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void lsim_k_linear_wgrad_tiled(const float* __restrict__ x, long ldx, const float* __restrict__ g, long ldg, const float* __restrict__ z, long ldz,
                               float* __restrict__ gy, long batch, int k_in, int n_out,
                               int k_blocks, int tiles, int slices, long rows_per_slice, float* __restrict__ part_dw, float* __restrict__ part_db, float zoff) {
    // zoff: the activation whose saved OUTPUT z is: d act / d pre = z > 0 ? 1 : z + zoff -- 1 for ELU (alpha = 1: torch's elu_backward on the result), 0 for
    // ReLU (its output is exactly 0 where the unit is off: torch's threshold_backward)
    // one BLOCK per (output tile, batch slice): its four waves split the slice's rows four ways and add their accumulators in LDS at the
    // end, so there is one partial result per block but four times as many waves in flight (two per SIMD: one wave's operand loads,
    // register rotation and waits hide behind the other's MFMAs)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                                // wave-uniform: scalar loop control
    // XCD-aware order: the dispatcher deals consecutive blocks round-robin to the 8 XCDs, each with its own L2.  Renumber so that XCD x owns
    // a CONTIGUOUS range of (slice, tile) pairs: all tiles of a slice then read that slice's rows of x / g through one L2 (fetched over
    // the fabric once instead of once per XCD that holds one of its tiles).  v is a bijection of [0, gridDim.x) for any block count.
    const int total = (int)gridDim.x, xcd = (int)blockIdx.x & 7;
    const long v = (long)xcd * (total >> 3) + (xcd < (total & 7) ? xcd : (total & 7)) + ((long)blockIdx.x >> 3);
    const long slice = v / tiles;
    const int tile = (int)(v - slice * tiles);
    const int nb = tile / k_blocks, kb = tile - nb * k_blocks;
    const int n_base = nb * 64, k_base = kb * 64 * KG;
    const int sub = lane >> 4, col = lane & 15;
    const long s0 = slice * rows_per_slice;
    long s1 = s0 + rows_per_slice;
    if (s1 > batch) s1 = batch;
    const long quarter = (((s1 - s0) + 3) / 4 + 3) & ~3L;          // rows per wave, a multiple of the 4 rows one MFMA step consumes
    long b0 = s0 + wv * quarter, b1 = b0 + quarter;
    if (b0 > s1) b0 = s1;
    if (b1 > s1) b1 = s1;
    ls_v4f acc[4][4 * KG];
    float dbacc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kt = 0; kt < 4 * KG; ++kt) acc[j][kt] = (ls_v4f){0.0f, 0.0f, 0.0f, 0.0f};
    const bool full = (n_base + 64 <= n_out) && (k_base + 64 * KG <= k_in);      // wave-uniform: interior tile
    const long safe = b0 < batch ? b0 : batch - 1;          // an existing row for the masked loads of rows past the slice
    {
        const bool live1 = KG == 1 || k_base + 64 < k_in;
        const bool wgy = FZ && kb == 0 && gy != nullptr, wgy_vec = (n_out & 3) == 0 && n_base + 64 <= n_out;      // gy == NULL: nobody needs grad_pre (a network's first layer)
#define LS_L3(LIVE1, WGY) b0 = ls_wgrad_tile_loop3<VX, VG, KG, FZ, LIVE1, WGY>(x, ldx, g, ldg, z, ldz, gy, b0, b1, full, n_base, k_base, n_out, k_in, sub, col, acc, dbacc, zoff)
        if (!wgy) { if (live1) LS_L3(true, false); else LS_L3(false, false); }
        else if (wgy_vec) { if (live1) LS_L3(true, true); else LS_L3(false, true); }
        // (g_y rows of an edge n tile: everything goes through the guarded loop below)
#undef LS_L3
    }
    ls_wgrad_tile_loop<VX, VG, KG, false, FZ>(x, ldx, g, ldg, z, ldz, gy, kb == 0 && gy != nullptr, b0, b1, safe, n_base, k_base, n_out, k_in, sub, col, acc, dbacc, zoff);   // the last < 12 rows
    // block reduction: wave 0 stores its accumulators to LDS (lane-major, 16-byte vectors: no bank conflicts), waves 1-3 add theirs in
    // turn; wave 3 ends up with the block's sums and writes the partial tile
    __shared__ float4 red[4 * 4 * KG][64];
    __shared__ float redb[4][64];
    for (int turn = 0; turn < 4; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int kt = 0; kt < 4 * KG; ++kt) {
                    float4 v = make_float4(acc[j][kt][0], acc[j][kt][1], acc[j][kt][2], acc[j][kt][3]);
                    if (turn > 0) { const float4 o = red[j * 4 * KG + kt][lane]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    if (turn < 3) red[j * 4 * KG + kt][lane] = v;
                    else { acc[j][kt][0] = v.x; acc[j][kt][1] = v.y; acc[j][kt][2] = v.z; acc[j][kt][3] = v.w; }
                }
                float d = dbacc[j];
                if (turn > 0) d += redb[j][lane];
                if (turn < 3) redb[j][lane] = d; else dbacc[j] = d;
            }
        }
        __syncthreads();
    }
    if (wv != 3) return;
    // accumulator (j, kt)[r] of lane (sub, col) is the output n = n_base + 4 (4 sub + r) + j, k = k_base + 64 (kt / 4) + 4 col + kt % 4
    float* pw = part_dw + (size_t)slice * n_out * k_in;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n_base + 4 * (4 * sub + r) + j;
            if (n >= n_out) continue;
#pragma unroll
            for (int q = 0; q < KG; ++q) {
                const int k = k_base + 64 * q + 4 * col;
                if (k >= k_in) continue;
                float* dst = pw + (size_t)n * k_in + k;
                if ((k_in & 3) == 0) {      // k + 3 < k_in and 16-byte aligned
                    *(float4*)dst = make_float4(acc[j][4 * q][r], acc[j][4 * q + 1][r], acc[j][4 * q + 2][r], acc[j][4 * q + 3][r]);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (k + c < k_in) dst[c] = acc[j][4 * q + c][r];
                }
            }
        }
    if (part_db && kb == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = dbacc[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = n_base + 4 * col + j;
            if (sub == 0 && n < n_out) part_db[(size_t)slice * n_out + n] = v;
        }
    }
}

// ---- weight gradient of the 128-multiple layers on the bf16 matrix pipe, fp32 in and out (LSIM_WGRAD_SPLIT_BF16=1; tools/micro/split_bf16.hip has
// the arithmetic and its measured accuracy): an fp32 value is EXACTLY the sum of three bf16 terms a0 + a1 + a2 (3 x 8 significand bits); the six
// products a_i b_j with i + j <= 2 are exact in the pipe's fp32 accumulator and leave a truncation of 2^-24 |a||b| per product -- fp32's own
// rounding -- while v_mfma_f32_16x16x32_bf16 moves 16 x the multiply-adds per cycle of v_mfma_f32_16x16x4_f32: six of them per K chunk are
// 2.67 x the fp32 pipe's rate.  One block per (128 x 128 output tile, batch slice); per step of 32 rows:
//   load  : thread (half, c4, rg) fetches rows 8 rg .. + 7 of column quad c4 of g (half 0; with z: times elu'(z), g_y stored, db summed) or x (half 1)
//   split : its 32 values into three bf16 planes, 8 consecutive rows of one column = one 16-byte LDS write (layout [plane][column][32 k + 8 pad]:
//           the pad makes both the writes of 16 consecutive lanes and the fragment reads of a quarter wave conflict-free)
//   mfma  : wave (wn, wk) owns the 64 x 64 quarter: 4 + 4 fragments x 3 planes by ds_read_b128, 16 tiles x 6 products, smallest terms first
// The next step's global loads are issued before the MFMAs; two blocks per CU (61 KB of LDS each) overlap one's load / split with the other's MFMAs.
typedef __bf16 ls_bf8 __attribute__((ext_vector_type(8)));
#define LS_SP_STRIDE 40
#ifndef LS_SP_KNOCKOUT
#define LS_SP_KNOCKOUT 0      // diagnostic builds (tools/wgrad_split_probe.py): 1 no MFMA phase, 2 no global loads in the loop, 3 no split / LDS writes
#endif
template <bool FZ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void lsim_k_linear_wgrad_split(const float* __restrict__ x, long ldx, const float* __restrict__ g, long ldg, const float* __restrict__ z, long ldz,
                               float* __restrict__ gy, long batch, int k_in, int n_out,
                               int k_blocks, int tiles, int slices, long rows_per_slice, float* __restrict__ part_dw, float* __restrict__ part_db) {
    __shared__ __attribute__((aligned(16))) __bf16 planes[2][3][128 * LS_SP_STRIDE];
    const int total = (int)gridDim.x, xcd = (int)blockIdx.x & 7;          // XCD-aware order, as lsim_k_linear_wgrad_tiled
    const long v = (long)xcd * (total >> 3) + (xcd < (total & 7) ? xcd : (total & 7)) + ((long)blockIdx.x >> 3);
    const long slice = v / tiles;
    const int tile = (int)(v - slice * tiles);
    const int nb = tile / k_blocks, kb = tile - nb * k_blocks;
    const int n_base = nb * 128, k_base = kb * 128;
    const int t = threadIdx.x, lane = t & 63;
    const int half = __builtin_amdgcn_readfirstlane(t >> 7);                // waves 0, 1: g; waves 2, 3: x
    const int pch = t & 127, rg = pch & 3, c4 = pch >> 2;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6), wn = wv & 1, wk = wv >> 1;
    const int col16 = lane & 15, kg = lane >> 4;
    const long s0 = slice * rows_per_slice;
    long s1 = s0 + rows_per_slice;
    if (s1 > batch) s1 = batch;
    const float* src = half ? x + k_base + 4 * c4 : g + n_base + 4 * c4;
    const long ld = half ? ldx : ldg;
    const bool is_g = half == 0, wgy = FZ && is_g && kb == 0 && gy != nullptr;
    float4 cur[8], zc[8];
    float dbacc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    ls_v4f acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = (ls_v4f){0.0f, 0.0f, 0.0f, 0.0f};
#define LS_SP_LOAD(RB) do {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                       \
            const long row_ = (RB) + 8 * rg + i;                                                                              \
            const bool in_ = row_ < s1;                                                                                       \
            const long rr_ = in_ ? row_ : s0;                                                                                 \
            const float4 v_ = *(const float4*)(src + rr_ * ld);                                                               \
            cur[i] = in_ ? v_ : make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                                          \
            if (FZ && is_g) zc[i] = *(const float4*)(z + rr_ * ldz + n_base + 4 * c4);                                         \
        } } while (0)
    if (s0 < s1) LS_SP_LOAD(s0);
    for (long rb = s0; rb < s1; rb += 32) {
        if (FZ && is_g) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                cur[i].x *= zc[i].x > 0.0f ? 1.0f : zc[i].x + 1.0f; cur[i].y *= zc[i].y > 0.0f ? 1.0f : zc[i].y + 1.0f;
                cur[i].z *= zc[i].z > 0.0f ? 1.0f : zc[i].z + 1.0f; cur[i].w *= zc[i].w > 0.0f ? 1.0f : zc[i].w + 1.0f;
                const long row = rb + 8 * rg + i;
                if (wgy && row < s1) *(float4*)(gy + row * (long)n_out + n_base + 4 * c4) = cur[i];
            }
        }
        if (is_g) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { dbacc[0] += cur[i].x; dbacc[1] += cur[i].y; dbacc[2] += cur[i].z; dbacc[3] += cur[i].w; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (LS_SP_KNOCKOUT == 3 && rb > s0) break;
            ls_bf8 p0, p1, p2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float val = j == 0 ? cur[i].x : j == 1 ? cur[i].y : j == 2 ? cur[i].z : cur[i].w;
                const __bf16 h = (__bf16)val;
                const float r1 = val - (float)h;           // exact
                const __bf16 m = (__bf16)r1;
                const float r2 = r1 - (float)m;            // exact
                p0[i] = h; p1[i] = m; p2[i] = (__bf16)r2;
            }
            const int o = (4 * c4 + j) * LS_SP_STRIDE + 8 * rg;
            *(ls_bf8*)&planes[half][0][o] = p0; *(ls_bf8*)&planes[half][1][o] = p1; *(ls_bf8*)&planes[half][2][o] = p2;
        }
        __syncthreads();
        if (LS_SP_KNOCKOUT != 2 && rb + 32 < s1) LS_SP_LOAD(rb + 32);
        if (LS_SP_KNOCKOUT == 1) { __syncthreads(); continue; }
        ls_bf8 fa[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[j][pl] = *(const ls_bf8*)&planes[0][pl][(64 * wn + 16 * j + col16) * LS_SP_STRIDE + 8 * kg];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ls_bf8 fb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[pl] = *(const ls_bf8*)&planes[1][pl][(64 * wk + 16 * q + col16) * LS_SP_STRIDE + 8 * kg];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j][2], fb[0], acc[j][q], 0, 0, 0);
                acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j][1], fb[1], acc[j][q], 0, 0, 0);
                acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j][0], fb[2], acc[j][q], 0, 0, 0);
                acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j][1], fb[0], acc[j][q], 0, 0, 0);
                acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j][0], fb[1], acc[j][q], 0, 0, 0);
                acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j][0], fb[0], acc[j][q], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#undef LS_SP_LOAD
    // accumulator (j, q)[r] of lane (kg, col16): n = n_base + 64 wn + 16 j + 4 kg + r (the A operand's row), k = k_base + 64 wk + 16 q + col16
    float* pw = part_dw + (size_t)slice * n_out * k_in;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n_base + 64 * wn + 16 * j + 4 * kg + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) pw[(size_t)n * k_in + k_base + 64 * wk + 16 * q + col16] = acc[j][q][r];
        }
    if (part_db && kb == 0 && is_g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float d = dbacc[j];
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            if (rg == 0) part_db[(size_t)slice * n_out + n_base + 4 * c4 + j] = d;
        }
    }
}
static int g_ls_wgrad_split = -1;        // -1: not decided yet (LSIM_WGRAD_SPLIT_BF16 at the first use), 0 / 1
static bool ls_wgrad_split_enabled() {
    if (g_ls_wgrad_split < 0) { const char* e = getenv("LSIM_WGRAD_SPLIT_BF16"); g_ls_wgrad_split = (e && e[0] == '1') ? 1 : 0; }
    return g_ls_wgrad_split == 1;
}
extern "C" int lsim_wgrad_split_bf16(int on) {
    const int was = ls_wgrad_split_enabled() ? 1 : 0;
    if (on == 0 || on == 1) g_ls_wgrad_split = on;
    return was;
}

#define LS_WGRAD_MAX_TILES 32

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

// launch plan shared by the workspace query and the call
struct LsWgradPlan {
    int small;            // 1: whole output in one wave's accumulators (lsim_k_linear_wgrad), 0: 64 x 128 tiles
    int partials;         // number of partial results per output element
    long rows;            // batch rows per partial
    int n_blocks, k_blocks;
    int split;            // 1: 128 x 128 tiles on the bf16 pipe (lsim_k_linear_wgrad_split) when the operands turn out 16-byte aligned
};
static int ls_wgrad_plan(long batch, int k_in, int n_out, LsWgradPlan* p) {
    // measured against TunableOp-selected hipBLASLt (tools/archive/wgrad_sweep.sh): the block-cooperative tiled kernel wins up to 512 -> 256
    // (295 vs 363 us) when the operand rows are 16-byte aligned; with unaligned rows (k_in % 4 != 0: 238 -> 512) it ties on dW + db alone
    // (394 + 7 vs 350 + 64 us, tools/archive/wgrad_one.sh) and wins once the ELU backward rides along (saves elu_backward's 100 us pass)
    const long count = (long)k_in * n_out;
    if (batch <= 0 || k_in <= 0 || n_out <= 0 || count > 140000) return LSIM_E_UNSUPPORTED;
    const int nt = (n_out + 15) / 16, kt = (k_in + 15) / 16;
    if (nt <= 8 && kt <= 8 && nt * kt <= LS_WGRAD_MAX_TILES && (long)n_out * k_in <= 4096) {
        p->small = 1;
        p->split = 0;
        long rpw = ((batch + 1023) / 1024 + 3) & ~3L;      // one wave per SIMD of the 256-CU part; 4 rows per MFMA
        if (rpw < 16) rpw = 16;
        p->rows = rpw;
        p->partials = (int)((batch + rpw - 1) / rpw);
        p->n_blocks = p->k_blocks = 1;
        return LSIM_OK;
    }
    p->small = 0;
    p->split = 0;
    p->n_blocks = (n_out + 63) / 64;
    p->k_blocks = k_in <= 64 ? 1 : (k_in + 127) / 128;     // 64-wide k tile for the narrow inputs, 128-wide otherwise
    const int tiles = p->n_blocks * p->k_blocks;
    long slices = 512 / tiles;                              // 4 waves per block: tiles * slices * 4 = 2048 waves, two per SIMD
    const long cap = (32L << 20) / ((long)n_out * k_in * 4); // keep the partial results (written once, read once) under 32 MB
    if (slices > cap) slices = cap;
    if (slices < 1) slices = 1;
    long rps = ((batch + slices - 1) / slices + 15) & ~15L;  // each of the block's four waves takes a quarter, in steps of 4 rows
    if (rps < 64) rps = 64;
    if (ls_wgrad_split_enabled() && n_out % 128 == 0 && k_in % 128 == 0) {
        p->split = 1;
        const int tiles128 = (n_out / 128) * (k_in / 128);
        slices = 512 / tiles128;                             // two blocks per CU
        if (slices > cap) slices = cap;
        if (slices < 1) slices = 1;
        rps = ((batch + slices - 1) / slices + 31) & ~31L;   // whole steps of 32 rows
    }
    p->rows = rps;
    p->partials = (int)((batch + rps - 1) / rps);
    return LSIM_OK;
}

extern "C" int lsim_linear_wgrad_workspace(long batch, int k_in, int n_out, size_t* bytes, int* num_partials) {
    if (!bytes || !num_partials) return LSIM_E_INVALID;
    LsWgradPlan p;
    int rc = ls_wgrad_plan(batch, k_in, n_out, &p);
    if (rc != LSIM_OK) return rc;
    *num_partials = p.partials;
    *bytes = (size_t)p.partials * ((size_t)n_out * k_in + n_out) * sizeof(float);
    return LSIM_OK;
}

static int ls_linear_wgrad_impl(const float* x, int64_t ldx, const float* g, int64_t ldg, const float* z, int64_t ldz, float* gy, int64_t batch,
                                int k_in, int n_out, float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream,
                                lsim_wgrad_pending* pending = nullptr, float zoff = 1.0f) {
    if (!x || !g || !dw || !workspace) return LSIM_E_INVALID;
    LsWgradPlan p;
    int rc = ls_wgrad_plan(batch, k_in, n_out, &p);
    if (rc != LSIM_OK) return rc;
    const size_t need = (size_t)p.partials * ((size_t)n_out * k_in + n_out) * sizeof(float);
    if (workspace_bytes < need || ldx < k_in || ldg < n_out) return LSIM_E_INVALID;
    const bool fz = z != nullptr;
    if (fz && (p.small || ldz < n_out)) return p.small ? LSIM_E_UNSUPPORTED : LSIM_E_INVALID;
    float* pdw = (float*)workspace;
    float* pdb = db ? pdw + (size_t)p.partials * n_out * k_in : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (p.small) {
        const int nt = (n_out + 15) / 16, kt = (k_in + 15) / 16;
        const int blocks = (p.partials + LS_WGRAD_WAVES_PER_BLOCK - 1) / LS_WGRAD_WAVES_PER_BLOCK;
        int bad = 1;
        switch (nt) {
#define LS_N(NT) case NT: bad = ls_wgrad_dispatch_k<NT>(kt, x, ldx, g, ldg, batch, k_in, n_out, p.rows, blocks, pdw, pdb, s); break;
            LS_N(1) LS_N(2) LS_N(3) LS_N(4) LS_N(5) LS_N(6) LS_N(7) LS_N(8)
#undef LS_N
            default: break;
        }
        if (bad) return LSIM_E_UNSUPPORTED;
    } else {
        const int tiles = p.n_blocks * p.k_blocks;
        const int blocks = tiles * p.partials;               // one block of four waves per (tile, slice)
        const int vx = ((ldx % 4 == 0) && (((uintptr_t)x & 15) == 0)) ? 2 : (((ldx % 2 == 0) && (((uintptr_t)x & 7) == 0)) ? 1 : 0);
        int vg = ((ldg % 4 == 0) && (((uintptr_t)g & 15) == 0)) ? 2 : 0;
        if (fz && !((ldz % 4 == 0) && (((uintptr_t)z & 15) == 0))) vg = 0;
        const int kg = k_in <= 64 ? 1 : 2;
        if (p.split && zoff == 1.0f && vx == 2 && vg == 2 && (!gy || (((uintptr_t)gy & 15) == 0))) {      // (the bf16-pipe kernel hard-codes ELU)
            const int kb128 = k_in / 128, tiles128 = (n_out / 128) * kb128;
            if (fz) hipLaunchKernelGGL((lsim_k_linear_wgrad_split<true>), dim3(tiles128 * p.partials), dim3(256), 0, s, x, (long)ldx, g, (long)ldg, z, (long)ldz,
                                       gy, (long)batch, k_in, n_out, kb128, tiles128, p.partials, p.rows, pdw, pdb);
            else hipLaunchKernelGGL((lsim_k_linear_wgrad_split<false>), dim3(tiles128 * p.partials), dim3(256), 0, s, x, (long)ldx, g, (long)ldg, z, (long)ldz,
                                    gy, (long)batch, k_in, n_out, kb128, tiles128, p.partials, p.rows, pdw, pdb);
        } else {
#define LS_T(VX, VG, KG, FZ) hipLaunchKernelGGL((lsim_k_linear_wgrad_tiled<VX, VG, KG, FZ>), dim3(blocks), dim3(256), 0, s, x, (long)ldx, g, (long)ldg, z, (long)ldz, \
                                                gy, (long)batch, k_in, n_out, p.k_blocks, tiles, p.partials, p.rows, pdw, pdb, zoff)
#define LS_TZ(VX, VG, KG) do { if (fz) LS_T(VX, VG, KG, true); else LS_T(VX, VG, KG, false); } while (0)
#define LS_TK(VX, VG) do { if (kg == 1) LS_TZ(VX, VG, 1); else LS_TZ(VX, VG, 2); } while (0)
        if (vg == 2) { if (vx == 2) LS_TK(2, 2); else if (vx == 1) LS_TK(1, 2); else LS_TK(0, 2); }
        else         { if (vx == 2) LS_TK(2, 0); else if (vx == 1) LS_TK(1, 0); else LS_TK(0, 0); }
#undef LS_TK
#undef LS_TZ
#undef LS_T
        }
    }
    const int count = n_out * k_in;
    if (pending) {       // the caller sums the partial results later (lsim_wgrad_reduce_batch)
        pending->part = pdw; pending->part2 = pdb; pending->out = dw; pending->out2 = db;
        pending->num_partials = p.partials; pending->count = count; pending->count2 = db ? n_out : 0; pending->reserved = 0;
        return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
    }
    hipLaunchKernelGGL(lsim_k_wgrad_reduce, dim3(ls_reduce_blocks(pdw, count) + (db ? (n_out + 15) / 16 : 0)), dim3(256), 0, s, (const float*)pdw, p.partials, count, dw,
                       (const float*)pdb, n_out, db);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

extern "C" int lsim_linear_wgrad_deferred(const float* x, int64_t ldx, const float* g, int64_t ldg, int64_t batch, int k_in, int n_out,
                                          float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream, lsim_wgrad_pending* pending) {
    if (!pending) return LSIM_E_INVALID;
    return ls_linear_wgrad_impl(x, ldx, g, ldg, nullptr, 0, nullptr, batch, k_in, n_out, dw, db, workspace, workspace_bytes, stream, pending);
}
extern "C" int lsim_linear_elu_wgrad_deferred(const float* x, int64_t ldx, const float* grad_out, int64_t ldg, const float* elu_out, int64_t ldz, int64_t batch,
                                              int k_in, int n_out, float* dw, float* db, float* grad_pre, void* workspace, size_t workspace_bytes, void* stream,
                                              lsim_wgrad_pending* pending) {
    if (!elu_out || !pending) return LSIM_E_INVALID;
    return ls_linear_wgrad_impl(x, ldx, grad_out, ldg, elu_out, ldz, grad_pre, batch, k_in, n_out, dw, db, workspace, workspace_bytes, stream, pending);
}
extern "C" int lsim_wgrad_reduce_batch(const lsim_wgrad_pending* items, int n, void* stream) {
    if (n < 0 || (n > 0 && !items)) return LSIM_E_INVALID;
    for (int i0 = 0; i0 < n; i0 += LS_REDUCE_BATCH) {
        LsReduceBatch b;
        b.n = n - i0 < LS_REDUCE_BATCH ? n - i0 : LS_REDUCE_BATCH;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            const lsim_wgrad_pending& p = items[i0 + i];
            if (!p.part || !p.out || p.num_partials <= 0 || p.count <= 0 || (p.count2 > 0 && (!p.part2 || !p.out2))) return LSIM_E_INVALID;
            b.it[i] = p;
            b.first[i] = blocks;
            blocks += ls_reduce_blocks(p.part, p.count) + (p.count2 > 0 ? (p.count2 + 15) / 16 : 0);
        }
        b.first[b.n] = blocks;
        hipLaunchKernelGGL(lsim_k_wgrad_reduce_batch, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    }
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

extern "C" int lsim_linear_wgrad(const float* x, int64_t ldx, const float* g, int64_t ldg, int64_t batch, int k_in, int n_out,
                                 float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    return ls_linear_wgrad_impl(x, ldx, g, ldg, nullptr, 0, nullptr, batch, k_in, n_out, dw, db, workspace, workspace_bytes, stream);
}

extern "C" int lsim_linear_elu_wgrad(const float* x, int64_t ldx, const float* grad_out, int64_t ldg, const float* elu_out, int64_t ldz, int64_t batch,
                                     int k_in, int n_out, float* dw, float* db, float* grad_pre, void* workspace, size_t workspace_bytes, void* stream) {
    if (!elu_out) return LSIM_E_INVALID;
    return ls_linear_wgrad_impl(x, ldx, grad_out, ldg, elu_out, ldz, grad_pre, batch, k_in, n_out, dw, db, workspace, workspace_bytes, stream);
}

// Linear + ReLU (the AMP discriminator's trunk, DISC:18-25): the same kernels with d act / d pre = [output > 0] (zoff = 0)
extern "C" int lsim_linear_relu_wgrad(const float* x, int64_t ldx, const float* grad_out, int64_t ldg, const float* relu_out, int64_t ldz, int64_t batch,
                                      int k_in, int n_out, float* dw, float* db, float* grad_pre, void* workspace, size_t workspace_bytes, void* stream) {
    if (!relu_out) return LSIM_E_INVALID;
    return ls_linear_wgrad_impl(x, ldx, grad_out, ldg, relu_out, ldz, grad_pre, batch, k_in, n_out, dw, db, workspace, workspace_bytes, stream, nullptr, 0.0f);
}
extern "C" int lsim_linear_relu_wgrad_deferred(const float* x, int64_t ldx, const float* grad_out, int64_t ldg, const float* relu_out, int64_t ldz, int64_t batch,
                                               int k_in, int n_out, float* dw, float* db, float* grad_pre, void* workspace, size_t workspace_bytes, void* stream,
                                               lsim_wgrad_pending* pending) {
    if (!relu_out || !pending) return LSIM_E_INVALID;
    return ls_linear_wgrad_impl(x, ldx, grad_out, ldg, relu_out, ldz, grad_pre, batch, k_in, n_out, dw, db, workspace, workspace_bytes, stream, pending, 0.0f);
}

// ---- Sinkhorn-Knopp assignment of HIMEstimator (HES:119-133): Q = exp(scores / eps)^T, then `iters` x {rows sum to 1/K, columns
// sum to 1/B}, returned as (Q * B)^T.  Every step of the reference only rescales rows (one factor per prototype) or columns (one
// factor per sample), so Q[k, b] = E[b, k] * u[k] * v[b] with E = exp(scores / eps) throughout (the initial division by the total
// sum cancels in the first row normalisation).  Instead of ~26 torch kernels over the (K, B) matrix per call:
//   first: E, per-block partial column sums of E                                  -> u[k] = 1 / (K * sum_b E[b, k])
//   mid  : v[b] = 1 / (B * sum_k E[b, k] u[k]); partial column sums of E * v       -> u[k] = 1 / (K * sum_b E[b, k] v[b])
//   last : out[b, k] = B * E[b, k] * u[k] * v[b] = E[b, k] u[k] / sum_k' E[b, k'] u[k']
// One THREAD per sample row with the row's K <= 64 values in registers (row sums, maxima and exponentials need no cross-lane
// traffic); a block owns LS_SK_ROWS = 256 consecutive samples and forms its partial column sums through an LDS transpose in a
// fixed order (deterministic).  blockIdx.y selects one of several independent matrices laid out back to back.
#define LS_SK_ROWS 256

template <int KMAX>
__device__ __forceinline__ void ls_row_load(const float* __restrict__ p, int K, bool vec, float fill, float (&v)[KMAX]) {
    // loads from clamped offsets + selects: no branch per element (K is a run-time value)
    if (vec) {        // K % 4 == 0 and 16-byte aligned rows
#pragma unroll
        for (int k = 0; k < KMAX; k += 4) {
            const float4 t = *(const float4*)(p + (k < K ? k : 0));
            v[k] = k < K ? t.x : fill; v[k + 1] = k < K ? t.y : fill; v[k + 2] = k < K ? t.z : fill; v[k + 3] = k < K ? t.w : fill;
        }
    } else {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { const float t = p[k < K ? k : 0]; v[k] = k < K ? t : fill; }
    }
}
template <int KMAX>
__device__ __forceinline__ void ls_row_store(float* __restrict__ p, int K, bool vec, const float (&v)[KMAX]) {
    if (vec) {
#pragma unroll
        for (int k = 0; k < KMAX; k += 4) if (k < K) *(float4*)(p + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) p[k] = v[k];
    }
}

// partial column sums of a block's 256 rows (one row per thread, zeros for rows past the batch): rows go through an LDS tile in two
// rounds of 128, thread (g, k) adds rows g, g + G, ... of column k, thread k adds the G group sums; every order is fixed.
// tile: LS_COLSUM_TILE(KMAX) floats.
#define LS_COLSUM_TILE(KMAX) (128 * ((KMAX) + 1))
template <int KMAX>
__device__ __forceinline__ void ls_block_colsum(const float (&v)[KMAX], float* __restrict__ tile, float* __restrict__ part_row /* [K] */, int K) {
    constexpr int G = 256 / KMAX;
    const int t = threadIdx.x, k = t & (KMAX - 1), g = t / KMAX;
    float s = 0.0f;
    for (int round = 0; round < 2; ++round) {
        __syncthreads();                      // the tile may still be read (previous round / previous use)
        if ((t >> 7) == round) {
#pragma unroll
            for (int j = 0; j < KMAX; ++j) tile[(t & 127) * (KMAX + 1) + j] = v[j];
        }
        __syncthreads();
        for (int r = g; r < 128; r += G) s += tile[r * (KMAX + 1) + k];
    }
    __syncthreads();
    tile[g * (KMAX + 1) + k] = s;
    __syncthreads();
    if (t < K) {
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < G; ++j) a += tile[j * (KMAX + 1) + t];
        part_row[t] = a;
    }
}

template <int KMAX>
__global__ __launch_bounds__(256) void lsim_k_sinkhorn_first(const float* __restrict__ scores, long lds, long scores_mat_stride, long batch, int K,
                                                             float inv_eps, bool vec_in, float* __restrict__ E, float* __restrict__ part) {
    __shared__ float tile[LS_COLSUM_TILE(KMAX)];
    const long m = blockIdx.y, b = (long)blockIdx.x * LS_SK_ROWS + threadIdx.x;
    const bool vec = (K & 3) == 0;
    float v[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) v[k] = 0.0f;
    if (b < batch) {
        ls_row_load<KMAX>(scores + m * scores_mat_stride + b * lds, K, vec_in, 0.0f, v);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) v[k] = k < K ? expf(v[k] * inv_eps) : 0.0f;
        ls_row_store<KMAX>(E + (m * batch + b) * K, K, vec, v);
    }
    ls_block_colsum<KMAX>(v, tile, part + (m * gridDim.x + blockIdx.x) * K, K);
}

template <int KMAX>
__global__ __launch_bounds__(256) void lsim_k_sinkhorn_mid(const float* __restrict__ E, const float* __restrict__ u, long batch, int K,
                                                           float* __restrict__ part) {
    __shared__ float tile[LS_COLSUM_TILE(KMAX)];
    __shared__ float us[KMAX];
    const long m = blockIdx.y, b = (long)blockIdx.x * LS_SK_ROWS + threadIdx.x;
    if (threadIdx.x < KMAX) us[threadIdx.x] = (int)threadIdx.x < K ? u[m * 64 + threadIdx.x] : 0.0f;
    __syncthreads();
    float v[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) v[k] = 0.0f;
    if (b < batch) {
        ls_row_load<KMAX>(E + (m * batch + b) * K, K, (K & 3) == 0, 0.0f, v);
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) t = fmaf(v[k], us[k], t);
        const float vb = 1.0f / ((float)batch * t);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) v[k] *= vb;
    }
    ls_block_colsum<KMAX>(v, tile, part + (m * gridDim.x + blockIdx.x) * K, K);
}

template <int KMAX>
__global__ __launch_bounds__(256) void lsim_k_sinkhorn_last(const float* __restrict__ E, const float* __restrict__ u, long batch, int K,
                                                            float* __restrict__ out) {
    __shared__ float us[KMAX];
    const long m = blockIdx.y, b = (long)blockIdx.x * LS_SK_ROWS + threadIdx.x;
    if (threadIdx.x < KMAX) us[threadIdx.x] = (int)threadIdx.x < K ? u[m * 64 + threadIdx.x] : 0.0f;
    __syncthreads();
    if (b >= batch) return;
    float v[KMAX];
    ls_row_load<KMAX>(E + (m * batch + b) * K, K, (K & 3) == 0, 0.0f, v);
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { v[k] *= us[k]; t += v[k]; }
    const float vb = 1.0f / ((float)batch * t);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) v[k] = (float)batch * v[k] * vb;
    ls_row_store<KMAX>(out + (m * batch + b) * K, K, (K & 3) == 0, v);
}

// u[k] = 1 / (K * sum over blocks of part[block][k]): 1024 threads = 32 (K <= 32) or 16 interleaved slices per prototype, eight partial
// rows in flight per thread (one block, nothing else to hide the loads behind: the 1600 rows the estimator's score kernel leaves took
// 15 us with two in flight), the slices combined in LDS; every order is fixed
__global__ __launch_bounds__(1024) void lsim_k_sinkhorn_scale(const float* __restrict__ part, int blocks, int K, float* __restrict__ u) {
    __shared__ float red[32][65];
    const int kp = K <= 32 ? 32 : 64, slices = 1024 / kp;
    const int k = threadIdx.x & (kp - 1), sl = threadIdx.x / kp;
    part += (size_t)blockIdx.y * blocks * K;
    u += blockIdx.y * 64;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.0f;
    if (k < K) {
        int i = sl;
        for (; i + 7 * slices < blocks; i += 8 * slices) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += part[(size_t)(i + j * slices) * K + k];
        }
        for (; i < blocks; i += slices) s[0] += part[(size_t)i * K + k];
    }
    red[sl][k] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (sl == 0 && k < K) {
        float t = 0.0f;
        for (int j = 0; j < slices; ++j) t += red[j][k];
        u[k] = 1.0f / ((float)K * t);
    }
}

static size_t ls_sinkhorn_floats(long batch, int K, int mats) {
    const long blocks = (batch + LS_SK_ROWS - 1) / LS_SK_ROWS;
    return (size_t)mats * (((size_t)batch * K + 63) / 64 * 64 + (size_t)blocks * K + 64);
}
struct LsSkBufs { float *E, *part, *u; int blocks; };
static LsSkBufs ls_sinkhorn_bufs(float* workspace, long batch, int K, int mats) {
    LsSkBufs b;
    b.blocks = (int)((batch + LS_SK_ROWS - 1) / LS_SK_ROWS);
    b.E = workspace;
    b.part = b.E + (size_t)mats * (((size_t)batch * K + 63) / 64 * 64);
    b.u = b.part + (size_t)mats * b.blocks * K;
    return b;
}

extern "C" int lsim_sinkhorn_workspace(long batch, int K, size_t* bytes) {
    if (!bytes || batch <= 0 || K <= 0 || K > 64) return LSIM_E_INVALID;
    *bytes = ls_sinkhorn_floats(batch, K, 1) * sizeof(float);
    return LSIM_OK;
}

// the column / row rescaling rounds after `first` has filled E and its partial sums: leaves the final u (E and u define the result)
template <int KMAX> static void ls_sinkhorn_rounds(const LsSkBufs& w, long batch, int K, int mats, int iters, hipStream_t s) {
    const dim3 grid(w.blocks, mats), one(1, mats);
    hipLaunchKernelGGL(lsim_k_sinkhorn_scale, one, dim3(1024), 0, s, (const float*)w.part, w.blocks, K, w.u);
    for (int it = 1; it < iters; ++it) {
        hipLaunchKernelGGL(lsim_k_sinkhorn_mid<KMAX>, grid, dim3(256), 0, s, (const float*)w.E, (const float*)w.u, batch, K, w.part);
        hipLaunchKernelGGL(lsim_k_sinkhorn_scale, one, dim3(1024), 0, s, (const float*)w.part, w.blocks, K, w.u);
    }
}

template <int KMAX> static void ls_sinkhorn_run(const float* scores, long lds, long batch, int K, float eps, int iters, float* out, float* workspace,
                                                hipStream_t s) {
    const LsSkBufs w = ls_sinkhorn_bufs(workspace, batch, K, 1);
    const bool vec_in = (K & 3) == 0 && (lds & 3) == 0 && ((uintptr_t)scores & 15) == 0;
    hipLaunchKernelGGL(lsim_k_sinkhorn_first<KMAX>, dim3(w.blocks, 1), dim3(256), 0, s, scores, lds, 0L, batch, K, 1.0f / eps, vec_in, w.E, w.part);
    ls_sinkhorn_rounds<KMAX>(w, batch, K, 1, iters, s);
    hipLaunchKernelGGL(lsim_k_sinkhorn_last<KMAX>, dim3(w.blocks, 1), dim3(256), 0, s, (const float*)w.E, (const float*)w.u, batch, K, out);
}

extern "C" int lsim_sinkhorn(const float* scores, int64_t lds, int64_t batch, int K, float eps, int iters, float* out,
                             void* workspace, size_t workspace_bytes, void* stream) {
    size_t need;
    int rc = lsim_sinkhorn_workspace(batch, K, &need);
    if (rc != LSIM_OK) return rc;
    if (!scores || !out || !workspace || workspace_bytes < need || iters < 1 || lds < K || eps <= 0.0f || ((uintptr_t)workspace & 15) != 0 ||
        ((uintptr_t)out & 15) != 0)
        return LSIM_E_INVALID;
    if (K <= 32) ls_sinkhorn_run<32>(scores, (long)lds, (long)batch, K, eps, iters, out, (float*)workspace, (hipStream_t)stream);
    else ls_sinkhorn_run<64>(scores, (long)lds, (long)batch, K, eps, iters, out, (float*)workspace, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// ---- estimator loss head of HIMEstimator.update (HES:76-108), forward and backward around the two encoder outputs:
//   pred_vel, l_s = enc[:, :3], enc[:, 3:];   z_s = l_s / max(|l_s|, 1e-12);   z_t = tgt / max(|tgt|, 1e-12)
//   S_s = z_s P^T, S_t = z_t P^T (P: K row-normalised prototypes);   q_s, q_t = sinkhorn(S_s), sinkhorn(S_t)       (no gradient)
//   swap = -0.5 mean_{b,k} (q_s log_softmax(S_t / T) + q_t log_softmax(S_s / T));   est = mean_{b,c} (pred_vel - vel)^2
// and d (est + swap) / d enc [B, 3 + D], / d tgt [B, D], / d P [K, D].  Launches: scores + first Sinkhorn pass of both matrices (1),
// the remaining Sinkhorn rounds of both together (2 iters - 1), loss + row gradients with the last Sinkhorn step folded in (1), finish
// (1), prototype gradient through lsim_linear_wgrad's single-wave kernel (2): 10 instead of the ~75 torch kernels of the same
// arithmetic.
// Scores and loss: FOUR LANES (one DPP quad) per sample row, lane q of the quad owns prototypes [q K/4, (q + 1) K/4).  One thread per
// row (rounds 2-4) left 1.5 waves per SIMD on the chip, every row load and store strided by the row length, 32 + 32 row values in
// registers: 48 + 49 us per call at the minibatch of 102 400.  With the quad a wave reads and writes 16 CONSECUTIVE rows (2 KB
// contiguous per instruction group), row reductions are two quad_perm DPP steps, and 4x the waves hide the latency.
#define LS_EST_ROWS 64                      // sample rows per block of 256 threads
#if !defined(LS_EST_WAVES_PER_EU)
#define LS_EST_WAVES_PER_EU 4
#endif
template <int CTRL> __device__ __forceinline__ float ls_est_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float ls_quad_x1(float v) { return ls_est_dpp<0xB1>(v); }     // quad_perm [1, 0, 3, 2]: from lane ^ 1
__device__ __forceinline__ float ls_quad_x2(float v) { return ls_est_dpp<0x4E>(v); }     // quad_perm [2, 3, 0, 1]: from lane ^ 2
__device__ __forceinline__ float ls_quad_sum(float v) { v += ls_quad_x1(v); v += ls_quad_x2(v); return v; }   // same bits on all four lanes
__device__ __forceinline__ float ls_quad_max(float v) { v = fmaxf(v, ls_quad_x1(v)); v = fmaxf(v, ls_quad_x2(v)); return v; }

// prototypes in LDS: row k at (k * DMAX + (k / KQ) * 4) floats -- the four lanes of a quad read rows KQ apart with ds_read_b128, and
// KQ * DMAX is a multiple of the 64 banks; one float4 of padding per quarter puts the four addresses on different bank quads
#define LS_EST_P_FLOATS(KMAX, DMAX) ((KMAX) * (DMAX) + 16)
template <int KMAX, int DMAX>
__device__ __forceinline__ void ls_load_proto_lds(const float* __restrict__ proto, int K, int D, float* __restrict__ P) {
    constexpr int KQ = KMAX / 4;
    for (int i = threadIdx.x; i < KMAX * DMAX; i += blockDim.x) {
        const int k = i / DMAX, d = i - k * DMAX;
        P[i + (k / KQ) * 4] = (k < K && d < D) ? proto[k * D + d] : 0.0f;
    }
}

// a lane's KQ consecutive values of a row segment p[0 .. KQ) that starts at prototype k0: 16-byte accesses when K % 4 == 0
template <int KQ>
__device__ __forceinline__ void ls_seg_load(const float* __restrict__ p, int k0, int K, bool vec, float fill, float (&v)[KQ]) {
    if (vec) {
#pragma unroll
        for (int j = 0; j < KQ; j += 4) {
            const bool in = k0 + j < K;
            const float4 t = *(const float4*)(p + (in ? j : -k0));         // clamped to the row start: always inside the row
            v[j] = in ? t.x : fill; v[j + 1] = in ? t.y : fill; v[j + 2] = in ? t.z : fill; v[j + 3] = in ? t.w : fill;
        }
    } else {
#pragma unroll
        for (int j = 0; j < KQ; ++j) { const bool in = k0 + j < K; const float t = p[in ? j : -k0]; v[j] = in ? t : fill; }
    }
}
template <int KQ>
__device__ __forceinline__ void ls_seg_store(float* __restrict__ p, int k0, int K, bool vec, const float (&v)[KQ]) {
    if (vec) {
#pragma unroll
        for (int j = 0; j < KQ; j += 4) if (k0 + j < K) *(float4*)(p + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < KQ; ++j) if (k0 + j < K) p[j] = v[j];
    }
}

// partial column sums of a block's LS_EST_ROWS rows on the quad layout (lane q of a quad: columns [k0, k0 + K/4) of its row): the four quads
// of a 16-lane DPP row hold the same columns of four rows -> two row rotations; then one LDS line per 16-lane group, added in a fixed order
template <int KMAX>
__device__ __forceinline__ void ls_est_group_colsum(const float (&e)[KMAX / 4], float (*__restrict__ gs)[KMAX] /* [GROUPS][KMAX] */, int k0) {
    const int lane = threadIdx.x & 63, group = threadIdx.x >> 4;
#pragma unroll
    for (int j = 0; j < KMAX / 4; ++j) {
        float t = e[j];
        t += ls_est_dpp<0x120 + 8>(t);          // row_ror:8
        t += ls_est_dpp<0x120 + 4>(t);          // row_ror:4
        if ((lane & 12) == 0) gs[group][k0 + j] = t;
    }
}
template <int KMAX>
__device__ __forceinline__ void ls_est_block_colsum(const float (*__restrict__ gsum)[LS_EST_ROWS / 4][KMAX] /* [2][GROUPS][KMAX] */,
                                                    float* __restrict__ part /* [2][gridDim.x][K] */, int K) {
    __syncthreads();
    if (threadIdx.x < 2 * KMAX) {
        const int m = threadIdx.x / KMAX, k = threadIdx.x % KMAX;
        if (k < K) {
            float a = 0.0f;
#pragma unroll
            for (int g = 0; g < LS_EST_ROWS / 4; ++g) a += gsum[m][g][k];
            part[((long)m * gridDim.x + blockIdx.x) * K + k] = a;
        }
    }
}

template <int KMAX, int DMAX>
__global__ __launch_bounds__(4 * LS_EST_ROWS) __attribute__((amdgpu_waves_per_eu(LS_EST_WAVES_PER_EU, 8))) void lsim_k_est_scores(const float* __restrict__ enc, long ld_o, const float* __restrict__ tgt, long ld_t,
                                                         const float* __restrict__ proto, long batch, int D, int K, float inv_eps,
                                                         float* __restrict__ z /* [2][B][D] */, float* __restrict__ inv_n /* [2][B] */,
                                                         float* __restrict__ S /* [2][B][K] */, float* __restrict__ part /* [2][gridDim.x][K] */) {
    constexpr int KQ = KMAX / 4, DQ = DMAX / 4, GROUPS = LS_EST_ROWS / 4;      // GROUPS: 16-lane DPP rows per block (4 sample rows each)
    __shared__ __attribute__((aligned(16))) float P[LS_EST_P_FLOATS(KMAX, DMAX)];
    __shared__ float gsum[2][GROUPS][KMAX];
    ls_load_proto_lds<KMAX, DMAX>(proto, K, D, P);
    __syncthreads();
    const int q = threadIdx.x & 3, k0 = q * KQ;
    const long b = (long)blockIdx.x * LS_EST_ROWS + (threadIdx.x >> 2);
    const bool live = b < batch, vec = (K & 3) == 0, zvec = (D & 3) == 0;
    const long bb = live ? b : batch - 1;           // rows past the batch compute on the last row (all lanes stay in the DPP steps), store nothing
    float zz[2][DMAX];
#pragma unroll
    for (int m = 0; m < 2; ++m) {                   // student, target
        const float* src = m ? tgt + bb * ld_t : enc + bb * ld_o + 3;
        float ss = 0.0f;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) { const float t = src[d < D ? d : 0]; zz[m][d] = d < D ? t : 0.0f; ss = fmaf(zz[m][d], zz[m][d], ss); }
        const float n = sqrtf(ss), inv = 1.0f / fmaxf(n, 1e-12f);
        if (live && q == 0) inv_n[(long)m * batch + b] = n < 1e-12f ? -inv : inv;          // sign flags the clamped branch of F.normalize
#pragma unroll
        for (int d = 0; d < DMAX; ++d) zz[m][d] *= inv;
        // lane q writes latents [q DQ, (q + 1) DQ) of the normalised row
        float o[DQ];
#pragma unroll
        for (int i = 0; i < DQ; ++i) o[i] = q == 0 ? zz[m][i] : q == 1 ? zz[m][DQ + i] : q == 2 ? zz[m][2 * DQ + i] : zz[m][3 * DQ + i];
        float* zdst = z + ((long)m * batch + bb) * D + q * DQ;
        if (live) ls_seg_store<DQ>(zdst, q * DQ, D, zvec, o);
    }
    float sc[2][KQ];
    const float* pq = P + q * (KQ * DMAX + 4);
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int d = 0; d < DMAX; d += 4) {
            const float4 pr = *(const float4*)(pq + j * DMAX + d);
            a0 = fmaf(zz[0][d], pr.x, a0); a0 = fmaf(zz[0][d + 1], pr.y, a0); a0 = fmaf(zz[0][d + 2], pr.z, a0); a0 = fmaf(zz[0][d + 3], pr.w, a0);
            a1 = fmaf(zz[1][d], pr.x, a1); a1 = fmaf(zz[1][d + 1], pr.y, a1); a1 = fmaf(zz[1][d + 2], pr.z, a1); a1 = fmaf(zz[1][d + 3], pr.w, a1);
        }
        asm volatile("" : "+v"(a0), "+v"(a1));      // both dot products finish HERE: left free, the target's half sinks below the student's stores
        sc[0][j] = a0; sc[1][j] = a1;                  // and all K/4 prototype rows wait for it in 128 registers (174 in all, 2 waves per SIMD)
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (live) ls_seg_store<KQ>(S + ((long)m * batch + bb) * K + k0, k0, K, vec, sc[m]);
        // E = exp(S / eps) is not stored: the Sinkhorn rounds and the loss kernel form it again from S with this same expression (26 MB less
        // to write here and 26 MB less to read in each of them)
#pragma unroll
        for (int j = 0; j < KQ; ++j) sc[m][j] = (live && k0 + j < K) ? expf(sc[m][j] * inv_eps) : 0.0f;
        ls_est_group_colsum<KMAX>(sc[m], gsum[m], k0);
    }
    ls_est_block_colsum<KMAX>(gsum, part, K);
}

// the Sinkhorn round between two rescalings, on the quad layout: v[b] = 1 / (B * sum_k E[b, k] u[k]), partial column sums of E * v
template <int KMAX>
__global__ __launch_bounds__(4 * LS_EST_ROWS) void lsim_k_est_mid(const float* __restrict__ S, const float* __restrict__ u, long batch, int K, float inv_eps,
                                                                  float* __restrict__ part /* [2][gridDim.x][K] */) {
    constexpr int KQ = KMAX / 4, GROUPS = LS_EST_ROWS / 4;
    __shared__ float us[2][KMAX];
    __shared__ float gsum[2][GROUPS][KMAX];
    if (threadIdx.x < 2 * KMAX) { const int m = threadIdx.x / KMAX, k = threadIdx.x % KMAX; us[m][k] = k < K ? u[m * 64 + k] : 0.0f; }
    __syncthreads();
    const int q = threadIdx.x & 3, k0 = q * KQ;
    const long b = (long)blockIdx.x * LS_EST_ROWS + (threadIdx.x >> 2);
    const bool live = b < batch, vec = (K & 3) == 0;
    const long bb = live ? b : batch - 1;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        float v[KQ];
        ls_seg_load<KQ>(S + ((long)m * batch + bb) * K + k0, k0, K, vec, 0.0f, v);
        float t = 0.0f;
#pragma unroll
        for (int j = 0; j < KQ; ++j) { v[j] = k0 + j < K ? expf(v[j] * inv_eps) : 0.0f; t = fmaf(v[j], us[m][k0 + j], t); }
        const float vb = live ? 1.0f / ((float)batch * ls_quad_sum(t)) : 0.0f;
#pragma unroll
        for (int j = 0; j < KQ; ++j) v[j] *= vb;
        ls_est_group_colsum<KMAX>(v, gsum[m], k0);
    }
    ls_est_block_colsum<KMAX>(gsum, part, K);
}

static __device__ __forceinline__ float ls_wave_sum64(float t) {
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    return t;
}

template <int KMAX, int DMAX>
__global__ __launch_bounds__(4 * LS_EST_ROWS) __attribute__((amdgpu_waves_per_eu(KMAX * DMAX > 512 ? 2 : LS_EST_WAVES_PER_EU, 8))) void lsim_k_est_loss(const float* __restrict__ enc, long ld_o, const float* __restrict__ vel, long ld_v,
                                                       const float* __restrict__ proto, float* __restrict__ S /* in: scores, out: d loss / d scores */,
                                                       const float* __restrict__ u, const float* __restrict__ z,
                                                       const float* __restrict__ inv_n, long batch, int D, int K, float inv_T, float inv_eps,
                                                       float* __restrict__ d_enc /* [B][3 + D] */, float* __restrict__ d_tgt /* [B][D] */,
                                                       float* __restrict__ part /* [gridDim.x][2] */) {
    constexpr int KQ = KMAX / 4, DQ = DMAX / 4, DH = DMAX / 2, WAVES = 4 * LS_EST_ROWS / 64;
    __shared__ __attribute__((aligned(16))) float P[LS_EST_P_FLOATS(KMAX, DMAX)];
    __shared__ float us[2][KMAX];
    __shared__ float red[WAVES][2];
    ls_load_proto_lds<KMAX, DMAX>(proto, K, D, P);
    if (threadIdx.x < 2 * KMAX) { const int m = threadIdx.x / KMAX, k = threadIdx.x % KMAX; us[m][k] = k < K ? u[m * 64 + k] : 0.0f; }
    __syncthreads();
    const int q = threadIdx.x & 3, k0 = q * KQ;
    const long b = (long)blockIdx.x * LS_EST_ROWS + (threadIdx.x >> 2);
    const bool live = b < batch, vec = (K & 3) == 0;
    const long bb = live ? b : batch - 1;
    const float c = -0.5f / ((float)batch * (float)K), ce = 2.0f / (3.0f * (float)batch);
    const float* pq = P + q * (KQ * DMAX + 4);
    const int d0 = (q & 1) * DH + (q >> 1) * DQ;    // the latents this lane ends up with after the two exchange steps below
    float est_acc = 0.0f, swap_acc = 0.0f;
    // m = 0: the target's assignment q_t weights the student's log-softmax -> gradient to the student scores; m = 1 the other way
    float s01[2][KQ];                               // the row's scores of both matrices: each is the other's assignment input, and d loss / d score replaces them below
#pragma unroll
    for (int m = 0; m < 2; ++m) ls_seg_load<KQ>(S + ((long)m * batch + bb) * K + k0, k0, K, vec, -INFINITY, s01[m]);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        float qv[KQ], x[KQ];
        float* srow = S + ((long)m * batch + bb) * K + k0;
        float qsum = 0.0f;
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            x[j] = s01[m][j];
            qv[j] = k0 + j < K ? expf(s01[1 - m][j] * inv_eps) * us[1 - m][k0 + j] : 0.0f;      // E = exp(S / eps) as the score kernel formed it
            qsum += qv[j];
        }
        const float qn = 1.0f / ls_quad_sum(qsum);              // last Sinkhorn step: B E u v = E u / sum_k E u
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < KQ; ++j) { x[j] *= inv_T; mx = fmaxf(mx, x[j]); }
        mx = ls_quad_max(mx);
        float se = 0.0f;
#pragma unroll
        for (int j = 0; j < KQ; ++j) se += expf(x[j] - mx);
        const float lse = mx + logf(ls_quad_sum(se));
        float gs = 0.0f, dot = 0.0f;
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            qv[j] *= qn;
            const float lp = k0 + j < K ? x[j] - lse : 0.0f;
            dot = fmaf(qv[j], lp, dot);
            gs += c * qv[j];
            x[j] = lp;
        }
        gs = ls_quad_sum(gs);
        if (live) swap_acc += dot;                              // the quad's four partial dots meet in the wave sum
#pragma unroll
        for (int j = 0; j < KQ; ++j) x[j] = k0 + j < K ? (c * qv[j] - expf(x[j]) * gs) * inv_T : 0.0f;      // d loss / d score
        if (live) ls_seg_store<KQ>(srow, k0, K, vec, x);
        // d z = dS P: every lane over its K/4 prototypes, then the quad adds up AND splits the D latents (lane ^ 1: halves, lane ^ 2: quarters)
        float dz[DMAX];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) dz[d] = 0.0f;
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
#pragma unroll
            for (int d = 0; d < DMAX; d += 4) {
                const float4 pr = *(const float4*)(pq + j * DMAX + d);
                dz[d] = fmaf(x[j], pr.x, dz[d]); dz[d + 1] = fmaf(x[j], pr.y, dz[d + 1]); dz[d + 2] = fmaf(x[j], pr.z, dz[d + 2]); dz[d + 3] = fmaf(x[j], pr.w, dz[d + 3]);
            }
        }
        float h[DH], g[DQ];
#pragma unroll
        for (int i = 0; i < DH; ++i) {
            const float give = (q & 1) ? dz[i] : dz[DH + i], keep = (q & 1) ? dz[DH + i] : dz[i];
            h[i] = keep + ls_quad_x1(give);
        }
#pragma unroll
        for (int i = 0; i < DQ; ++i) {
            const float give = (q & 2) ? h[i] : h[DQ + i], keep = (q & 2) ? h[DQ + i] : h[i];
            g[i] = keep + ls_quad_x2(give);
        }
        // F.normalize backward on latents [d0, d0 + DQ)
        const float* zrow = z + ((long)m * batch + bb) * D;
        float zz[DQ], zd = 0.0f;
#pragma unroll
        for (int i = 0; i < DQ; ++i) { const bool in = d0 + i < D; const float t = zrow[in ? d0 + i : 0]; zz[i] = in ? t : 0.0f; zd = fmaf(zz[i], g[i], zd); }
        zd = ls_quad_sum(zd);
        const float inv = inv_n[(long)m * batch + bb];
        float* dst = m ? d_tgt + bb * D : d_enc + bb * (3 + D) + 3;
#pragma unroll
        for (int i = 0; i < DQ; ++i)
            if (live && d0 + i < D) dst[d0 + i] = inv < 0.0f ? g[i] * -inv : (g[i] - zz[i] * zd) * inv;      // clamped branch: plain scale
    }
    if (live && q < 3) {
        const float e = enc[b * ld_o + q] - vel[b * ld_v + q];
        est_acc = e * e;
        d_enc[b * (3 + D) + q] = ce * e;
    }
    est_acc = ls_wave_sum64(est_acc);
    swap_acc = ls_wave_sum64(swap_acc);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[w][0] = est_acc; red[w][1] = swap_acc; }
    __syncthreads();
    if (threadIdx.x < 2) {
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) a += red[i][threadIdx.x];
        part[(size_t)blockIdx.x * 2 + threadIdx.x] = a;
    }
}

// the last launch of the loss head, two jobs side by side:
//   block 0             losses[0] = est, [1] = swap, [2] = est + swap from the loss kernel's per-block sums
//   blocks 1 ..         d P = sum of the prototype weight-gradient partials [num_partials][count]: 64 outputs (16 float4 columns) x 64 slices
//                       per block, eight rows in flight per thread -- the general lsim_k_wgrad_reduce (16 slices, four rows in flight) needed
//                       12.8 us for these 1024 partial rows of 512 outputs, as a launch of its own behind a 5 us finish launch
// every sum in a fixed order
__global__ __launch_bounds__(1024) void lsim_k_est_tail(const float* __restrict__ part, int blocks, long batch, int K, float* __restrict__ losses,
                                                        const float* __restrict__ wpart, int num_partials, int count, float* __restrict__ d_proto) {
    __shared__ float4 red4[64][17];
    if (blockIdx.x == 0) {
        float (*red)[2] = (float (*)[2])&red4[0][0];           // 1024 x 2 floats
        float a = 0.0f, s = 0.0f;
        for (int i = threadIdx.x; i < blocks; i += 1024) { a += part[(size_t)i * 2]; s += part[(size_t)i * 2 + 1]; }
        red[threadIdx.x][0] = a; red[threadIdx.x][1] = s;
        __syncthreads();
        for (int st = 512; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) { red[threadIdx.x][0] += red[threadIdx.x + st][0]; red[threadIdx.x][1] += red[threadIdx.x + st][1]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float est = red[0][0] / (3.0f * (float)batch), swap = -0.5f * red[0][1] / ((float)batch * (float)K);
            losses[0] = est; losses[1] = swap; losses[2] = est + swap;
        }
        return;
    }
    const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int o = ((int)blockIdx.x - 1) * 64 + 4 * ol;         // count % 4 == 0 (checked by the caller)
    auto add = [](float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    float4 s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = make_float4(0, 0, 0, 0);
    if (o < count) {
        int w = sl;
        for (; w + 7 * 64 < num_partials; w += 8 * 64) {
#pragma unroll
            for (int j = 0; j < 8; ++j) add(s[j], *(const float4*)(wpart + (size_t)(w + j * 64) * count + o));
        }
        for (; w < num_partials; w += 64) add(s[0], *(const float4*)(wpart + (size_t)w * count + o));
    }
    add(s[0], s[1]); add(s[2], s[3]); add(s[4], s[5]); add(s[6], s[7]); add(s[0], s[2]); add(s[4], s[6]); add(s[0], s[4]);
    red4[sl][ol] = s[0];
    __syncthreads();
    if (sl == 0 && o < count) {
        float4 t = make_float4(0, 0, 0, 0);
        for (int j = 0; j < 64; ++j) add(t, red4[j][ol]);
        d_proto[o] = t.x; d_proto[o + 1] = t.y; d_proto[o + 2] = t.z; d_proto[o + 3] = t.w;
    }
}

struct LsEstPlan { size_t z, inv_n, S, sk, part, wg, total; };
static int ls_est_plan(long batch, int D, int K, LsEstPlan* p) {
    if (batch <= 0 || D <= 0 || D > 32 || K <= 0 || K > 64) return LSIM_E_INVALID;
    size_t wg_bytes; int np;
    int rc = lsim_linear_wgrad_workspace(2 * batch, D, K, &wg_bytes, &np);
    if (rc != LSIM_OK) return rc;
    const size_t blocks = (size_t)((batch + LS_EST_ROWS - 1) / LS_EST_ROWS);      // of the score and loss kernels; the Sinkhorn rounds use LS_SK_ROWS
    size_t o = 0;
    auto take = [&o](size_t floats) { const size_t at = o; o += (floats + 63) & ~(size_t)63; return at; };    // 256-byte aligned pieces
    p->z = take(2 * (size_t)batch * D);
    p->inv_n = take(2 * (size_t)batch);
    p->S = take(2 * (size_t)batch * K);
    p->sk = take(2 * blocks * K + 128);                                                     // partial column sums of both matrices, u [2][64]
    p->part = take(blocks * 2);
    p->wg = take((wg_bytes + 3) / 4);
    p->total = o;
    return LSIM_OK;
}

extern "C" int lsim_estimator_loss_workspace(int64_t batch, int latent, int K, size_t* bytes) {
    if (!bytes) return LSIM_E_INVALID;
    LsEstPlan p;
    int rc = ls_est_plan((long)batch, latent, K, &p);
    if (rc != LSIM_OK) return rc;
    *bytes = p.total * sizeof(float);
    return LSIM_OK;
}

struct LsEstArgs {
    const float *enc, *tgt, *proto, *vel;
    long ld_enc, ld_tgt, ld_vel, batch;
    int D, K, iters;
    float inv_T, eps;
    float *losses, *g_enc, *g_tgt;
};
template <int KMAX, int DMAX> static void ls_est_launch(const LsEstArgs& a, float* ws, const LsEstPlan& p, hipStream_t s) {
    const int eb = (int)((a.batch + LS_EST_ROWS - 1) / LS_EST_ROWS);
    float *part = ws + p.sk, *u = part + (size_t)2 * eb * a.K;
    const dim3 grid(eb), one(1, 2), threads(4 * LS_EST_ROWS);
    const float inv_eps = 1.0f / a.eps;
    hipLaunchKernelGGL((lsim_k_est_scores<KMAX, DMAX>), grid, threads, 0, s, a.enc, a.ld_enc, a.tgt, a.ld_tgt, a.proto, a.batch, a.D, a.K, inv_eps, ws + p.z,
                       ws + p.inv_n, ws + p.S, part);
    hipLaunchKernelGGL(lsim_k_sinkhorn_scale, one, dim3(1024), 0, s, (const float*)part, eb, a.K, u);
    for (int it = 1; it < a.iters; ++it) {
        hipLaunchKernelGGL((lsim_k_est_mid<KMAX>), grid, threads, 0, s, (const float*)(ws + p.S), (const float*)u, a.batch, a.K, inv_eps, part);
        hipLaunchKernelGGL(lsim_k_sinkhorn_scale, one, dim3(1024), 0, s, (const float*)part, eb, a.K, u);
    }
    hipLaunchKernelGGL((lsim_k_est_loss<KMAX, DMAX>), grid, threads, 0, s, a.enc, a.ld_enc, a.vel, a.ld_vel, a.proto, ws + p.S, (const float*)u,
                       (const float*)(ws + p.z), (const float*)(ws + p.inv_n), a.batch, a.D, a.K, a.inv_T, inv_eps, a.g_enc, a.g_tgt, ws + p.part);
}

extern "C" int lsim_estimator_loss(const float* enc_out, int64_t ld_enc, const float* tgt_out, int64_t ld_tgt, const float* proto, const float* vel,
                                   int64_t ld_vel, int64_t batch, int latent, int K, float temperature, float sinkhorn_eps, int sinkhorn_iters,
                                   float* losses3, float* grad_enc, float* grad_tgt, float* grad_proto, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    LsEstPlan p;
    int rc = ls_est_plan((long)batch, latent, K, &p);
    if (rc != LSIM_OK) return rc;
    if (!enc_out || !tgt_out || !proto || !vel || !losses3 || !grad_enc || !grad_tgt || !grad_proto || !workspace ||
        workspace_bytes < p.total * sizeof(float) || ld_enc < 3 + latent || ld_tgt < latent || ld_vel < 3 || temperature <= 0.0f ||
        sinkhorn_eps <= 0.0f || sinkhorn_iters < 1 || ((uintptr_t)workspace & 15) != 0)
        return LSIM_E_INVALID;
    float* ws = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    LsEstArgs a = {enc_out, tgt_out, proto, vel, (long)ld_enc, (long)ld_tgt, (long)ld_vel, (long)batch, latent, K, sinkhorn_iters,
                   1.0f / temperature, sinkhorn_eps, losses3, grad_enc, grad_tgt};
    if (K <= 32 && latent <= 16) ls_est_launch<32, 16>(a, ws, p, s);
    else if (K <= 32) ls_est_launch<32, 32>(a, ws, p, s);
    else if (latent <= 16) ls_est_launch<64, 16>(a, ws, p, s);
    else ls_est_launch<64, 32>(a, ws, p, s);
    // d P[k][d] = sum over both halves of dS[b][k] z[b][d]: a Linear weight gradient with x = z [2B, D], g = dS [2B, K]; its partial results
    // are added up by the tail launch, next to the loss sums
    size_t wg_bytes; int np;
    lsim_linear_wgrad_workspace(2 * batch, latent, K, &wg_bytes, &np);
    const int eb = (int)((batch + LS_EST_ROWS - 1) / LS_EST_ROWS);
    lsim_wgrad_pending pend;
    const bool fold = (latent * K) % 4 == 0;
    rc = ls_linear_wgrad_impl(ws + p.z, latent, ws + p.S, K, nullptr, 0, nullptr, 2 * batch, latent, K, grad_proto, nullptr, ws + p.wg, wg_bytes, s, fold ? &pend : nullptr);
    if (rc != LSIM_OK) return rc;
    if (fold && (((uintptr_t)pend.part & 15) != 0 || pend.count != latent * K)) return LSIM_E_INVALID;        // the workspace piece is 256-byte aligned
    hipLaunchKernelGGL(lsim_k_est_tail, dim3(fold ? 1 + (pend.count + 63) / 64 : 1), dim3(1024), 0, s, (const float*)(ws + p.part), eb, (long)batch, K, losses3,
                       fold ? pend.part : nullptr, fold ? pend.num_partials : 0, fold ? pend.count : 0, grad_proto);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// ---- row gather: block = 256 / CPT rows x CPT column lanes (CPT = the row width rounded up to a power of two, at most 256)
__global__ __launch_bounds__(256) void lsim_k_gather_rows(const uint32_t* __restrict__ src, long cols, const long long* __restrict__ index, long n,
                                                          uint32_t* __restrict__ dst, long dst_ld, int cpt_log2) {
    const int cpt = 1 << cpt_log2, rpb = 256 >> cpt_log2;
    const long r = (long)blockIdx.x * rpb + (threadIdx.x >> cpt_log2);
    if (r >= n) return;
    const uint32_t* s = src + (size_t)index[r] * cols;
    uint32_t* d = dst + (size_t)r * dst_ld;
    for (long c = threadIdx.x & (cpt - 1); c < cols; c += cpt) d[c] = s[c];
}
// wide rows (the observation fields: 238 / 270 floats): one WAVE per row, 8-byte elements, four rows of a wave in flight at once, a resident
// grid that strides over the rows.  [The form above gives every wide row a 256-thread block of its own: 409 600 blocks of two 4-byte
// copies each ran at 2.2 TB/s over the ten fields of a shuffle (1.18 ms per update, r05_kernel_stats_train.csv).]
#define LS_GATHER_MAXC 3          // 64-lane chunks of 8-byte elements per row: rows up to 384 floats
__global__ __launch_bounds__(256) void lsim_k_gather_rows_wide(const uint2* __restrict__ src, long cols2, const long long* __restrict__ index, long n,
                                                               uint2* __restrict__ dst, long dst_ld2) {
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long r0 = wave * U; r0 < n; r0 += nwaves * U) {
        uint2 v[U][LS_GATHER_MAXC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = r0 + u < n ? r0 + u : n - 1;
            const uint2* s = src + (size_t)index[r] * cols2;
#pragma unroll
            for (int m = 0; m < LS_GATHER_MAXC; ++m) {
                const long c = lane + 64 * m;
                if (c < cols2) v[u][m] = s[c];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r0 + u >= n) break;
            uint2* d = dst + (size_t)(r0 + u) * dst_ld2;
#pragma unroll
            for (int m = 0; m < LS_GATHER_MAXC; ++m) {
                const long c = lane + 64 * m;
                if (c < cols2) d[c] = v[u][m];
            }
        }
    }
}
extern "C" int lsim_gather_rows_ld(const void* src, int64_t cols, const int64_t* index, int64_t n, void* dst, int64_t dst_ld, void* stream) {
    if (!src || !index || !dst || cols <= 0 || n < 0 || dst_ld < cols) return LSIM_E_INVALID;
    if (n == 0) return LSIM_OK;
    if (cols >= 64 && cols % 2 == 0 && dst_ld % 2 == 0 && cols <= 128 * LS_GATHER_MAXC && (((uintptr_t)src | (uintptr_t)dst) & 7) == 0) {
        const long waves = (n + 3) / 4;
        long blocks = (waves + 3) / 4;
        if (blocks > 2048) blocks = 2048;           // 8 192 waves: 32 per CU
        hipLaunchKernelGGL(lsim_k_gather_rows_wide, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint2*)src, (long)(cols / 2),
                           (const long long*)index, (long)n, (uint2*)dst, (long)(dst_ld / 2));
        return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
    }
    int lg = 0;
    while ((1 << lg) < cols && lg < 8) ++lg;
    const long rpb = 256 >> lg;
    hipLaunchKernelGGL(lsim_k_gather_rows, dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)src, (long)cols,
                       (const long long*)index, (long)n, (uint32_t*)dst, (long)dst_ld, lg);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}
extern "C" int lsim_gather_rows(const void* src, int64_t cols, const int64_t* index, int64_t n, void* dst, void* stream) {
    return lsim_gather_rows_ld(src, cols, index, n, dst, cols, stream);
}

// ---- F.normalize(w, dim=-1) in place for a small matrix: one wave per row, lane = column (+ 64 i): one round trip to memory instead of a
// serial loop of `cols` dependent iterations per thread (13 us for the 32 x 16 prototypes, in front of every estimator loss)
__global__ __launch_bounds__(64) void lsim_k_normalize_rows(float* __restrict__ w, int rows, int cols, float eps) {
    float* p = w + (size_t)blockIdx.x * cols;
    float v[4], ss = 0.0f;                       // rows * cols <= 4096 and rows >= 1: cols <= 4096 -> looped in chunks of 256
    for (int c0 = 0; c0 < cols; c0 += 256) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int c = c0 + threadIdx.x + 64 * i; v[i] = c < cols ? p[c] : 0.0f; ss = fmaf(v[i], v[i], ss); }
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float d = fmaxf(sqrtf(ss), eps);
    if (cols <= 256) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int c = threadIdx.x + 64 * i; if (c < cols) p[c] = v[i] / d; }
    } else {
        for (int c = threadIdx.x; c < cols; c += 64) p[c] = p[c] / d;
    }
}
extern "C" int lsim_normalize_rows(float* w, int rows, int cols, float eps, void* stream) {
    if (!w || rows <= 0 || cols <= 0 || (long)rows * cols > 4096 || !(eps > 0.0f)) return LSIM_E_INVALID;
    hipLaunchKernelGGL(lsim_k_normalize_rows, dim3(rows), dim3(64), 0, (hipStream_t)stream, w, rows, cols, eps);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// ---- clipped-PPO loss of HIMPPO.update (HIMP:136-176), forward and backward in one pass over the minibatch.
//   logp_b   = sum_j -(a - mu)^2 / (2 sigma^2) - log sigma - log sqrt(2 pi)            ratio_b = exp(logp_b - old_logp_b)
//   surrogate = mean_b max(-A_b ratio_b, -A_b clamp(ratio_b, 1 - eps, 1 + eps))
//   value     = mean_b max((v - R)^2, (tv + clamp(v - tv, -eps, eps) - R)^2)            (or mean (R - v)^2 without clipping)
//   entropy   = mean_b sum_j (0.5 + 0.5 log(2 pi) + log sigma)
//   kl        = mean_b sum_j log(sigma / old_sigma + 1e-5) + (old_sigma^2 + (old_mu - mu)^2) / (2 sigma^2) - 0.5      (no gradient)
//   loss      = surrogate + c_v value - c_e entropy
// and d loss / d mu [B, A], d loss / d sigma [B, A], d loss / d v [B] with torch's sub-gradient conventions (maximum: ties split evenly,
// clamp: gradient passes on the closed interval).  One thread per sample; block partial sums are added in a fixed order by a second
// launch.  Replaces ~90 small torch kernels per minibatch (log-prob, entropy, KL, ratio, clamps, maxima, means and their backward).
struct LsPpoLossArgs {
    const float* mu; const float* sigma; const float* value;                 // [B, A], [B, A], [B]
    const float* actions; const float* old_logp; const float* adv; const float* returns; const float* target_values;
    const float* old_mu; const float* old_sigma;
    float* g_mu; float* g_sigma; float* g_value;                             // gradients of the total loss (for grad_output = 1)
    float* partial;                                                          // [blocks][4]: surrogate, value, entropy, kl sums
    long batch; int A;
    float clip, value_coef, entropy_coef; int clipped_value;
    int std_mode;                                                            // 1: sigma is std [A] (row stride 0), g_sigma = per-block column sums [blocks][A] behind the 4 statistics
};

#define LS_PPO_MAX_A 60
__global__ __launch_bounds__(256) void lsim_k_ppo_loss(LsPpoLossArgs a) {
    __shared__ float red[4][4];
    __shared__ float red_gs[4][LS_PPO_MAX_A];
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    float s_sur = 0.0f, s_val = 0.0f, s_ent = 0.0f, s_kl = 0.0f;
    float dS_keep = 0.0f;      // std mode: d surrogate / d logp of this sample for the column sums below (0 past the batch)
    if (b < a.batch) {
        const int A = a.A;
        const float invB = 1.0f / (float)a.batch;
        const float* mu = a.mu + b * A; const float* sg = a.std_mode ? a.sigma : a.sigma + b * A; const float* ac = a.actions + b * A;
        const float* omu = a.old_mu + b * A; const float* osg = a.old_sigma + b * A;
        float logp = 0.0f, ent = 0.0f, kl = 0.0f;
        for (int j = 0; j < A; ++j) {
            const float s = sg[j], d = ac[j] - mu[j], ls = logf(s);
            logp += -(d * d) / (2.0f * s * s) - ls - 0.9189385332046727f;
            ent += 1.4189385332046727f + ls;
            const float dm = omu[j] - mu[j];
            kl += logf(s / osg[j] + 1.0e-5f) + (osg[j] * osg[j] + dm * dm) / (2.0f * s * s) - 0.5f;
        }
        const float adv = a.adv[b];
        const float ratio = expf(logp - a.old_logp[b]);
        const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float s1 = -adv * ratio, s2 = -adv * rc;
        s_sur = fmaxf(s1, s2);
        // d surrogate / d ratio: branch 1 always differentiable; branch 2 only where the clamp passes (closed interval); ties split evenly
        const float in_range = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
        const float w1 = s1 > s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f), w2 = 1.0f - w1;
        const float dS_dlogp = invB * (-adv) * (w1 + w2 * in_range) * ratio;
        // value loss
        const float v = a.value[b], R = a.returns[b];
        float dV_dv;
        if (a.clipped_value) {
            const float tv = a.target_values[b], dv = v - tv;
            const float vc = tv + fminf(fmaxf(dv, -a.clip), a.clip);
            const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
            s_val = fmaxf(l1, l2);
            const float pass = (dv >= -a.clip && dv <= a.clip) ? 1.0f : 0.0f;
            const float u1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f), u2 = 1.0f - u1;
            dV_dv = invB * (u1 * 2.0f * (v - R) + u2 * 2.0f * (vc - R) * pass);
        } else {
            s_val = (R - v) * (R - v);
            dV_dv = invB * 2.0f * (v - R);
        }
        a.g_value[b] = a.value_coef * dV_dv;
        for (int j = 0; j < A; ++j) {
            const float s = sg[j], d = ac[j] - mu[j];
            a.g_mu[b * A + j] = dS_dlogp * d / (s * s);
            if (!a.std_mode) a.g_sigma[b * A + j] = dS_dlogp * (d * d / (s * s * s) - 1.0f / s) - a.entropy_coef * invB / s;
        }
        s_ent = ent; s_kl = kl;
        dS_keep = dS_dlogp;
    }
    if (a.std_mode) {      // block-uniform: column sums of d total / d sigma over the block's samples, the same fixed order as the statistics
        const int A = a.A;
        const float invB = 1.0f / (float)a.batch;
        const bool live = b < a.batch;
        for (int j = 0; j < A; ++j) {
            float x = 0.0f;
            if (live) {
                const float s = a.sigma[j], d = a.actions[b * A + j] - a.mu[b * A + j];
                x = dS_keep * (d * d / (s * s * s) - 1.0f / s) - a.entropy_coef * invB / s;
            }
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0) red_gs[threadIdx.x >> 6][j] = x;
        }
    }
    // block sums in a fixed order: wave shuffle tree, then the four waves
    float v4[4] = {s_sur, s_val, s_ent, s_kl};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = v4[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = x;
    }
    __syncthreads();
    const int stride = a.std_mode ? 4 + a.A : 4;
    if (threadIdx.x < 4) a.partial[(size_t)blockIdx.x * stride + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (a.std_mode && (int)threadIdx.x < a.A)
        a.partial[(size_t)blockIdx.x * stride + 4 + threadIdx.x] = (red_gs[0][threadIdx.x] + red_gs[1][threadIdx.x]) + (red_gs[2][threadIdx.x] + red_gs[3][threadIdx.x]);
}
// std mode: wave 0 forms out[0..4] exactly as lsim_k_ppo_loss_finish does (same order of additions: the statistics do not depend on the mode);
// waves 1-3: g_std[j] = sum over the blocks of their column sums, thread = (column j, one of 3 interleaved parts)
__global__ __launch_bounds__(256) void lsim_k_ppo_loss_finish_std(const float* __restrict__ partial, int blocks, long batch, int A, float value_coef, float entropy_coef,
                                                                 float* __restrict__ out, float* __restrict__ g_std) {
    const int nq = 4 + A;
    __shared__ float m[4];
    __shared__ float acc[3][64];
    if (threadIdx.x < 64) {
        const int k = threadIdx.x & 3, part = threadIdx.x >> 2;      // 16 interleaved partial sums per quantity
        float s = 0.0f;
        for (int i = part; i < blocks; i += 16) s += partial[(size_t)i * nq + k];
        for (int off = 32; off >= 4; off >>= 1) s += __shfl_down(s, off, 64);
        if (threadIdx.x < 4) m[k] = s / (float)batch;
    } else {
        const int j = threadIdx.x & 63, part = (threadIdx.x >> 6) - 1;
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};                      // four partial rows in flight (one at a time: 13.6 us for 400 blocks)
        if (j < A) {
            int i = part;
            for (; i + 9 < blocks; i += 12) {
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += partial[(size_t)(i + 3 * u) * nq + 4 + j];
            }
            for (; i < blocks; i += 3) s[0] += partial[(size_t)i * nq + 4 + j];
        }
        acc[part][j] = (s[0] + s[1]) + (s[2] + s[3]);
    }
    __syncthreads();
    if (threadIdx.x < 4) out[threadIdx.x] = m[threadIdx.x];
    if (threadIdx.x == 0) out[4] = m[0] + value_coef * m[1] - entropy_coef * m[2];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + A) g_std[threadIdx.x - 64] = (acc[0][threadIdx.x - 64] + acc[1][threadIdx.x - 64]) + acc[2][threadIdx.x - 64];
}

// out[0..3] = means of surrogate, value, entropy, kl; out[4] = total loss
__global__ __launch_bounds__(64) void lsim_k_ppo_loss_finish(const float* __restrict__ partial, int blocks, long batch, float value_coef, float entropy_coef,
                                                             float* __restrict__ out) {
    const int k = threadIdx.x & 3, part = threadIdx.x >> 2;      // 16 interleaved partial sums per quantity
    float s = 0.0f;
    for (int i = part; i < blocks; i += 16) s += partial[(size_t)i * 4 + k];
    for (int off = 32; off >= 4; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ float m[4];
    if (threadIdx.x < 4) { m[k] = s / (float)batch; }
    __syncthreads();
    if (threadIdx.x < 4) out[k] = m[k];
    if (threadIdx.x == 0) out[4] = m[0] + value_coef * m[1] - entropy_coef * m[2];
}

extern "C" int lsim_ppo_loss_workspace(long batch, size_t* bytes) {
    if (!bytes || batch <= 0) return LSIM_E_INVALID;
    *bytes = (size_t)((batch + 255) / 256) * 4 * sizeof(float);
    return LSIM_OK;
}

extern "C" int lsim_ppo_loss(const float* mu, const float* sigma, const float* value, const float* actions, const float* old_logp, const float* advantages,
                             const float* returns, const float* target_values, const float* old_mu, const float* old_sigma, int64_t batch, int num_actions,
                             float clip_param, float value_loss_coef, float entropy_coef, int use_clipped_value_loss,
                             float* out5, float* grad_mu, float* grad_sigma, float* grad_value, void* workspace, size_t workspace_bytes, void* stream) {
    size_t need;
    if (lsim_ppo_loss_workspace(batch, &need) != LSIM_OK) return LSIM_E_INVALID;
    if (!mu || !sigma || !value || !actions || !old_logp || !advantages || !returns || !old_mu || !old_sigma || !out5 || !grad_mu || !grad_sigma ||
        !grad_value || !workspace || workspace_bytes < need || num_actions <= 0 || (use_clipped_value_loss && !target_values)) return LSIM_E_INVALID;
    LsPpoLossArgs a;
    a.mu = mu; a.sigma = sigma; a.value = value; a.actions = actions; a.old_logp = old_logp; a.adv = advantages; a.returns = returns;
    a.target_values = target_values; a.old_mu = old_mu; a.old_sigma = old_sigma; a.g_mu = grad_mu; a.g_sigma = grad_sigma; a.g_value = grad_value;
    a.partial = (float*)workspace; a.batch = batch; a.A = num_actions; a.clip = clip_param; a.value_coef = value_loss_coef;
    a.entropy_coef = entropy_coef; a.clipped_value = use_clipped_value_loss; a.std_mode = 0;
    const int blocks = (int)((batch + 255) / 256);
    hipLaunchKernelGGL(lsim_k_ppo_loss, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(lsim_k_ppo_loss_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, blocks, (long)batch, value_loss_coef,
                       entropy_coef, out5);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

extern "C" int lsim_ppo_loss_std_workspace(long batch, int num_actions, size_t* bytes) {
    if (!bytes || batch <= 0 || num_actions <= 0 || num_actions > LS_PPO_MAX_A) return LSIM_E_INVALID;
    *bytes = (size_t)((batch + 255) / 256) * (4 + num_actions) * sizeof(float);
    return LSIM_OK;
}
extern "C" int lsim_ppo_loss_std(const float* mu, const float* std, const float* value, const float* actions, const float* old_logp, const float* advantages,
                                 const float* returns, const float* target_values, const float* old_mu, const float* old_sigma, int64_t batch, int num_actions,
                                 float clip_param, float value_loss_coef, float entropy_coef, int use_clipped_value_loss,
                                 float* out5, float* grad_mu, float* grad_std, float* grad_value, void* workspace, size_t workspace_bytes, void* stream) {
    size_t need;
    if (lsim_ppo_loss_std_workspace(batch, num_actions, &need) != LSIM_OK) return LSIM_E_INVALID;
    if (!mu || !std || !value || !actions || !old_logp || !advantages || !returns || !old_mu || !old_sigma || !out5 || !grad_mu || !grad_std ||
        !grad_value || !workspace || workspace_bytes < need || (use_clipped_value_loss && !target_values)) return LSIM_E_INVALID;
    LsPpoLossArgs a;
    a.mu = mu; a.sigma = std; a.value = value; a.actions = actions; a.old_logp = old_logp; a.adv = advantages; a.returns = returns;
    a.target_values = target_values; a.old_mu = old_mu; a.old_sigma = old_sigma; a.g_mu = grad_mu; a.g_sigma = nullptr; a.g_value = grad_value;
    a.partial = (float*)workspace; a.batch = batch; a.A = num_actions; a.clip = clip_param; a.value_coef = value_loss_coef;
    a.entropy_coef = entropy_coef; a.clipped_value = use_clipped_value_loss; a.std_mode = 1;
    const int blocks = (int)((batch + 255) / 256);
    hipLaunchKernelGGL(lsim_k_ppo_loss, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(lsim_k_ppo_loss_finish_std, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, blocks, (long)batch, num_actions,
                       value_loss_coef, entropy_coef, out5, grad_std);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// ---- adaptive learning rate of HIMPPO (HIMP:144-156) on the device: lr /= 1.5 when the KL estimate exceeds twice the target, lr *= 1.5
// when it is below half of it (and positive), clamped to [lr_min, lr_max].  Keeps the rule out of the host so that the per-minibatch
// .item() of the reference (a pipeline drain, ~4 % of the update) disappears; the optimisers read the same device scalar.
__global__ void lsim_k_adaptive_lr(const float* __restrict__ kl_mean, float desired_kl, float lr_min, float lr_max, float factor, float* __restrict__ lr) {
    const float k = *kl_mean;
    float l = *lr;
    if (k > 2.0f * desired_kl) l = fmaxf(lr_min, l / factor);
    else if (k < 0.5f * desired_kl && k > 0.0f) l = fminf(lr_max, l * factor);
    *lr = l;
}

extern "C" int lsim_adaptive_lr(const float* kl_mean_dev, float desired_kl, float lr_min, float lr_max, float factor, float* lr_dev, void* stream) {
    if (!kl_mean_dev || !lr_dev || desired_kl <= 0.0f || factor <= 1.0f || lr_min <= 0.0f || lr_max < lr_min) return LSIM_E_INVALID;
    hipLaunchKernelGGL(lsim_k_adaptive_lr, dim3(1), dim3(1), 0, (hipStream_t)stream, kl_mean_dev, desired_kl, lr_min, lr_max, factor, lr_dev);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// ---- gradient clipping + Adam in two launches (HIMP:183-184, HES:113-114: clip_grad_norm_ then optimizer.step(); torch's foreach
// clipping and fused Adam need ~12 launches per optimiser step, 24 per minibatch):
//   sumsq : per-tensor-slice partial sums of g^2; step counters += 1  (grid: LS_ADAM_SLICES x tensors)
//   apply : total norm (fixed summation order, every block for itself), clip coefficient min(max_norm / (norm + 1e-6), 1);
//           g <- coef * g (written back, as clip_grad_norm_ does);  m <- lerp(m, g, 1 - b1);  v <- b2 v + (1 - b2) g^2;
//           p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)           (torch.optim.Adam, no amsgrad / weight decay)
// The tensors are the caller's (torch's parameter, .grad and optimizer-state tensors: checkpoints stay torch's); pointers travel by
// value in the kernel arguments, so nothing is staged on the device.
#define LS_ADAM_MAX_TENSORS 48
#define LS_ADAM_SLICES 32
struct LsAdamTable {
    float* p[LS_ADAM_MAX_TENSORS];
    float* g[LS_ADAM_MAX_TENSORS];
    float* m[LS_ADAM_MAX_TENSORS];
    float* v[LS_ADAM_MAX_TENSORS];
    float* step[LS_ADAM_MAX_TENSORS];
    int n[LS_ADAM_MAX_TENSORS];
    float wd[LS_ADAM_MAX_TENSORS];     // L2 weight decay of torch.optim.Adam (grad + wd * param), per tensor
    int count;
    int clip_count;                    // tensors [0, clip_count) form the clipped-norm group (clip_grad_norm_ over a subset of the parameters)
};

__global__ __launch_bounds__(256) void lsim_k_adam_sumsq(LsAdamTable t, float* __restrict__ part /* [count][LS_ADAM_SLICES] */) {
    __shared__ float red[256];
    const int ti = blockIdx.y, n = t.n[ti];
    const float* __restrict__ g = t.g[ti];
    float s = 0.0f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += LS_ADAM_SLICES * 256) { const float x = g[i]; s = fmaf(x, x, s); }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[ti * LS_ADAM_SLICES + blockIdx.x] = red[0];
        if (blockIdx.x == 0) *t.step[ti] += 1.0f;        // this launch reads no step counter; lsim_k_adam_apply reads the incremented ones
    }
}

// The clip coefficient is formed by every block for itself, from the partial sums, in the order lsim_k_adam_finish used (that launch -- 5.7 us
// behind a dependent launch, twice per minibatch -- is gone): thread i adds tensor i's LS_ADAM_SLICES partials, thread 0 the tensors of the
// clipped group.  Block (0, 0) leaves {coefficient, norm, step count} in scal for the caller.
__global__ __launch_bounds__(256) void lsim_k_adam_apply(LsAdamTable t, const float* __restrict__ part, float max_norm, float* __restrict__ scal,
                                                         const float* __restrict__ lr_dev, float lr_host, float b1, float b2, float eps) {
    __shared__ float tot[LS_ADAM_MAX_TENSORS];
    __shared__ float coef_s;
    if ((int)threadIdx.x < t.count) {
        float s = 0.0f;
        for (int k = 0; k < LS_ADAM_SLICES; ++k) s += part[threadIdx.x * LS_ADAM_SLICES + k];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int k = 0; k < t.clip_count; ++k) s += tot[k];
        const float norm = sqrtf(s);
        coef_s = max_norm > 0.0f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
        if (blockIdx.x == 0 && blockIdx.y == 0) { scal[0] = coef_s; scal[1] = norm; scal[2] = *t.step[0]; }
    }
    __syncthreads();
    const int ti = blockIdx.y, n = t.n[ti];
    float* __restrict__ p = t.p[ti];
    float* __restrict__ g = t.g[ti];
    float* __restrict__ m = t.m[ti];
    float* __restrict__ v = t.v[ti];
    // the bias correction uses THIS tensor's step count (already incremented by lsim_k_adam_sumsq), as torch's Adam does: a parameter that
    // received gradients on fewer steps (frozen layer, partially restored state) must not borrow tensor 0's counter
    const float coef = ti < t.clip_count ? coef_s : 1.0f, step = *t.step[ti], lr = lr_dev ? *lr_dev : lr_host, wd = t.wd[ti];
    const float bc1 = 1.0f - powf(b1, step), bc2s = sqrtf(1.0f - powf(b2, step));
    const float step_size = lr / bc1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += LS_ADAM_SLICES * 256) {
        const float gc = g[i] * coef;                                  // what clip_grad_norm_ leaves in .grad
        const float gi = wd != 0.0f ? gc + wd * p[i] : gc;             // grad.add(param, alpha=weight_decay): not written back
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);             // torch's lerp form
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        g[i] = gc; m[i] = mi; v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) / bc2s + eps);
    }
}

extern "C" int lsim_adam_clip_step_workspace(int count, size_t* bytes) {
    if (!bytes || count <= 0 || count > LS_ADAM_MAX_TENSORS) return LSIM_E_INVALID;
    *bytes = ((size_t)count * LS_ADAM_SLICES + 4) * sizeof(float);
    return LSIM_OK;
}

extern "C" int lsim_adam_clip_step_ex(int count, const int64_t* numel, float* const* params, float* const* grads, float* const* exp_avg,
                                      float* const* exp_avg_sq, float* const* steps, const float* weight_decay, int clip_count,
                                      const float* lr_dev, float lr_host, float beta1, float beta2, float eps, float max_grad_norm,
                                      float* grad_norm_out, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int lsim_adam_clip_step(int count, const int64_t* numel, float* const* params, float* const* grads, float* const* exp_avg,
                                   float* const* exp_avg_sq, float* const* steps, const float* lr_dev, float lr_host, float beta1, float beta2,
                                   float eps, float max_grad_norm, float* grad_norm_out, void* workspace, size_t workspace_bytes, void* stream) {
    return lsim_adam_clip_step_ex(count, numel, params, grads, exp_avg, exp_avg_sq, steps, nullptr, count, lr_dev, lr_host, beta1, beta2, eps,
                                  max_grad_norm, grad_norm_out, workspace, workspace_bytes, stream);
}

extern "C" int lsim_adam_clip_step_ex(int count, const int64_t* numel, float* const* params, float* const* grads, float* const* exp_avg,
                                      float* const* exp_avg_sq, float* const* steps, const float* weight_decay, int clip_count,
                                      const float* lr_dev, float lr_host, float beta1, float beta2, float eps, float max_grad_norm,
                                      float* grad_norm_out, void* workspace, size_t workspace_bytes, void* stream) {
    size_t need;
    int rc = lsim_adam_clip_step_workspace(count, &need);
    if (rc != LSIM_OK) return rc;
    if (!numel || !params || !grads || !exp_avg || !exp_avg_sq || !steps || !workspace || workspace_bytes < need || beta1 < 0.0f || beta1 >= 1.0f ||
        beta2 < 0.0f || beta2 >= 1.0f || eps <= 0.0f || clip_count < 0 || clip_count > count)
        return LSIM_E_INVALID;
    LsAdamTable t;
    t.count = count;
    t.clip_count = clip_count;
    for (int i = 0; i < count; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || !steps[i] || numel[i] <= 0 || numel[i] > 0x7fffffffLL) return LSIM_E_INVALID;
        t.p[i] = params[i]; t.g[i] = grads[i]; t.m[i] = exp_avg[i]; t.v[i] = exp_avg_sq[i]; t.step[i] = steps[i]; t.n[i] = (int)numel[i];
        t.wd[i] = weight_decay ? weight_decay[i] : 0.0f;
        if (t.wd[i] < 0.0f) return LSIM_E_INVALID;
    }
    for (int i = count; i < LS_ADAM_MAX_TENSORS; ++i) { t.p[i] = t.g[i] = t.m[i] = t.v[i] = t.step[i] = nullptr; t.n[i] = 0; t.wd[i] = 0.0f; }
    float* part = (float*)workspace;
    float* scal = part + (size_t)count * LS_ADAM_SLICES;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lsim_k_adam_sumsq, dim3(LS_ADAM_SLICES, count), dim3(256), 0, s, t, part);
    hipLaunchKernelGGL(lsim_k_adam_apply, dim3(LS_ADAM_SLICES, count), dim3(256), 0, s, t, (const float*)part, max_grad_norm, scal, lr_dev, lr_host, beta1,
                       beta2, eps);
    if (grad_norm_out && hipMemcpyAsync(grad_norm_out, scal + 1, sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return LSIM_E_HIP;
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// ---- actor input of HIMActorCritic (HAC:136-141): [current one-step observation | estimated velocity | L2-normalised latent] from the
// observation history and the estimator's encoder output (no gradient flows through it): one launch instead of norm + clamp + div + cat
__global__ __launch_bounds__(256) void lsim_k_actor_input(const float* __restrict__ obs, long ld_obs, int n_one, const float* __restrict__ enc, long ld_enc,
                                                          int n_lat, long batch, float* __restrict__ out) {
    const int W = n_one + 3 + n_lat;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= batch * W) return;
    const long row = idx / W;
    const int c = (int)(idx - row * W);
    float v;
    if (c < n_one) v = obs[row * ld_obs + c];
    else if (c < n_one + 3) v = enc[row * ld_enc + (c - n_one)];
    else {
        const float* z = enc + row * ld_enc + 3;
        float ss = 0.0f;
        for (int k = 0; k < n_lat; ++k) ss = fmaf(z[k], z[k], ss);
        v = z[c - n_one - 3] / fmaxf(sqrtf(ss), 1e-12f);              // F.normalize(p=2, eps=1e-12)
    }
    out[idx] = v;
}

// the same for the usual 64-wide input (45 + 3 + 16) with a 16-wide latent: one wavefront per row, lane = column -- no index division per element,
// and the latent's norm is one sum over the 16-lane DPP row that holds it instead of 16 loads in each of 16 threads (60 -> 17 us at 102 400 rows)
__global__ __launch_bounds__(256) void lsim_k_actor_input64(const float* __restrict__ obs, long ld_obs, int n_one, const float* __restrict__ enc, long ld_enc,
                                                            long batch, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= batch) return;
    float v = lane < n_one ? obs[row * ld_obs + lane] : enc[row * ld_enc + (lane - n_one)];
    float ss = lane >= 48 ? v * v : 0.0f;                      // lanes 48..63 = the latent (n_one + 3 == 48)
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
    if (lane >= 48) v = v / fmaxf(sqrtf(ss), 1e-12f);          // F.normalize(p=2, eps=1e-12)
    out[row * 64 + lane] = v;
}

extern "C" int lsim_actor_input(const float* obs, int64_t ld_obs, int num_one_step_obs, const float* enc_out, int64_t ld_enc, int latent,
                                int64_t batch, float* out, void* stream) {
    if (!obs || !enc_out || !out || batch <= 0 || num_one_step_obs <= 0 || latent <= 0 || ld_obs < num_one_step_obs || ld_enc < 3 + latent)
        return LSIM_E_INVALID;
    if (num_one_step_obs == 45 && latent == 16) {
        hipLaunchKernelGGL(lsim_k_actor_input64, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream, obs, (long)ld_obs, num_one_step_obs,
                           enc_out, (long)ld_enc, (long)batch, out);
        return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
    }
    const long total = (long)batch * (num_one_step_obs + 3 + latent);
    hipLaunchKernelGGL(lsim_k_actor_input, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, (long)ld_obs, num_one_step_obs,
                       enc_out, (long)ld_enc, latent, (long)batch, out);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

