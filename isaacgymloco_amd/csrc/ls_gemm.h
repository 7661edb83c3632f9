// ls_gemm.h -- forward of a hidden layer of the learner's MLPs with the activation in the epilogue (include/lsim.h, lsim_linear_elu_forward):
//
//     out[b, n] = elu(sum_k x[b, k] W[n, k] + bias[n])        b < 102 400 minibatch rows, n_out = 64 .. 512, k_in = 64 .. 512
//
// BLAS runs the product at ~90 % of the fp32 MFMA peak, but has no ELU epilogue: the pre-activation is written (210 MB for 512 columns),
// read again by an elementwise kernel and written a second time -- ten such passes per minibatch, 8 % of the GPU time of a training
// iteration (profiles/r05_kernel_stats_train.csv: elu_kernel).  Here the activation is applied to the accumulators and the layer's output
// is written once.
//
// v_mfma_f32_16x16x4_f32 with the WEIGHT rows as the A operand and the SAMPLE rows as the B operand (D[n][m]): a lane then holds four
// CONSECUTIVE output columns of one sample (registers r = 0..3 of D are rows 4 (lane / 16) + r) and the result leaves in 16-byte stores.
// Block = 4 waves, macro tile BM samples x BN features, K in chunks of 32 staged through LDS (the next chunk's global loads are in flight
// during the MFMAs of the current one).  LDS rows have a pitch of 36 floats: the 16 lanes of a ds_read_b128 group
// read 16 rows at the same k, and 36 r mod 64 are 16 different multiples of 4 -- no bank conflicts.  One ds_read_b128 gives a lane k =
// 4 q .. 4 q + 3 of its row (q = lane / 16): MFMA step s of a 16-wide sub-chunk uses component s of both operands, i.e. the k order inside
// the sub-chunk is permuted identically for both -- the sum is over the same 16 products.
#pragma once
#include <hip/hip_runtime.h>

// exp of the ELU epilogue: the hardware exponential (v_exp_f32 of x log2 e: 1 ulp of 2^x, absolute error of elu(x) <= 2e-7 for x <= 0) instead of
// the library expf (a dozen instructions with range reduction): 80 accumulators per wave and tile make the epilogue 5 % (512 -> 256) to 40 %
// (64 -> 512, K = 64: 320 MFMAs per tile) of a tile's instruction stream.  -DLS_FWD_PRECISE_EXP: expf
#if !defined(LS_FWD_NARROW_BK)
#define LS_FWD_NARROW_BK 32          // K chunk of the 64-wide feature tiles (64 measured 2-7 % slower on 256 -> 128 and 128 -> 64, 1.6 x slower on the scalar-load form)
#endif
#if defined(LS_FWD_PRECISE_EXP)
#define LS_FWD_EXP(x) expf(x)
#else
#define LS_FWD_EXP(x) __expf(x)
#endif



// stage one BK-wide chunk of ROWS rows (global, row-major, leading dimension ld) in registers: thread t takes k-quad (t & 7) of rows
// (t >> 3) + 32 i.  VEC: 2 = 16-byte loads (ld % 4 == 0, base 16-byte aligned, K % 4 == 0), 1 = 8-byte (even ld and K), 0 = scalars.
// Rows >= rows_total and k >= K read as zero (clamped addresses, selects).
template <int ROWS, int BK, int VEC>
__device__ __forceinline__ void ls_fwd_fetch(const float* __restrict__ p, long ld, long row0, long rows_total, int k0, int K, float4 (&v)[ROWS * BK / 1024]) {
    constexpr int QPR = BK / 4, RPP = 256 / QPR;               // k-quads per row, rows per pass of the block's 256 threads
    const int c = threadIdx.x % QPR, r = threadIdx.x / QPR;
    const int k = k0 + 4 * c;
#pragma unroll
    for (int i = 0; i < ROWS / RPP; ++i) {
        const long row = row0 + r + RPP * i;
        const bool rok = row < rows_total;
        const float* q = p + (rok ? row : rows_total - 1) * ld;
        if (VEC == 2) {
            const bool ok = rok && k < K;
            const float4 t = *(const float4*)(q + (k < K ? k : 0));
            v[i] = ok ? t : make_float4(0, 0, 0, 0);
        } else if (VEC == 1) {
            const bool ok0 = rok && k < K, ok1 = rok && k + 2 < K;
            const float2 a = *(const float2*)(q + (k < K ? k : 0)), b = *(const float2*)(q + (k + 2 < K ? k + 2 : 0));
            v[i] = make_float4(ok0 ? a.x : 0.0f, ok0 ? a.y : 0.0f, ok1 ? b.x : 0.0f, ok1 ? b.y : 0.0f);
        } else {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const bool ok = rok && k + j < K; const float t = q[k + j < K ? k + j : 0]; e[j] = ok ? t : 0.0f; }
            v[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
}
template <int ROWS, int BK>
__device__ __forceinline__ void ls_fwd_stage(float* __restrict__ tile, const float4 (&v)[ROWS * BK / 1024]) {
    constexpr int QPR = BK / 4, RPP = 256 / QPR;
    const int c = threadIdx.x % QPR, r = threadIdx.x / QPR;
#pragma unroll
    for (int i = 0; i < ROWS / RPP; ++i) *(float4*)(tile + (r + RPP * i) * (BK + 4) + 4 * c) = v[i];
}

// waves 2 (samples) x 2 (features), per wave TM x TN MFMA tiles of 16 x 16: BM = 32 TM samples, BN = 32 TN features per block.
// TM = 5 (160 samples): 102 400 rows = 640 sample tiles; 512 and 128 features then give 2560 and 1280 tiles = 5 and 2.5 per block (the
// host picks the feature tile, see lsim_linear_elu_forward).
// PERSISTENT blocks, two per CU (the launch asks for enough LDS that a third does not fit): a block walks its list of tiles with the
// (tile, chunk) sequence flattened -- the first chunk of the next tile is fetched during the last MFMAs of the current one and is in
// flight during its epilogue, instead of a 2 us fetch with nothing to hide behind at the start of each of the 2560 blocks.
// Workgroups go round-robin to the 8 XCDs (each with its own L2): XCD j takes the j-th eighth of the tile list, feature tile fastest, and
// its 64 blocks stride through it together, so that the blocks sharing a sample tile read it through ONE L2 at about the same time.
// One LDS buffer: the next chunk is fetched into registers before the MFMAs of the current one and stored behind a barrier.
// ACT: the epilogue -- LS_FWD_ACT_ELU: elu(acc + bias); LS_FWD_ACT_MASK: aux[m, n] > 0 ? acc + bias : 0 (the ReLU mask of a saved activation `aux`, leading
// dimension ldaux: the product and torch's threshold_backward in one pass -- lsim_linear_masked_forward); LS_FWD_ACT_NONE: acc + bias
#define LS_FWD_ACT_NONE 0
#define LS_FWD_ACT_ELU 1
#define LS_FWD_ACT_MASK 2
template <int TM, int TN, int BK, int VX, int VW, int ACT>
__global__ __launch_bounds__(256, 2) void lsim_k_linear_fwd(const float* __restrict__ x, long ldx, const float* __restrict__ W, const float* __restrict__ bias,
                                                             long M, int K, int N, float* __restrict__ out, long ldo, int n_tiles, unsigned total_tiles,
                                                             const float* __restrict__ aux, long ldaux) {
    constexpr int BM = 32 * TM, BN = 32 * TN, LS_FWD_PITCH = BK + 4, LS_FWD_BK = BK;
    extern __shared__ __attribute__((aligned(16))) float ls_fwd_lds[];
    float* Xs = ls_fwd_lds;
    float* Ws = ls_fwd_lds + BM * LS_FWD_PITCH;
    const unsigned xcd = blockIdx.x & 7, per_xcd = (total_tiles + 7) >> 3, stride = gridDim.x >> 3;      // gridDim.x is a multiple of 8
    const unsigned last = min((xcd + 1) * per_xcd, total_tiles);
    unsigned tile = xcd * per_xcd + (blockIdx.x >> 3);
    if (tile >= last) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, kq = lane >> 4;
    const int wm = (wave & 1) * (BM / 2), wn = (wave >> 1) * (BN / 2);
    ls_v4f acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (ls_v4f){0.0f, 0.0f, 0.0f, 0.0f};
    float4 px[BM * BK / 1024], pw[BN * BK / 1024];
    const int chunks = (K + LS_FWD_BK - 1) / LS_FWD_BK;
    long m0 = (long)(tile / n_tiles) * BM;
    int n0 = (int)(tile % n_tiles) * BN;
    ls_fwd_fetch<BM, BK, VX>(x, ldx, m0, M, 0, K, px);
    ls_fwd_fetch<BN, BK, VW>(W, (long)K, (long)n0, (long)N, 0, K, pw);
    const float* xs = Xs + (wm + c16) * LS_FWD_PITCH + 4 * kq;
    const float* ws = Ws + (wn + c16) * LS_FWD_PITCH + 4 * kq;
    bool first = true;
    for (;;) {
        const unsigned next_tile = tile + stride;
        const bool more = next_tile < last;
        const long m1 = (long)(next_tile / n_tiles) * BM;
        const int n1 = (int)(next_tile % n_tiles) * BN;
        for (int kc = 0; kc < chunks; ++kc) {
#if defined(LS_FWD_PROBE_NOSTAGE)
            if (first)
#endif
            {
            if (!first) __syncthreads();                        // every wave is done reading the previous chunk
            first = false;
            ls_fwd_stage<BM, BK>(Xs, px);
            ls_fwd_stage<BN, BK>(Ws, pw);
            __syncthreads();
            }
#if !defined(LS_FWD_PROBE_NOFETCH)
            if (kc + 1 < chunks) {
                ls_fwd_fetch<BM, BK, VX>(x, ldx, m0, M, (kc + 1) * LS_FWD_BK, K, px);
                ls_fwd_fetch<BN, BK, VW>(W, (long)K, (long)n0, (long)N, (kc + 1) * LS_FWD_BK, K, pw);
            } else if (more) {
                ls_fwd_fetch<BM, BK, VX>(x, ldx, m1, M, 0, K, px);
                ls_fwd_fetch<BN, BK, VW>(W, (long)K, (long)n1, (long)N, 0, K, pw);
            }
#endif
#if defined(LS_FWD_PROBE_NOMFMA)
            if (K > 100000)
#endif
            {
                // operand fragments of sub-chunk h + 1 are read while the MFMAs of sub-chunk h run (two register sets)
                float4 xf[2][TM], wf[2][TN];
#define LS_FWD_FRAGS(SET, H)                                                                                                        \
                _Pragma("unroll") for (int b = 0; b < TM; ++b) xf[SET][b] = *(const float4*)(xs + b * 16 * LS_FWD_PITCH + 16 * (H));   \
                _Pragma("unroll") for (int a = 0; a < TN; ++a) wf[SET][a] = *(const float4*)(ws + a * 16 * LS_FWD_PITCH + 16 * (H));
                // step s outermost: consecutive MFMAs write different accumulators
#define LS_FWD_STEP(SET, C)                                                                                                         \
                _Pragma("unroll") for (int a = 0; a < TN; ++a)                                                                     \
                    _Pragma("unroll") for (int b = 0; b < TM; ++b)                                                                 \
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[SET][a].C, xf[SET][b].C, acc[a][b], 0, 0, 0);
                LS_FWD_FRAGS(0, 0)
                __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
                for (int h = 0; h < LS_FWD_BK / 16; ++h) {
#if defined(LS_FWD_PROBE_NOLDS)
#define LS_FWD_FRAGS2(SET, H) if (K > 100000) { LS_FWD_FRAGS(SET, H) }
#else
#define LS_FWD_FRAGS2(SET, H) LS_FWD_FRAGS(SET, H)
#endif
                    if (h & 1) {
                        if (h + 1 < LS_FWD_BK / 16) { LS_FWD_FRAGS2(0, h + 1) }
                        LS_FWD_STEP(1, x) LS_FWD_STEP(1, y) LS_FWD_STEP(1, z) LS_FWD_STEP(1, w)
                    } else {
                        if (h + 1 < LS_FWD_BK / 16) { LS_FWD_FRAGS2(1, h + 1) }
                        LS_FWD_STEP(0, x) LS_FWD_STEP(0, y) LS_FWD_STEP(0, z) LS_FWD_STEP(0, w)
                    }
                    // the schedule of this sub-chunk: one fragment read, then its share of the MFMAs (left alone the compiler issues the
                    // reads where their data is needed and waits for them with the MFMA pipe empty)
                    if (h + 1 < LS_FWD_BK / 16) {
#pragma unroll
                        for (int i = 0; i < TM + TN; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                         // DS read
                            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN / (TM + TN), 0);   // MFMA
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN - (TM + TN) * (4 * TM * TN / (TM + TN)), 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
                    }
                }
#undef LS_FWD_STEP
#undef LS_FWD_FRAGS2
#undef LS_FWD_FRAGS
            }
        }
        // epilogue: lane = sample m0 + wm + 16 b + c16, features n0 + wn + 16 a + 4 kq .. + 3
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = n0 + wn + 16 * a + 4 * kq;
            const bool nok = n < N;                                 // N % 4 == 0: the quad is inside or outside
            const float4 bv = (bias && nok) ? make_float4(bias[n], bias[n + 1], bias[n + 2], bias[n + 3]) : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const long m = m0 + wm + 16 * b + c16;
                float e[4] = {acc[a][b][0] + bv.x, acc[a][b][1] + bv.y, acc[a][b][2] + bv.z, acc[a][b][3] + bv.w};
                if (ACT == LS_FWD_ACT_ELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = e[j] > 0.0f ? e[j] : LS_FWD_EXP(e[j]) - 1.0f;  // torch's elu_kernel: exp(x) - 1, not expm1
                }
                if (ACT == LS_FWD_ACT_MASK) {
                    if (nok && m < M) {
                        const float4 mk = *(const float4*)(aux + m * ldaux + n);
                        e[0] = mk.x > 0.0f ? e[0] : 0.0f; e[1] = mk.y > 0.0f ? e[1] : 0.0f; e[2] = mk.z > 0.0f ? e[2] : 0.0f; e[3] = mk.w > 0.0f ? e[3] : 0.0f;
                    }
                }
                if (nok && m < M) *(float4*)(out + m * ldo + n) = make_float4(e[0], e[1], e[2], e[3]);
                acc[a][b] = (ls_v4f){0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
        if (!more) break;
        tile = next_tile; m0 = m1; n0 = n1;
    }
}

static int ls_linear_fwd_supported(long batch, int k_in, int n_out) { return batch > 0 && k_in > 0 && n_out > 0 && n_out % 4 == 0; }

template <int TM, int TN, int BK, int ACT>
static void ls_linear_fwd_launch(const float* x, long ldx, const float* W, const float* bias, long M, int K, int N, float* out, long ldo, hipStream_t s,
                                 const float* aux = nullptr, long ldaux = 0) {
    constexpr int BM = 32 * TM, BN = 32 * TN;
    const int n_tiles = (N + BN - 1) / BN;
    const long m_tiles = (M + BM - 1) / BM;
    const unsigned total = (unsigned)(m_tiles * n_tiles);
    static int cus_of[64] = {0};                                                  // per device ordinal (one process may drive several devices)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cus_of[dev]) { hipDeviceProp_t pr; cus_of[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256; }
    unsigned blocks = (unsigned)(2 * cus_of[dev]) & ~7u;                          // two persistent blocks per CU; the kernel wants a multiple of 8 (block & 7 = XCD)
    if (blocks < 8u) blocks = 8u;
    if (blocks > ((total + 7) & ~7u)) blocks = (total + 7) & ~7u;
    const dim3 grid(blocks), threads(256);
    const int vx = ((ldx % 4 == 0) && (K % 4 == 0) && (((uintptr_t)x & 15) == 0)) ? 2 : ((ldx % 2 == 0) && (K % 2 == 0) && (((uintptr_t)x & 7) == 0)) ? 1 : 0;
    const int vw = ((K % 4 == 0) && (((uintptr_t)W & 15) == 0)) ? 2 : ((K % 2 == 0) && (((uintptr_t)W & 7) == 0)) ? 1 : 0;
    constexpr size_t tile_bytes = (size_t)(BM + BN) * (BK + 4) * sizeof(float);
    constexpr size_t lds = tile_bytes > 56 * 1024 ? tile_bytes : 56 * 1024;       // three blocks would need 168 KB: exactly two per CU
    static_assert(lds <= 64 * 1024, "above 64 KB of dynamic LDS the kernel needs hipFuncAttributeMaxDynamicSharedMemorySize, per device");
#define LS_F(VX, VW) hipLaunchKernelGGL((lsim_k_linear_fwd<TM, TN, BK, VX, VW, ACT>), grid, threads, lds, s, x, ldx, W, bias, M, K, N, out, ldo, n_tiles, total, aux, ldaux)
    if (vx == 2 && vw == 2) LS_F(2, 2);
    else if (vx >= 1 && vw >= 1) LS_F(1, 1);
    else LS_F(0, 0);
#undef LS_F
}

extern "C" int lsim_linear_elu_forward(const float* x, int64_t ldx, const float* weight, const float* bias, int64_t batch, int k_in, int n_out,
                                       float* out, int64_t ldo, void* stream) {
    if (!x || !weight || !out || ldx < k_in || ldo < n_out) return LSIM_E_INVALID;
    if (!ls_linear_fwd_supported((long)batch, k_in, n_out) || ldo % 4 != 0 || (((uintptr_t)out & 15) != 0))
        return LSIM_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    // feature tile: 128 wide from 256 features on -- 512 -> 256 with 64-wide tiles (2560 tiles, 5 per block) read the sample rows through L2
    // four times and ran at 267 us next to 246 with 128-wide ones (1280 tiles, 3 or 2 per block): the L2 -> LDS traffic costs more than
    // the uneven split.  Narrow layers keep 64 (128 would leave 640 tiles for 512 blocks).
    if (n_out > 128) ls_linear_fwd_launch<5, 4, 32, LS_FWD_ACT_ELU>(x, (long)ldx, weight, bias, (long)batch, k_in, n_out, out, (long)ldo, s);
    else ls_linear_fwd_launch<5, 2, LS_FWD_NARROW_BK, LS_FWD_ACT_ELU>(x, (long)ldx, weight, bias, (long)batch, k_in, n_out, out, (long)ldo, s);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}

// out[b, n] = mask_src[b, n] > 0 ? sum_k x[b, k] W[n, k] : 0 -- a product whose consumer is a ReLU's backward (torch: the GEMM, then
// threshold_backward(grad, saved_output, 0) reading both and writing a third [batch, n_out] array): the gradient penalty's
// d u1 = m1 * (dg W1^T) (learn/amp.py _GradPenFn).  Same kernel and limits as lsim_linear_elu_forward; mask_src [batch, n_out], leading dimension ldm % 4 == 0.
extern "C" int lsim_linear_masked_forward(const float* x, int64_t ldx, const float* weight, const float* mask_src, int64_t ldm, int64_t batch, int k_in, int n_out,
                                          float* out, int64_t ldo, void* stream) {
    if (!x || !weight || !mask_src || !out || ldx < k_in || ldo < n_out || ldm < n_out) return LSIM_E_INVALID;
    if (!ls_linear_fwd_supported((long)batch, k_in, n_out) || ldo % 4 != 0 || ldm % 4 != 0 || (((uintptr_t)out & 15) != 0) || (((uintptr_t)mask_src & 15) != 0))
        return LSIM_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (n_out > 128) ls_linear_fwd_launch<5, 4, 32, LS_FWD_ACT_MASK>(x, (long)ldx, weight, nullptr, (long)batch, k_in, n_out, out, (long)ldo, s, mask_src, (long)ldm);
    else ls_linear_fwd_launch<5, 2, LS_FWD_NARROW_BK, LS_FWD_ACT_MASK>(x, (long)ldx, weight, nullptr, (long)batch, k_in, n_out, out, (long)ldo, s, mask_src, (long)ldm);
    return hipGetLastError() == hipSuccess ? LSIM_OK : LSIM_E_HIP;
}
