// Prototype + stand-alone benchmark of the LDS-tiled weight-gradient kernel (round 3):  dW[n, k] = sum_b G[b, n] * X[b, k]
// over a tall batch, fp32, v_mfma_f32_32x32x2_f32.  The product kernel (csrc/ls_learn.h, lsim_k_wgrad_lds) is this kernel; the file
// stays as the A/B harness:  hipcc --offload-arch=gfx950 -O3 tools/micro/gemm_tn.hip -o tools/micro/gemm_tn && tools/micro/gemm_tn
//
// Block = 4 waves on one (n tile, k tile, batch slice); the slice's rows stream through LDS in chunks of KB = 16 rows (double
// buffered), stored exactly as in memory ([row][column]); a wave owns 64 (n) x 32 TK (k) outputs = 2 x TK MFMA tiles:
//   A operand (32 n x 2 rows): one ds_read_b64 -- lane (i, h) reads G[row0 + h][n0 + 2 i .. + 1], register j is row i of n-tile j
//   B operand (2 rows x 32 k): one ds_read_b{32 TK} -- lane (i, h) reads X[row0 + h][k0 + TK i ..], register q is column i of k-tile q
// so every LDS read is a contiguous run over the lanes (no bank conflicts) and output (j, q)[i][c] is dW[n0 + 2 i + j][k0 + TK c + q].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int TK> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<4> { typedef float4 T; };

template <int WAVES_N, int WAVES_K, int TK, bool FZ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_wgrad_lds(const float* __restrict__ x, long ldx, const float* __restrict__ g, long ldg, const float* __restrict__ z, long ldz,
                 float* __restrict__ gy, long batch, int k_in, int n_out, int k_blocks, int tiles, long rows_per_slice,
                 float* __restrict__ part_dw, float* __restrict__ part_db) {
    constexpr int BN = WAVES_N * 64, WK = 32 * TK, BK = WAVES_K * WK, KB = 16;
    constexpr int GV = BN / 64, XV = BK / 64;          // float4 loads per thread and chunk
    __shared__ __attribute__((aligned(16))) float sG[2][KB][BN];
    __shared__ __attribute__((aligned(16))) float sX[2][KB][BK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv / WAVES_K, wk = wv - wn * WAVES_K;
    const int total = (int)gridDim.x, xcd = (int)blockIdx.x & 7;
    const long v = (long)xcd * (total >> 3) + (xcd < (total & 7) ? xcd : (total & 7)) + ((long)blockIdx.x >> 3);
    const long slice = v / tiles;
    const int tile = (int)(v - slice * tiles);
    const int nb = tile / k_blocks, kb = tile - nb * k_blocks;
    const int n_base = nb * BN, k_base = kb * BK;
    const long s0 = slice * rows_per_slice;
    long s1 = s0 + rows_per_slice;
    if (s1 > batch) s1 = batch;
    const int i32 = lane & 31, h = lane >> 5;

    v16f acc[2][TK];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < TK; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][q][r] = 0.0f;
    float dbacc[2] = {0.0f, 0.0f};

    // per-thread load slots: G chunk = KB x BN floats = 4 BN float4, thread t takes float4 index t + 256 p
    float4 rg[GV], rz[GV], rx[XV];
    const bool write_gy = FZ && kb == 0 && gy != nullptr;
#define LOAD_CHUNK(ROW0) do {                                                                                       \
        _Pragma("unroll") for (int p = 0; p < GV; ++p) {                                                             \
            const int idx = tid + 256 * p, r = idx / (BN / 4), c = 4 * (idx - r * (BN / 4));                        \
            const long row = (ROW0) + r; const int n = n_base + c;                                                   \
            const bool ok = row < s1 && n < n_out;                                                                   \
            const long rr = ok ? row : s0; const int nn = ok ? n : 0;                                                \
            float4 t = *(const float4*)(g + rr * ldg + nn);                                                          \
            if (FZ) { const float4 u = *(const float4*)(z + rr * ldz + nn);                                          \
                t.x *= u.x > 0.0f ? 1.0f : u.x + 1.0f; t.y *= u.y > 0.0f ? 1.0f : u.y + 1.0f;                         \
                t.z *= u.z > 0.0f ? 1.0f : u.z + 1.0f; t.w *= u.w > 0.0f ? 1.0f : u.w + 1.0f;                         \
                if (write_gy && ok) *(float4*)(gy + row * (long)n_out + n) = t; }                                    \
            rg[p] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);                                                        \
        }                                                                                                            \
        _Pragma("unroll") for (int p = 0; p < XV; ++p) {                                                             \
            const int idx = tid + 256 * p, r = idx / (BK / 4), c = 4 * (idx - r * (BK / 4));                        \
            const long row = (ROW0) + r; const int k = k_base + c;                                                   \
            const bool ok = row < s1 && k < k_in;                                                                    \
            const long rr = ok ? row : s0; const int kk = ok ? k : 0;                                                \
            const float4 t = *(const float4*)(x + rr * ldx + kk);                                                    \
            rx[p] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);                                                        \
        } } while (0)
#define STORE_CHUNK(BUF) do {                                                                                       \
        _Pragma("unroll") for (int p = 0; p < GV; ++p) {                                                             \
            const int idx = tid + 256 * p, r = idx / (BN / 4), c = 4 * (idx - r * (BN / 4));                        \
            *(float4*)&sG[BUF][r][c] = rg[p]; }                                                                      \
        _Pragma("unroll") for (int p = 0; p < XV; ++p) {                                                             \
            const int idx = tid + 256 * p, r = idx / (BK / 4), c = 4 * (idx - r * (BK / 4));                        \
            *(float4*)&sX[BUF][r][c] = rx[p]; } } while (0)

    const long nrows = s1 - s0;
    const int nchunks = (int)((nrows + KB - 1) / KB);
    if (nchunks > 0) { LOAD_CHUNK(s0); STORE_CHUNK(0); }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const long next_row0 = s0 + (long)(ch + 1) * KB;
        if (ch + 1 < nchunks) LOAD_CHUNK(next_row0);
#pragma unroll
        for (int kp = 0; kp < KB / 2; ++kp) {
            const int row = 2 * kp + h;
            const float2 a = *(const float2*)&sG[buf][row][wn * 64 + 2 * i32];
            const typename VecT<TK>::T bv = *(const typename VecT<TK>::T*)&sX[buf][row][wk * WK + TK * i32];
            float b[TK];
            if constexpr (TK == 1) b[0] = bv;
            else if constexpr (TK == 2) { b[0] = bv.x; b[1] = bv.y; }
            else { b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w; }
            dbacc[0] += a.x; dbacc[1] += a.y;
#pragma unroll
            for (int q = 0; q < TK; ++q) {
                acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[q], acc[0][q], 0, 0, 0);
                acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[q], acc[1][q], 0, 0, 0);
            }
        }
        if (ch + 1 < nchunks) STORE_CHUNK(buf ^ 1);
        __syncthreads();
    }
#undef LOAD_CHUNK
#undef STORE_CHUNK
    // D(32x32) register r of lane (c = lane % 32, hh = lane / 32) is row i = 8 (r / 4) + 4 hh + r % 4, column c
    float* pw = part_dw + (size_t)slice * n_out * k_in;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r >> 2) + 4 * h + (r & 3);
            const int n = n_base + wn * 64 + 2 * i + j;
            const int k = k_base + wk * WK + TK * i32;
            if (n >= n_out) continue;
            float* dst = pw + (size_t)n * k_in + k;
            if (TK == 4 && (k_in & 3) == 0 && k + 3 < k_in) *(float4*)dst = make_float4(acc[j][0][r], acc[j][TK > 1 ? 1 : 0][r], acc[j][TK > 2 ? 2 : 0][r], acc[j][TK > 3 ? 3 : 0][r]);
            else {
#pragma unroll
                for (int q = 0; q < TK; ++q) if (k + q < k_in) dst[q] = acc[j][q][r];
            }
        }
    if (part_db && kb == 0 && wk == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float vv = dbacc[j];
            vv += __shfl_xor(vv, 32, 64);
            const int n = n_base + wn * 64 + 2 * i32 + j;
            if (h == 0 && n < n_out) part_db[(size_t)slice * n_out + n] = vv;
        }
    }
}

__global__ void k_reduce(const float* part, int num, int count, float* out) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= count) return;
    float s = 0.f;
    for (int w = 0; w < num; ++w) s += part[(size_t)w * count + o];
    out[o] = s;
}
__global__ void k_ref(const float* x, long ldx, const float* g, long ldg, const float* z, long batch, int k_in, int n_out, double* dw) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out * k_in) return;
    const int n = o / k_in, k = o - n * k_in;
    double s = 0.0;
    for (long b = 0; b < batch; ++b) {
        float gv = g[b * ldg + n];
        if (z) { const float zz = z[b * ldg + n]; gv *= zz > 0.f ? 1.f : zz + 1.f; }
        s += (double)gv * (double)x[b * ldx + k];
    }
    dw[o] = s;
}

template <int WN, int WKK, int TK, bool FZ>
static float run_one(const float* x, long ldx, const float* g, const float* z, float* gy, long batch, int k_in, int n_out, float* part, float* dw, float* db,
                     int iters, int* blocks_out) {
    constexpr int BN = WN * 64, BK = WKK * 32 * TK;
    const int n_blocks = (n_out + BN - 1) / BN, k_blocks = (k_in + BK - 1) / BK, tiles = n_blocks * k_blocks;
    long slices = 512 / tiles; if (slices < 1) slices = 1;
    long rps = ((batch + slices - 1) / slices + 15) & ~15L;
    const int partials = (int)((batch + rps - 1) / rps);
    const int blocks = tiles * partials;
    *blocks_out = blocks;
    float* pdb = part + (size_t)partials * n_out * k_in;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < iters + 2; ++it) {
        if (it == 2) CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_wgrad_lds<WN, WKK, TK, FZ>), dim3(blocks), dim3(256), 0, 0, x, ldx, g, (long)n_out, z, (long)n_out, gy, batch, k_in, n_out,
                           k_blocks, tiles, rps, part, pdb);
        hipLaunchKernelGGL(k_reduce, dim3((n_out * k_in + 255) / 256), dim3(256), 0, 0, (const float*)part, partials, n_out * k_in, dw);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / iters;
}

int main(int argc, char** argv) {
    const long batch = argc > 1 ? atol(argv[1]) : 102400;
    struct Shape { int k, n; } shapes[] = {{512, 256}, {256, 128}, {64, 512}, {128, 64}, {240, 512}, {272, 128}};
    const int maxk = 512, maxn = 512;
    float *x, *g, *z, *gy, *part, *dw, *db; double* ref;
    CK(hipMalloc(&x, (size_t)batch * maxk * 4)); CK(hipMalloc(&g, (size_t)batch * maxn * 4)); CK(hipMalloc(&z, (size_t)batch * maxn * 4));
    CK(hipMalloc(&gy, (size_t)batch * maxn * 4)); CK(hipMalloc(&part, (size_t)256 << 20)); CK(hipMalloc(&dw, (size_t)maxk * maxn * 4));
    CK(hipMalloc(&db, maxn * 4)); CK(hipMalloc(&ref, (size_t)maxk * maxn * 8));
    {
        std::vector<float> hx((size_t)batch * maxk), hg((size_t)batch * maxn), hz((size_t)batch * maxn);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; };
        for (auto& v : hx) v = rnd();
        for (auto& v : hg) v = rnd() * 0.01f;
        for (auto& v : hz) v = rnd();
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice));
    }
    if (argc > 2) {   // debug: G[b][n] = n + 1 and X[b][k] = k + 1 on row 0 only -> dW[n][k] = (n + 1)(k + 1); print the first mismatches
        const int K = 256, N = 128;
        std::vector<float> hx((size_t)batch * K, 0.f), hg((size_t)batch * N, 0.f);
        const int mode = atoi(argv[2]);
        if (mode == 0) {
            for (int n = 0; n < N; ++n) hg[n] = (float)(n + 1);
            for (int k = 0; k < K; ++k) hx[k] = (float)(k + 1);
        } else {     // every row: G[b][n] = (n % 4 + 1) * (b % 3), X[b][k] = (k % 8 + 1) * (b % 2 + 1): exact small integers
            for (long b = 0; b < batch; ++b) { for (int n = 0; n < N; ++n) hg[b * N + n] = (float)((n % 4 + 1) * (b % 3)); for (int k = 0; k < K; ++k) hx[b * K + k] = (float)((k % 8 + 1) * (b % 2 + 1)); }
        }
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
        int blocks;
        run_one<2, 2, 4, false>(x, K, g, nullptr, nullptr, batch, K, N, part, dw, db, 1, &blocks);
        std::vector<float> h((size_t)N * K);
        CK(hipMemcpy(h.data(), dw, h.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int n = 0; n < N && bad < 40; ++n) for (int k = 0; k < K && bad < 40; ++k) {
            float want = (float)(n + 1) * (k + 1);
            if (mode != 0) { double sb = 0; for (long b = 0; b < batch; ++b) sb += (double)((b % 3) * (b % 2 + 1)); want = (float)(sb * (n % 4 + 1) * (k % 8 + 1)); }
            if (h[(size_t)n * K + k] != want) { printf("dW[%d][%d] = %g (want %g)\n", n, k, h[(size_t)n * K + k], want); ++bad; }
        }
        printf("debug done, blocks %d, bad %d\n", blocks, bad);
        return 0;
    }
    printf("{\"batch\": %ld, \"runs\": [\n", batch);
    const char* only = getenv("GEMM_SHAPE");
    int shape_index = -1;
    for (auto sh : shapes) {
        ++shape_index;
        if (only && atoi(only) != shape_index) continue;
        for (int fz = 0; fz < 2; ++fz) {
            for (int variant = 0; variant < 3; ++variant) {
                int blocks = 0; float ms = -1.f; const char* name = "";
                const long ldx = sh.k;
#define RUN(WN, WKK, TK) (fz ? run_one<WN, WKK, TK, true>(x, ldx, g, z, gy, batch, sh.k, sh.n, part, dw, db, 10, &blocks) \
                             : run_one<WN, WKK, TK, false>(x, ldx, g, nullptr, nullptr, batch, sh.k, sh.n, part, dw, db, 10, &blocks))
                if (variant == 0) { name = "128x256"; if (sh.k < 128) continue; ms = RUN(2, 2, 4); }
                else if (variant == 1) { name = "128x128"; ms = RUN(2, 2, 2); }
                else { name = "256x64"; if (sh.k > 128) continue; ms = RUN(4, 1, 2); }
                hipLaunchKernelGGL(k_ref, dim3((sh.n * sh.k + 255) / 256), dim3(256), 0, 0, x, ldx, g, (long)sh.n, fz ? z : nullptr, batch, sh.k, sh.n, ref);
                std::vector<float> h((size_t)sh.n * sh.k); std::vector<double> r((size_t)sh.n * sh.k);
                CK(hipMemcpy(h.data(), dw, h.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), ref, r.size() * 8, hipMemcpyDeviceToHost));
                double err = 0, mag = 0;
                for (size_t i = 0; i < h.size(); ++i) { err = fmax(err, fabs((double)h[i] - r[i])); mag = fmax(mag, fabs(r[i])); }
                printf("  {\"k_in\": %d, \"n_out\": %d, \"elu\": %d, \"tile\": \"%s\", \"blocks\": %d, \"us\": %.1f, \"tflops\": %.1f, \"max_err\": %.3g, \"max_ref\": %.3g},\n",
                       sh.k, sh.n, fz, name, blocks, ms * 1e3, 2.0 * batch * sh.k * sh.n / (ms * 1e-3) * 1e-12, err, mag);
                fflush(stdout);
            }
        }
    }
    printf("  {}]}\n");
    return 0;
}
