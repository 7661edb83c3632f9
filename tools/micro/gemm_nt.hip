// Prototype + stand-alone benchmark of the fused forward kernel of a (Linear, ELU) pair (round 3):  Z[m, n] = elu(sum_k X[m, k] W[n, k] + b[n])
// over a tall batch, fp32, v_mfma_f32_32x32x2_f32, bias + ELU in the epilogue (the separate elu pass over [M, N] disappears).
// The product kernel (csrc/ls_learn.h, lsim_k_linear_elu_fwd) is this kernel; the file stays as the A/B harness:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gemm_nt.hip -o tools/micro/gemm_nt && tools/micro/gemm_nt
//
// Block = 4 waves on a BM x BN tile of Z; K streams through LDS in chunks of 16 (double buffered).  Both operands are K-contiguous in
// memory and stay so in LDS ([row][20]: 16 values + 4 pad, so that the 16-byte reads of 8 consecutive lanes fall into 8 different bank
// groups).  One ds_read_b128 of lane (i, h) fetches row i, k = kk + 4 h .. + 3: component c of the A read and component c of the B read
// are the operands of ONE MFMA (its two k slices are kk + c and kk + 4 + c), so a read pair feeds four MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define NT_KC 16
#define NT_LDK 20

// VEC: 4 = 16-byte global loads (rows 16-byte aligned), 2 = 8-byte, 1 = scalar.  Loads one float4 worth (4 consecutive k) of row `row`
template <int VEC>
__device__ __forceinline__ float4 nt_load4(const float* __restrict__ p, long ld, long row, bool row_ok, int k, int kmax) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok || k >= kmax) return v;
    const float* q = p + row * ld + k;
    if (k + 3 < kmax) {
        if (VEC == 4) return *(const float4*)q;
        if (VEC == 2) { const float2 a = *(const float2*)q, b = *(const float2*)(q + 2); return make_float4(a.x, a.y, b.x, b.y); }
        return make_float4(q[0], q[1], q[2], q[3]);
    }
    v.x = q[0];
    if (k + 1 < kmax) v.y = q[1];
    if (k + 2 < kmax) v.z = q[2];
    return v;
}

// WAVES_M x WAVES_N waves (4 in all); a wave owns 64 rows x 32 TN columns; ACT: 1 = ELU (alpha 1), 0 = none
template <int WAVES_M, int WAVES_N, int TN, int VX, int VW, int ACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_linear_act_fwd(const float* __restrict__ x, long ldx, const float* __restrict__ w, long ldw, const float* __restrict__ bias,
                      float* __restrict__ z, long ldz, long M, int K, int N, int n_blocks) {
    constexpr int BM = WAVES_M * 64, BN = WAVES_N * 32 * TN;
    constexpr int XV = BM * 4 / 256, WV = BN * 4 / 256;      // float4 slots per thread and chunk
    __shared__ __attribute__((aligned(16))) float sX[2][BM][NT_LDK];
    __shared__ __attribute__((aligned(16))) float sW[2][BN][NT_LDK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WAVES_N, wn = wv - wm * WAVES_N;
    const long mb = (long)blockIdx.x / n_blocks;
    const int nb = (int)((long)blockIdx.x - mb * n_blocks);
    const long m_base = mb * BM;
    const int n_base = nb * BN;
    const int i32 = lane & 31, h = lane >> 5;

    v16f acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    float4 rx[XV], rw[WV];
    // staging slot s of thread t: float4 index t + 256 s -> row = idx / 4, k4 = idx % 4
#define NT_LOAD(K0) do {                                                                                             \
        _Pragma("unroll") for (int s_ = 0; s_ < XV; ++s_) { const int idx_ = tid + 256 * s_, r_ = idx_ >> 2, k4_ = idx_ & 3;        \
            rx[s_] = nt_load4<VX>(x, ldx, m_base + r_, m_base + r_ < M, (K0) + 4 * k4_, K); }                         \
        _Pragma("unroll") for (int s_ = 0; s_ < WV; ++s_) { const int idx_ = tid + 256 * s_, r_ = idx_ >> 2, k4_ = idx_ & 3;        \
            rw[s_] = nt_load4<VW>(w, ldw, n_base + r_, n_base + r_ < N, (K0) + 4 * k4_, K); } } while (0)
#define NT_STORE(BUF) do {                                                                                           \
        _Pragma("unroll") for (int s_ = 0; s_ < XV; ++s_) { const int idx_ = tid + 256 * s_, r_ = idx_ >> 2, k4_ = idx_ & 3;        \
            *(float4*)&sX[BUF][r_][4 * k4_] = rx[s_]; }                                                               \
        _Pragma("unroll") for (int s_ = 0; s_ < WV; ++s_) { const int idx_ = tid + 256 * s_, r_ = idx_ >> 2, k4_ = idx_ & 3;        \
            *(float4*)&sW[BUF][r_][4 * k4_] = rw[s_]; } } while (0)

    const int nchunks = (K + NT_KC - 1) / NT_KC;
    NT_LOAD(0); NT_STORE(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const int next_k0 = (ch + 1) * NT_KC;
        if (ch + 1 < nchunks) NT_LOAD(next_k0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {          // k = 8 half .. 8 half + 7: lane (i, h) holds k = 8 half + 4 h + c in component c
            float4 a[2], b[TN];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = *(const float4*)&sX[buf][wm * 64 + 32 * t + i32][8 * half + 4 * h];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = *(const float4*)&sW[buf][wn * 32 * TN + 32 * t + i32][8 * half + 4 * h];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TN; ++tb) {
                        const float av = c == 0 ? a[ta].x : (c == 1 ? a[ta].y : (c == 2 ? a[ta].z : a[ta].w));
                        const float bv = c == 0 ? b[tb].x : (c == 1 ? b[tb].y : (c == 2 ? b[tb].z : b[tb].w));
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ta][tb], 0, 0, 0);
                    }
            }
        }
        if (ch + 1 < nchunks) NT_STORE(buf ^ 1);
        __syncthreads();
    }
#undef NT_LOAD
#undef NT_STORE
    // epilogue: D register r of lane (j = lane % 32, hh = lane / 32) is row 8 (r / 4) + 4 hh + r % 4, column j: bias, activation, store
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int n = n_base + wn * 32 * TN + 32 * tb + i32;
        const float bv = (bias != nullptr && n < N) ? bias[n] : 0.0f;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m_base + wm * 64 + 32 * ta + 8 * (r >> 2) + 4 * h + (r & 3);
                float y = acc[ta][tb][r] + bv;
                if (ACT == 1) y = y > 0.0f ? y : expm1f(y);
                if (m < M && n < N) z[m * ldz + n] = y;
            }
    }
}

__global__ void k_ref(const float* x, long ldx, const float* w, const float* bias, long M, int K, int N, int act, float* z) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= M * N) return;
    const long m = o / N; const int n = (int)(o - m * N);
    double s = bias[n];
    for (int k = 0; k < K; ++k) s += (double)x[m * ldx + k] * (double)w[(long)n * K + k];
    float y = (float)s;
    if (act) y = y > 0.f ? y : expm1f(y);
    z[o] = y;
}

template <int WM, int WN, int TN, int VX, int VW>
static float run_one(const float* x, long ldx, const float* w, const float* b, float* z, long M, int K, int N, int iters) {
    constexpr int BM = WM * 64, BN = WN * 32 * TN;
    const int n_blocks = (N + BN - 1) / BN;
    const long blocks = ((M + BM - 1) / BM) * n_blocks;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < iters + 2; ++it) {
        if (it == 2) CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_linear_act_fwd<WM, WN, TN, VX, VW, 1>), dim3((unsigned)blocks), dim3(256), 0, 0, x, ldx, w, (long)K, b, z, (long)N, M, K, N, n_blocks);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / iters;
}

int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 102400;
    struct Shape { int k, n, ldx; } shapes[] = {{512, 256, 512}, {256, 128, 256}, {64, 512, 64}, {238, 512, 238}, {270, 128, 270}, {128, 64, 128}, {45, 128, 238}};
    const int maxk = 512, maxn = 512;
    float *x, *w, *b, *z, *zr;
    CK(hipMalloc(&x, (size_t)M * maxk * 4 + 64)); CK(hipMalloc(&w, (size_t)maxk * maxn * 4)); CK(hipMalloc(&b, maxn * 4));
    CK(hipMalloc(&z, (size_t)M * maxn * 4)); CK(hipMalloc(&zr, (size_t)M * maxn * 4));
    {
        std::vector<float> hx((size_t)M * maxk + 16), hw((size_t)maxk * maxn), hb(maxn);
        unsigned s = 777u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd() * 0.08f;
        for (auto& v : hb) v = rnd() * 0.1f;
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    }
    const char* only = getenv("GEMM_SHAPE");
    printf("{\"batch\": %ld, \"runs\": [\n", M);
    int si = -1;
    for (auto sh : shapes) {
        ++si;
        if (only && atoi(only) != si) continue;
        const float* xp = sh.k == 45 ? x + 3 : x;            // the target network's input: a column slice of the critic observations
        for (int variant = 0; variant < 3; ++variant) {
            float ms = -1.f; const char* name = "";
            const int vx = sh.k == 45 ? 1 : ((sh.ldx % 4 == 0) ? 4 : 2), vw = (sh.k % 4 == 0) ? 4 : ((sh.k % 2 == 0) ? 2 : 1);
#define RUNV(WM, WN, TN) (vx == 4 ? (vw == 4 ? run_one<WM, WN, TN, 4, 4>(xp, sh.ldx, w, b, z, M, sh.k, sh.n, 10) : run_one<WM, WN, TN, 4, 2>(xp, sh.ldx, w, b, z, M, sh.k, sh.n, 10)) \
                          : vx == 2 ? run_one<WM, WN, TN, 2, 2>(xp, sh.ldx, w, b, z, M, sh.k, sh.n, 10) : run_one<WM, WN, TN, 1, 1>(xp, sh.ldx, w, b, z, M, sh.k, sh.n, 10))
            if (variant == 0) { name = "128x256"; if (sh.n < 256) continue; ms = RUNV(2, 2, 4); }
            else if (variant == 1) { name = "128x128"; if (sh.n < 128) continue; ms = RUNV(2, 2, 2); }
            else { name = "256x64"; if (sh.n > 128) continue; ms = RUNV(4, 1, 2); }
            hipLaunchKernelGGL(k_ref, dim3((unsigned)((M * sh.n + 255) / 256)), dim3(256), 0, 0, xp, (long)sh.ldx, w, b, M, sh.k, sh.n, 1, zr);
            std::vector<float> h((size_t)M * sh.n), r((size_t)M * sh.n);
            CK(hipMemcpy(h.data(), z, h.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), zr, r.size() * 4, hipMemcpyDeviceToHost));
            double err = 0, mag = 0;
            for (size_t i = 0; i < h.size(); ++i) { err = fmax(err, fabs((double)h[i] - r[i])); mag = fmax(mag, fabs((double)r[i])); }
            printf("  {\"k_in\": %d, \"n_out\": %d, \"ldx\": %d, \"tile\": \"%s\", \"us\": %.1f, \"tflops\": %.1f, \"max_err\": %.3g, \"max_ref\": %.3g},\n",
                   sh.k, sh.n, sh.ldx, name, ms * 1e3, 2.0 * M * sh.k * sh.n / (ms * 1e-3) * 1e-12, err, mag);
            fflush(stdout);
        }
    }
    printf("  {}]}\n");
    return 0;
}
