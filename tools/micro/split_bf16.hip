// fp32 GEMM work on the bf16 matrix pipe: a = a0 + a1 + a2 with three bf16 terms is an EXACT split of an fp32 value (3 x 8 significand bits),
// and the six products a_i b_j with i + j <= 2 -- each exact in the pipe's fp32 accumulator -- leave a truncation of 2^-24 |a||b|, the size of
// fp32's own rounding.  v_mfma_f32_16x16x32_bf16 moves 16x the multiply-adds per cycle of v_mfma_f32_16x16x4_f32, so six of them per K chunk
// are 2.67x the fp32 pipe's rate on paper (419 against 157 TFLOP/s).  This probe measures the two things a kernel built on it would live on:
//   rate     : the steady-state loop of a 64 x 128 register tile per wave (4 x 8 tiles of 16 x 16) -- 96 fresh fp32 operand values per lane
//              and K = 32 step, split in registers (v_cvt_pk_bf16_f32 + subtract), then 192 bf16 MFMAs -- against the same tile on the
//              fp32 pipe (256 MFMAs per K = 32);
//   accuracy : one 16 x 16 tile of A^T B over a long K (the weight gradient's shape: K = rows of the minibatch) through the fp32 pipe, the
//              six-product and the three-product forms, against an fp64 sum on the host.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/split_bf16.hip -o tools/micro/split_bf16      Run: tools/micro/split_bf16 [json-out]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Split { bf8 p0, p1, p2; };
__device__ __forceinline__ Split split8(const float* v) {
    Split s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        const float r1 = v[i] - (float)h;           // exact
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;             // exact
        s.p0[i] = h; s.p1[i] = m; s.p2[i] = (__bf16)r2;
    }
    return s;
}
#define MF(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)

// ---- rate -------------------------------------------------------------------------------------------------------------------------------
// MODE 0: MFMAs alone (operands split once); 1: fresh operand values and their split every step; 2: as 1 with the three-product form
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_rate_split(float* out, int iters, float seed) {
    v4f acc[4][8];
    for (int j = 0; j < 4; ++j) for (int t = 0; t < 8; ++t) acc[j][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    float a[4][8], b[8][8];
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; a[j][i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f + seed; }
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; b[t][i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f + seed; }
    Split sa[4];
    for (int j = 0; j < 4; ++j) sa[j] = split8(a[j]);
    Split sb0[8];
    if (MODE == 0) for (int t = 0; t < 8; ++t) sb0[t] = split8(b[t]);
    for (int it = 0; it < iters; ++it) {
        if (MODE != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[j][i] += 0.001953125f;
                sa[j] = split8(a[j]);
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            Split sb;
            if (MODE == 0) sb = sb0[t];
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i) b[t][i] += 0.00390625f;
                sb = split8(b[t]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE != 2) { MF(sa[j].p2, sb.p0, acc[j][t]); MF(sa[j].p1, sb.p1, acc[j][t]); MF(sa[j].p0, sb.p2, acc[j][t]); }
                MF(sa[j].p1, sb.p0, acc[j][t]); MF(sa[j].p0, sb.p1, acc[j][t]); MF(sa[j].p0, sb.p0, acc[j][t]);
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int t = 0; t < 8; ++t) s += acc[j][t][0] + acc[j][t][1] + acc[j][t][2] + acc[j][t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same tile and K step on the fp32 pipe: 8 k-steps of 4, fresh operand values every step
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_rate_f32(float* out, int iters, float seed) {
    v4f acc[4][8];
    for (int j = 0; j < 4; ++j) for (int t = 0; t < 8; ++t) acc[j][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    float a[4][8], b[8][8];
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; a[j][i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f + seed; }
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; b[t][i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f + seed; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[j][i] += 0.001953125f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) b[t][i] += 0.00390625f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], b[t][i], acc[j][t], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int t = 0; t < 8; ++t) s += acc[j][t][0] + acc[j][t][1] + acc[j][t][2] + acc[j][t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- accuracy ---------------------------------------------------------------------------------------------------------------------------
// C[i][j] = sum_k A[k][i] B[k][j], A and B stored [K][16] (the weight gradient's operands: rows = samples).  One wave, one tile, three ways.
// 16x16x32 bf16 operand layout: lane l holds k = 8 (l / 16) .. + 7 of row / column l % 16; C: column l % 16, rows 4 (l / 16) .. + 3.
__global__ __launch_bounds__(64) void k_acc(const float* __restrict__ A, const float* __restrict__ B, int K, float* c_f32, float* c_six, float* c_six2, float* c_three) {
    const int lane = threadIdx.x, col = lane & 15, kg = lane >> 4;
    v4f f32 = {0, 0, 0, 0}, six = {0, 0, 0, 0}, hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0}, three = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        float a[8], b[8];
        for (int i = 0; i < 8; ++i) { a[i] = A[(long)(k0 + 8 * kg + i) * 16 + col]; b[i] = B[(long)(k0 + 8 * kg + i) * 16 + col]; }
        // fp32 pipe: 16x16x4 takes k = l / 16 of each group of four
        for (int s = 0; s < 8; ++s) {
            const float af = A[(long)(k0 + 4 * s + kg) * 16 + col], bf = B[(long)(k0 + 4 * s + kg) * 16 + col];
            f32 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, f32, 0, 0, 0);
        }
        const Split sa = split8(a), sb = split8(b);
        MF(sa.p2, sb.p0, six); MF(sa.p1, sb.p1, six); MF(sa.p0, sb.p2, six); MF(sa.p1, sb.p0, six); MF(sa.p0, sb.p1, six); MF(sa.p0, sb.p0, six);
        MF(sa.p2, sb.p0, lo); MF(sa.p1, sb.p1, lo); MF(sa.p0, sb.p2, lo); MF(sa.p1, sb.p0, lo); MF(sa.p0, sb.p1, lo); MF(sa.p0, sb.p0, hi);
        MF(sa.p1, sb.p0, three); MF(sa.p0, sb.p1, three); MF(sa.p0, sb.p0, three);
    }
    for (int r = 0; r < 4; ++r) {
        const int o = (4 * kg + r) * 16 + col;
        c_f32[o] = f32[r]; c_six[o] = six[r]; c_six2[o] = hi[r] + lo[r]; c_three[o] = three[r];
    }
}

static double time_ms(void (*launch)(float*, int), float* out, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(out, iters / 8); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}
#define LAUNCHER(NAME, KERNEL) static void NAME(float* out, int iters) { hipLaunchKernelGGL(KERNEL, dim3(256), dim3(256), 0, 0, out, iters, 0.25f); }
LAUNCHER(l_f32, k_rate_f32)
LAUNCHER(l_s0, k_rate_split<0>)
LAUNCHER(l_s1, k_rate_split<1>)
LAUNCHER(l_s2, k_rate_split<2>)

int main(int argc, char** argv) {
    float* out; CK(hipMalloc(&out, 256 * 256 * sizeof(float)));
    const int iters = 20000;
    // per step and wave: 4 x 8 tiles x 16 x 16 x 32 multiply-adds; 256 blocks x 4 waves
    const double flop = 2.0 * 32 * 16 * 16 * 32 * (double)iters * 256 * 4;
    const double t_f32 = time_ms(l_f32, out, iters), t_s0 = time_ms(l_s0, out, iters), t_s1 = time_ms(l_s1, out, iters), t_s2 = time_ms(l_s2, out, iters);
    const double r_f32 = flop / t_f32 * 1e-9, r_s0 = flop / t_s0 * 1e-9, r_s1 = flop / t_s1 * 1e-9, r_s2 = flop / t_s2 * 1e-9;
    printf("rate (fp32-equivalent TFLOP/s, 256 blocks x 4 waves, one wave per SIMD):\n");
    printf("  fp32 pipe, fresh operands            %8.1f\n  six bf16 products, MFMAs alone       %8.1f\n  six bf16 products + split each step  %8.1f\n  three bf16 products + split          %8.1f\n",
           r_f32, r_s0, r_s1, r_s2);

    // accuracy
    struct Case { const char* name; int K; int kind; } cases[] = {{"K=102400 normal x normal", 102400, 0}, {"K=102400 gradient-like (wide range, sparse sign)", 102400, 1},
                                                                    {"K=512 normal x normal", 512, 0}, {"K=102400 same sign (no cancellation)", 102400, 2}};
    double res[4][4][2];
    for (int c = 0; c < 4; ++c) {
        const int K = cases[c].K;
        std::vector<float> A((size_t)K * 16), B((size_t)K * 16);
        unsigned long long s = 0x9E3779B97F4A7C15ull + c;
        auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) * (1.0 / 9007199254740992.0); };
        auto nrm = [&]() { return std::sqrt(-2.0 * std::log(u() + 1e-300)) * std::cos(6.283185307179586 * u()); };
        for (size_t i = 0; i < A.size(); ++i) {
            double x = nrm(), y = nrm();
            if (cases[c].kind == 1) { x *= std::exp(3.0 * nrm()) * 1e-3; y = (u() < 0.5 ? 0.0 : y); }
            if (cases[c].kind == 2) { x = std::fabs(x); y = std::fabs(y); }
            A[i] = (float)x; B[i] = (float)y;
        }
        float *dA, *dB, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 4 * 256 * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dA, dB, K, dC, dC + 256, dC + 512, dC + 768);
        std::vector<float> C(4 * 256); CK(hipMemcpy(C.data(), dC, 4 * 256 * 4, hipMemcpyDeviceToHost));
        std::vector<double> ref(256, 0.0), mag(256, 0.0);
        for (int k = 0; k < K; ++k) for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            const double p = (double)A[(size_t)k * 16 + i] * (double)B[(size_t)k * 16 + j]; ref[i * 16 + j] += p; mag[i * 16 + j] += std::fabs(p); }
        printf("%s: error / sum |a||b|   (max, rms over the 256 outputs)\n", cases[c].name);
        const char* names[4] = {"fp32 pipe (16x16x4 f32)", "six products, one accumulator", "six products, a0 b0 apart", "three products"};
        for (int m = 0; m < 4; ++m) {
            double mx = 0, sq = 0;
            for (int o = 0; o < 256; ++o) { const double e = std::fabs((double)C[m * 256 + o] - ref[o]) / mag[o]; mx = e > mx ? e : mx; sq += e * e; }
            res[c][m][0] = mx; res[c][m][1] = std::sqrt(sq / 256);
            printf("  %-32s %.3e  %.3e\n", names[m], mx, res[c][m][1]);
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        fprintf(f, "{\"what\": \"fp32 GEMM work on the bf16 matrix pipe (tools/micro/split_bf16.hip): register-tile loop rates and accuracy of one 16x16 tile against fp64\",\n");
        fprintf(f, " \"rate_tflops_fp32_equivalent\": {\"fp32_pipe\": %.1f, \"six_products_mfma_alone\": %.1f, \"six_products_with_split\": %.1f, \"three_products_with_split\": %.1f},\n", r_f32, r_s0, r_s1, r_s2);
        fprintf(f, " \"accuracy_error_over_sum_abs_products\": {\n");
        const char* keys[4] = {"fp32_pipe", "six_products", "six_products_a0b0_apart", "three_products"};
        for (int c = 0; c < 4; ++c) {
            fprintf(f, "  \"%s\": {", cases[c].name);
            for (int m = 0; m < 4; ++m) fprintf(f, "\"%s\": {\"max\": %.3e, \"rms\": %.3e}%s", keys[m], res[c][m][0], res[c][m][1], m < 3 ? ", " : "");
            fprintf(f, "}%s\n", c < 3 ? "," : "");
        }
        fprintf(f, " }}\n"); fclose(f);
    }
    return 0;
}
