// Sustained fp32 MFMA rate of the device (v_mfma_f32_16x16x4_f32, the instruction of the learner-side kernels): WAVES waves per SIMD
// issuing independent MFMAs from registers only.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the operand pattern of the weight-gradient kernels: 4 A registers x 8 B registers -> 32 accumulators, A/B refreshed every step
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k48(float* out, int iters, float a0) {
    v4f acc[4][8];
    for (int j = 0; j < 4; ++j) for (int t = 0; t < 8; ++t) acc[j][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    float a[4], b[8];
    for (int j = 0; j < 4; ++j) a[j] = a0 + threadIdx.x * (j + 1);
    for (int t = 0; t < 8; ++t) b[t] = a0 * (t + 2);
    if (MODE == 2) {          // random bit patterns in every lane and register (data-dependent power: real gradients look like this)
        unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
        for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; a[j] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }
        for (int t = 0; t < 8; ++t) { h = h * 1664525u + 1013904223u; b[t] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {          // new operand values every step (as if freshly loaded): 12 VALU per 32 MFMAs
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += 1.0f;
#pragma unroll
            for (int t = 0; t < 8; ++t) b[t] += 0.5f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[t], acc[j][t], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int t = 0; t < 8; ++t) s += acc[j][t][0] + acc[j][t][1] + acc[j][t][2] + acc[j][t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the steady-state loop of the tiled weight-gradient kernel with operands from (cache-resident) memory: three register stages, per step
// three 16-byte loads per lane and 32 MFMAs (A: 4 components of one vector, B: 2 x 4 components)
template <bool BARRIER, int NL = 3, int ADDR = 0 /* 0: per-lane 64-bit pointers, 1: uniform base + 32-bit lane offset */>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void kld(const float* __restrict__ src, float* out, int iters, int mask) {
    v4f acc[4][8];
    for (int j = 0; j < 4; ++j) for (int t = 0; t < 8; ++t) acc[j][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    const float4* p = (const float4*)src + (ADDR == 0 ? threadIdx.x : 0);
    const unsigned lane_off = ADDR == 0 ? 0u : threadIdx.x;
    float4 a0, x0, y0, a1, x1, y1, a2, x2, y2;
#define LD(A, X, Y, I) do { const int o = ((I) * 768) & mask; if (NL > 0) A = p[(unsigned)o + lane_off]; if (NL > 1) X = p[(unsigned)(o + 256) + lane_off]; if (NL > 2) Y = p[(unsigned)(o + 512) + lane_off]; } while (0)
    a0 = x0 = y0 = a1 = x1 = y1 = a2 = x2 = y2 = p[0];
#define CMP(A, X, Y) do { const float av[4] = {A.x, A.y, A.z, A.w}; const float bv[8] = {X.x, X.y, X.z, X.w, Y.x, Y.y, Y.z, Y.w};   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int t = 0; t < 8; ++t)                                  \
            acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[t], acc[j][t], 0, 0, 0); } while (0)
    int i = blockIdx.x;
    LD(a0, x0, y0, i); LD(a1, x1, y1, i + 1);
    for (int it = 0; it < iters; it += 3, i += 3) {
        // sched_barrier: hipcc's scheduler otherwise sinks the loads down to their first use (s_waitcnt vmcnt(0) before the MFMAs)
#define SB() do { if (BARRIER) __builtin_amdgcn_sched_barrier(0); } while (0)
        LD(a2, x2, y2, i + 2); SB(); CMP(a0, x0, y0); SB();
        LD(a0, x0, y0, i + 3); SB(); CMP(a1, x1, y1); SB();
        LD(a1, x1, y1, i + 4); SB(); CMP(a2, x2, y2); SB();
#undef SB
    }
#undef LD
#undef CMP
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int t = 0; t < 8; ++t) s += acc[j][t][0] + acc[j][t][1] + acc[j][t][2] + acc[j][t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same work with the three loads of a step spread between the MFMAs (one load after every 8 / 12 / 12 MFMAs) instead of clustered
template <int PRIO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void kspread(const float* __restrict__ src, float* out, int iters, int mask) {
    v4f acc[4][8];
    for (int j = 0; j < 4; ++j) for (int t = 0; t < 8; ++t) acc[j][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    const float4* p = (const float4*)src + threadIdx.x;
    float4 a0, x0, y0, a1, x1, y1, a2, x2, y2;
    a0 = x0 = y0 = a1 = x1 = y1 = a2 = x2 = y2 = p[0];
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(A, X, Y, J) do { const float av = (J) == 0 ? A.x : (J) == 1 ? A.y : (J) == 2 ? A.z : A.w; const float bv[8] = {X.x, X.y, X.z, X.w, Y.x, Y.y, Y.z, Y.w}; \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                                     \
        _Pragma("unroll") for (int t = 0; t < 8; ++t) acc[J][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[J][t], 0, 0, 0);  \
        if (PRIO) __builtin_amdgcn_s_setprio(0); } while (0)
#define STEP(CA, CX, CY, LA, LX, LY, I) do { const int o = ((I) * 768) & mask;                                               \
        MF(CA, CX, CY, 0); SB(); LA = p[o]; SB(); MF(CA, CX, CY, 1); SB(); LX = p[o + 256]; SB();                           \
        MF(CA, CX, CY, 2); SB(); LY = p[o + 512]; SB(); MF(CA, CX, CY, 3); SB(); } while (0)
    int i = blockIdx.x;
    for (int it = 0; it < iters; it += 3, i += 3) {
        STEP(a0, x0, y0, a2, x2, y2, i + 2);
        STEP(a1, x1, y1, a0, x0, y0, i + 3);
        STEP(a2, x2, y2, a1, x1, y1, i + 4);
    }
#undef STEP
#undef MF
#undef SB
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int t = 0; t < 8; ++t) s += acc[j][t][0] + acc[j][t][1] + acc[j][t][2] + acc[j][t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// one wave per SIMD owning a 128 x 128 tile: 64 accumulators (256 registers), four loads spread over the 64 MFMAs of a step
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void kbig(const float* __restrict__ src, float* out, int iters, int mask) {
    v4f acc[8][8];
    for (int j = 0; j < 8; ++j) for (int t = 0; t < 8; ++t) acc[j][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    const float4* p = (const float4*)src + threadIdx.x;
    float4 st[3][4];
    for (int s = 0; s < 3; ++s) for (int q = 0; q < 4; ++q) st[s][q] = p[0];
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF16(C, JJ) do { const float av[8] = {st[C][0].x, st[C][0].y, st[C][0].z, st[C][0].w, st[C][1].x, st[C][1].y, st[C][1].z, st[C][1].w};                  \
        const float bv[8] = {st[C][2].x, st[C][2].y, st[C][2].z, st[C][2].w, st[C][3].x, st[C][3].y, st[C][3].z, st[C][3].w};                                    \
        __builtin_amdgcn_s_setprio(1);                                                                                                                            \
        _Pragma("unroll") for (int j = 2 * (JJ); j < 2 * (JJ) + 2; ++j) _Pragma("unroll") for (int t = 0; t < 8; ++t)                                          \
            acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[t], acc[j][t], 0, 0, 0);                                                                   \
        __builtin_amdgcn_s_setprio(0); } while (0)
#define STEP(C, L, I) do { const int o = ((I) * 1024) & mask;                                                                \
        MF16(C, 0); SB(); st[L][0] = p[o]; SB(); MF16(C, 1); SB(); st[L][1] = p[o + 256]; SB();                             \
        MF16(C, 2); SB(); st[L][2] = p[o + 512]; SB(); MF16(C, 3); SB(); st[L][3] = p[o + 768]; SB(); } while (0)
    int i = blockIdx.x;
    for (int it = 0; it < iters; it += 3, i += 3) {
        STEP(0, 2, i + 2);
        STEP(1, 0, i + 3);
        STEP(2, 1, i + 4);
    }
#undef STEP
#undef MF16
#undef SB
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int t = 0; t < 8; ++t) s += acc[j][t][0] + acc[j][t][1] + acc[j][t][2] + acc[j][t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same 64 x 128 tile through v_mfma_f32_32x32x2_f32 (half the MFMA instructions, 16-pass): per 2-row step one 8-byte load (A: 64
// columns) and one 16-byte load (B: 128 columns) and 8 MFMAs; three register stages, loads spread between the MFMAs
typedef float v16f __attribute__((ext_vector_type(16)));
template <int PRIO, int LOADS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k32(const float* __restrict__ src, float* out, int iters, int mask) {
    v16f acc[2][4];
    for (int j = 0; j < 2; ++j) for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) acc[j][t][e] = 0.f;
    const float4* p = (const float4*)src + threadIdx.x;
    const float2* p2 = (const float2*)src + threadIdx.x;
    float2 a0, a1, a2; float4 x0, x1, x2;
    a0 = a1 = a2 = p2[0]; x0 = x1 = x2 = p[0];
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(A, X, J) do { const float av = (J) == 0 ? A.x : A.y; const float bv[4] = {X.x, X.y, X.z, X.w};                    \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                                 \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[J][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[J][t], 0, 0, 0);  \
        if (PRIO) __builtin_amdgcn_s_setprio(0); } while (0)
#define STEP(CA, CX, LA, LX, I) do { const int o = ((I) * 512) & mask;                                                       \
        MF(CA, CX, 0); SB(); if (LOADS) { LA = p2[2 * o]; } SB(); MF(CA, CX, 1); SB(); if (LOADS) { LX = p[o + 256]; } SB(); } while (0)
    int i = blockIdx.x;
    for (int it = 0; it < iters; it += 3, i += 3) {
        STEP(a0, x0, a2, x2, i + 2);
        STEP(a1, x1, a0, x0, i + 3);
        STEP(a2, x2, a1, x1, i + 4);
    }
#undef STEP
#undef MF
#undef SB
    float s = 0.f;
    for (int j = 0; j < 2; ++j) for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) s += acc[j][t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int PRIO, int LOADS> void run32(int blocks, int iters, int mask, const char* what) {
    float *out, *src; hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&src, ((size_t)mask + 1024) * 16 + (1 << 20)); hipMemset(src, 0, ((size_t)mask + 1024) * 16 + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k32<PRIO, LOADS>), dim3(blocks), dim3(256), 0, 0, (const float*)src, out, iters, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k32<PRIO, LOADS>), dim3(blocks), dim3(256), 0, 0, (const float*)src, out, iters, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 2;
    printf("%s: %.3f ms, %.1f TFLOP/s\n", what, ms, flop / ms / 1e9);
    hipFree(out); hipFree(src);
}
void runbig(int blocks, int iters, int mask, const char* what) {
    float *out, *src; hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&src, ((size_t)mask + 2048) * 16 + (1 << 20)); hipMemset(src, 0, ((size_t)mask + 2048) * 16 + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kbig, dim3(blocks), dim3(256), 0, 0, (const float*)src, out, iters, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kbig, dim3(blocks), dim3(256), 0, 0, (const float*)src, out, iters, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 64 * 2.0 * 16 * 16 * 4;
    printf("%s: %.3f ms, %.1f TFLOP/s\n", what, ms, flop / ms / 1e9);
    hipFree(out); hipFree(src);
}
template <int PRIO> void runspread(int blocks, int iters, int mask, const char* what) {
    float *out, *src; hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&src, ((size_t)mask + 1024) * 16 + (1 << 20)); hipMemset(src, 0, ((size_t)mask + 1024) * 16 + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kspread<PRIO>, dim3(blocks), dim3(256), 0, 0, (const float*)src, out, iters, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kspread<PRIO>, dim3(blocks), dim3(256), 0, 0, (const float*)src, out, iters, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 32 * 2.0 * 16 * 16 * 4;
    printf("%s: %.3f ms, %.1f TFLOP/s\n", what, ms, flop / ms / 1e9);
    hipFree(out); hipFree(src);
}
template <bool BARRIER, int NL = 3, int ADDR = 0> void runld(int blocks, int iters, int mask, const char* what) {
    float *out, *src; hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&src, ((size_t)mask + 1024) * 16 + (1 << 20)); hipMemset(src, 0, ((size_t)mask + 1024) * 16 + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kld<BARRIER, NL, ADDR>), dim3(blocks), dim3(256), 0, 0, (const float*)src, out, iters, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kld<BARRIER, NL, ADDR>), dim3(blocks), dim3(256), 0, 0, (const float*)src, out, iters, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 32 * 2.0 * 16 * 16 * 4;
    printf("%s: %.3f ms, %.1f TFLOP/s\n", what, ms, flop / ms / 1e9);
    hipFree(out); hipFree(src);
}
template <int MODE> void run48(int blocks, int iters, const char* what) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k48<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k48<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 32 * 2.0 * 16 * 16 * 4;
    printf("%s: %.3f ms, %.1f TFLOP/s\n", what, ms, flop / ms / 1e9);
    hipFree(out);
}
template <int NACC> void run(int blocks, int iters, const char* what) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * NACC * 2.0 * 16 * 16 * 4;
    printf("%s: blocks %d (x4 waves), %d independent accumulators, %d iters: %.3f ms, %.1f TFLOP/s\n", what, blocks, NACC, iters, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    run<32>(256, 4000, "1 wave/SIMD ");
    run<32>(512, 4000, "2 waves/SIMD");
    run<32>(1024, 4000, "4 waves/SIMD");
    run<4>(512, 32000, "2 waves/SIMD, 4 acc (dependent every 4)");
    run<32>(512, 40000, "2 waves/SIMD, long (sustained ~0.3 s)");
    run48<0>(512, 4000, "2 waves/SIMD, 4 A x 8 B registers, constant operands");
    run48<1>(512, 4000, "2 waves/SIMD, 4 A x 8 B registers, operands rewritten every step");
    run48<2>(512, 4000, "2 waves/SIMD, 4 A x 8 B registers, random operand bit patterns");
    run48<2>(512, 40000, "2 waves/SIMD, 4 A x 8 B registers, random operand bit patterns, sustained");
    runld<false>(512, 3999, 0x3ff, "2 waves/SIMD, 3 loads + 32 MFMAs per step, 16 KB window (L1/L2 hits)");
    runld<false>(512, 3999, 0xfffff, "2 waves/SIMD, 3 loads + 32 MFMAs per step, 16 MB window (L2 / MALL)");
    runld<true>(512, 3999, 0x3ff, "  same with sched_barrier between load groups and MFMA groups, 16 KB window");
    runld<true>(512, 3999, 0xfffff, "  same with sched_barrier between load groups and MFMA groups, 16 MB window");
    runld<true, 3, 1>(512, 3999, 0x3ff, "  3 loads per step, uniform base + 32-bit lane offset addressing");
    runld<true, 2, 1>(512, 3999, 0x3ff, "  2 loads per step, uniform base + 32-bit lane offset addressing");
    runld<true, 0>(512, 3999, 0x3ff, "  0 loads per step");
    runld<true, 1>(512, 3999, 0x3ff, "  1 load per step");
    runld<true, 2>(512, 3999, 0x3ff, "  2 loads per step");
    runspread<0>(512, 3999, 0x3ff, "  3 loads per step SPREAD between the MFMAs, 16 KB window");
    runspread<0>(512, 3999, 0xfffff, "  3 loads per step SPREAD between the MFMAs, 16 MB window");
    runspread<1>(512, 3999, 0x3ff, "  spread + s_setprio(1) around each MFMA group, 16 KB window");
    runspread<1>(512, 3999, 0xfffff, "  spread + s_setprio(1) around each MFMA group, 16 MB window");
    run32<0, 0>(512, 7998, 0x3ff, "  32x32x2: 8 MFMAs per 2-row step, no loads");
    run32<0, 1>(512, 7998, 0x3ff, "  32x32x2: 8 MFMAs + 2 loads per 2-row step, 16 KB window");
    run32<0, 1>(512, 7998, 0xfffff, "  32x32x2: 8 MFMAs + 2 loads per 2-row step, 16 MB window");
    run32<1, 1>(512, 7998, 0x3ff, "  32x32x2 + s_setprio, 16 KB window");
    run32<1, 1>(512, 7998, 0xfffff, "  32x32x2 + s_setprio, 16 MB window");
    runbig(256, 3999, 0xfffff, "  1 wave/SIMD, 128 x 128 tile: 4 loads spread over 64 MFMAs per step, 16 MB window");
    return 0;
}
