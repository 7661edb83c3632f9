// Sustained fp32 MFMA rate of the device (v_mfma_f32_16x16x4_f32, the instruction of the learner-side kernels): WAVES waves per SIMD
// issuing independent MFMAs from registers only.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
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
    return 0;
}
