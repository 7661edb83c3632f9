// VALU issue rate of gfx950 as kernel A sees it: wave64 v_fma_f32 chains, independent (8 accumulators per lane) and dependent
// (1 accumulator), at 1 / 2 / 4 / 8 waves per SIMD.  Reports wave-instructions per shader cycle per SIMD (s_memtime ticks), the
// effective shader clock (s_memtime ticks per second of the 100 MHz wall counter), and the rate by wall time.  Settles the "2 or 4
// cycles per wave64 VALU instruction" question of VERDICT r1 (MI355X_MICROARCH.md: SIMD-32, 2 cycles).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_peak.hip -o tools/micro/valu_peak ; run: tools/micro/valu_peak > out.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NCHAIN>
__global__ __launch_bounds__(256) void k_fma(float* out, unsigned long long* ticks, int iters, float a, float b) {
    float x[NCHAIN];
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) x[i] = (float)threadIdx.x * 1e-3f + (float)i;
    const unsigned long long t0 = __builtin_readcyclecounter();   // s_memtime: shader-clock ticks
    const unsigned long long w0 = wall_clock64();                  // constant 100 MHz
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / NCHAIN; ++r) {
#pragma unroll
            for (int i = 0; i < NCHAIN; ++i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        ticks[2 * w] = t1 - t0;
        ticks[2 * w + 1] = w1 - w0;
    }
}

template <int NCHAIN>
static int run(int waves_per_simd, int iters, const char* what, bool last) {
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves = one per SIMD) x waves_per_simd
    float* out; unsigned long long* ticks;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CK(hipMalloc(&ticks, (size_t)blocks * 4 * 2 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_fma<NCHAIN>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters / 8, 1.0001f, 1e-6f);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_fma<NCHAIN>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters, 1.0001f, 1e-6f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
    CK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int w = 0; w < blocks * 4; ++w) { cyc += (double)h[2 * w]; wall += (double)h[2 * w + 1]; }
    cyc /= blocks * 4; wall /= blocks * 4;                       // mean per wave
    const double insts_per_wave = (double)iters * 64.0;
    const double clock_hz = cyc / (wall / 100e6);
    // Headline: cycles per wave64 instruction per SIMD from the WALL rate of the whole launch (1024 SIMDs): independent of how the blocks
    // were placed.  The in-wave tick figure assumes all waves_per_simd waves of a SIMD are co-resident for the whole `cyc`, which the
    // dispatcher does not guarantee (round 2's table was off by ~2x at 4-8 waves where the blocks ran in two rounds): it is
    // kept as "cycles_per_wave_inst_if_coresident" and is only meaningful where kernel_ms ~ cyc / clock.
    const double wall_rate = insts_per_wave * blocks * 4 / (ms * 1e-3);      // wave-instructions per second, whole chip
    const double cyc_wall = clock_hz * 1024.0 / wall_rate;
    const double ipc_simd = insts_per_wave * waves_per_simd / cyc;
    const double resident_frac = (cyc / clock_hz) / (ms * 1e-3);              // share of the launch one wave was alive for
    printf("  {\"test\": \"%s\", \"chains_per_lane\": %d, \"waves_per_simd\": %d, \"cycles_per_wave_inst\": %.3f, "
           "\"cycles_per_wave_inst_if_coresident\": %.3f, \"wave_alive_frac_of_launch\": %.3f, \"effective_clock_ghz\": %.3f, "
           "\"kernel_ms\": %.3f, \"chip_wave_insts_per_s\": %.4g, \"fp32_tflops\": %.1f}%s\n",
           what, NCHAIN, waves_per_simd, cyc_wall, 1.0 / ipc_simd, resident_frac, clock_hz * 1e-9, ms, wall_rate, wall_rate * 128.0 * 1e-12,
           last ? "" : ",");
    CK(hipFree(out)); CK(hipFree(ticks));
    return 0;
}

int main() {
    printf("{\"device\": \"gfx950\", \"instruction\": \"v_fma_f32 (wave64)\", \"note\": \"cycles = s_memtime ticks per wave; clock = ticks / wall_clock64 (100 MHz)\", \"runs\": [\n");
    const int it = 20000;
    if (run<8>(1, it, "independent", false)) return 1;
    if (run<8>(2, it, "independent", false)) return 1;
    if (run<8>(4, it, "independent", false)) return 1;
    if (run<8>(8, it, "independent", false)) return 1;
    if (run<1>(1, it, "dependent", false)) return 1;
    if (run<1>(2, it, "dependent", false)) return 1;
    if (run<1>(4, it, "dependent", false)) return 1;
    if (run<1>(8, it, "dependent", false)) return 1;
    if (run<2>(4, it, "two chains", false)) return 1;
    if (run<4>(4, it, "four chains", true)) return 1;
    printf("]}\n");
    return 0;
}
