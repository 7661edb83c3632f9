// A/B required by the north star ("MFMA used only for the small dense Jacobian / mass-matrix contractions where rocprof shows it wins over
// scalar FMA"): the one dense contraction of kernel A that is matrix shaped, the Delassus build  W = J' Y^T  (R x R, R <= 36 constraint
// rows, inner dimension 18), as the kernel does it today -- lane i forms its row with 9 FMAs per column from register-resident J'_i and
// LDS-resident Y_j, exploiting that a row touches the base (6) and ONE leg (3) -- against v_mfma_f32_16x16x4_f32 on the zero-padded dense
// operands (inner dimension 20 = 5 steps, ceil(R / 16)^2 output tiles), including what the MFMA form needs around it: J' written to LDS as
// dense 20-vectors, and the D tiles transposed through LDS so that lane i ends up with row i in registers (what the Gauss-Seidel sweep
// consumes).  One wave per robot, 4096 robots (4 waves per SIMD: kernel A's occupancy), REP builds per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/delassus_mfma.hip -o tools/micro/delassus_mfma && tools/micro/delassus_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define MAXR 36
#define NV 18

struct Shared {
    float Y[48][NV];        // rows of M^-1 J^T in the kernel's sparse layout: [0..5] base part, [6 + 3 leg ..] own-leg part, zero elsewhere
    float Jd[48][20];       // MFMA form only: dense, zero-padded J' rows
    float Wt[48][49];       // MFMA form only: D tiles on their way to one row per lane (padded stride: no bank conflicts on the row reads)
};

// inputs per robot: R, and per row i: jb[6], jl[3], leg; Y rows.  Generated on the device from a hash so that no input traffic is timed.
__device__ __forceinline__ float hashf(unsigned a, unsigned b) {
    unsigned h = a * 2654435761u ^ (b + 0x9e3779b9u) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

template <int MODE /* 0 VALU rows, 1 MFMA */, int RT /* row tiles of 16: ceil(R / 16) */>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_delassus(float* out, int R, int reps) {
    __shared__ Shared sh;
    const int lane = threadIdx.x;
    const unsigned env = blockIdx.x;
    // ---- set-up (not representative of anything; the same for both modes)
    const int leg = lane % 5 - 1;                         // -1: a row on the base only
    float jb[6], jl[3];
    for (int k = 0; k < 6; ++k) jb[k] = lane < R ? hashf(env * 64 + lane, k) : 0.0f;
    for (int k = 0; k < 3; ++k) jl[k] = (lane < R && leg >= 0) ? hashf(env * 64 + lane, 8 + k) : 0.0f;
    const int lo = leg >= 0 ? 6 + 3 * leg : 6;
    for (int idx = lane; idx < 48 * NV; idx += 64) {
        const int r = idx / NV, c = idx - r * NV, rleg = r % 5 - 1;
        const bool own = c < 6 || (rleg >= 0 && c >= 6 + 3 * rleg && c < 9 + 3 * rleg);
        sh.Y[r][c] = (r < R && own) ? hashf(env * 64 + r, 16 + c) : 0.0f;
    }
    __syncthreads();
    float check = 0.0f;
    for (int rep = 0; rep < reps; ++rep) {
        float W[MAXR];
        if (MODE == 0) {
            // ---- today's form: 9 FMAs per column, the 6 base entries of Y_j are broadcast reads, the 3 leg entries depend on the lane's leg
#pragma unroll
            for (int j = 0; j < MAXR; ++j) {
                if (j < R) {              // wave-uniform
                    const float* Yj = sh.Y[j];
                    float w = 0.0f;
#pragma unroll
                    for (int k = 0; k < 6; ++k) w += jb[k] * Yj[k];
                    w += jl[0] * Yj[lo] + jl[1] * Yj[lo + 1] + jl[2] * Yj[lo + 2];
                    W[j] = w;
                } else W[j] = 0.0f;
            }
        } else {
            // ---- MFMA form.  1: dense J' rows to LDS (20 floats per row: 6 base, 12 leg slots, 2 pad)
            if (lane < 16 * RT) {
                float d[20];
#pragma unroll
                for (int c = 0; c < 20; ++c) d[c] = 0.0f;
#pragma unroll
                for (int k = 0; k < 6; ++k) d[k] = jb[k];
#pragma unroll
                for (int l = 0; l < 4; ++l)
#pragma unroll
                    for (int k = 0; k < 3; ++k) d[6 + 3 * l + k] = (leg == l) ? jl[k] : 0.0f;
#pragma unroll
                for (int c = 0; c < 20; c += 4) *(float4*)&sh.Jd[lane][c] = make_float4(d[c], d[c + 1], d[c + 2], d[c + 3]);
            }
            __syncthreads();
            // 2: RT x RT tiles, 5 k-steps: A[i = lane % 16][k = lane / 16] = J'[16 ti + i][4 s + k], B[k][j] = Y[16 tj + j][4 s + k]
            v4f acc[RT][RT];
#pragma unroll
            for (int ti = 0; ti < RT; ++ti)
#pragma unroll
                for (int tj = 0; tj < RT; ++tj) acc[ti][tj] = (v4f){0.f, 0.f, 0.f, 0.f};
            const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                float a[RT], b[RT];
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    a[t] = sh.Jd[16 * t + r16][4 * s + kq];
                    b[t] = (4 * s + kq < NV) ? sh.Y[16 * t + r16][4 * s + kq] : 0.0f;
                }
#pragma unroll
                for (int ti = 0; ti < RT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < RT; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
            }
            // 3: D[i = 4 (lane / 16) + r][j = lane % 16] of every tile to LDS, then one row per lane back into registers
#pragma unroll
            for (int ti = 0; ti < RT; ++ti)
#pragma unroll
                for (int tj = 0; tj < RT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sh.Wt[16 * ti + 4 * kq + r][16 * tj + r16] = acc[ti][tj][r];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < MAXR; ++j) W[j] = (j < 16 * RT && lane < 16 * RT) ? sh.Wt[lane][j] : 0.0f;
            __syncthreads();
        }
        // consume the row the way the sweep does (every entry, in registers); the rep index keeps the builds from being merged
        float c = 0.0f;
#pragma unroll
        for (int j = 0; j < MAXR; ++j) c = fmaf(W[j], (float)(j + 1 + rep), c);
        check += c;
        jb[0] += 1e-7f * check;          // the next build depends on this one (as sub-step n + 1 depends on n)
    }
    out[env * 64 + lane] = check;
}

template <int MODE, int RT>
static float run(float* out, int R, int reps, int iters, std::vector<float>* host) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < iters + 2; ++it) {
        if (it == 2) CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_delassus<MODE, RT>), dim3(4096), dim3(64), 0, 0, out, R, reps);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    if (host) { host->resize(4096 * 64); CK(hipMemcpy(host->data(), out, host->size() * 4, hipMemcpyDeviceToHost)); }
    return ms / iters;
}

int main() {
    float* out; CK(hipMalloc(&out, 4096 * 64 * 4));
    const int reps = 64, iters = 20;
    printf("{\"waves\": 4096, \"waves_per_simd\": 4, \"builds_per_launch\": %d, \"runs\": [\n", reps);
    const int Rs[] = {6, 9, 15, 16, 24, 32, 36};
    for (int R : Rs) {
        std::vector<float> hv, hm, h1;
        // one build (reps = 1) for the value comparison: with more, the feedback term amplifies rounding differences
        run<0, 3>(out, R, 1, 1, &hv);
        float t_valu = run<0, 3>(out, R, reps, iters, nullptr);
        float t_mfma;
        if (R <= 16) { run<1, 1>(out, R, 1, 1, &hm); t_mfma = run<1, 1>(out, R, reps, iters, nullptr); }
        else if (R <= 32) { run<1, 2>(out, R, 1, 1, &hm); t_mfma = run<1, 2>(out, R, reps, iters, nullptr); }
        else { run<1, 3>(out, R, 1, 1, &hm); t_mfma = run<1, 3>(out, R, reps, iters, nullptr); }
        double err = 0, mag = 0;
        for (size_t i = 0; i < hv.size(); ++i) { err = fmax(err, fabs((double)hv[i] - hm[i])); mag = fmax(mag, fabs((double)hv[i])); }
        printf("  {\"rows\": %d, \"valu_us_per_build\": %.3f, \"mfma_us_per_build\": %.3f, \"mfma_over_valu\": %.2f, \"max_rel_diff\": %.2g}%s\n",
               R, t_valu * 1e3 / reps, t_mfma * 1e3 / reps, t_mfma / t_valu, err / (mag + 1e-30), R == 36 ? "" : ",");
        fflush(stdout);
    }
    printf("]}\n");
    return 0;
}
