// ls_math.h -- small fixed-size math for the leggedsim kernels (fp32).
// Compiles as device code under hipcc and as plain C++ under g++ (tests/emu lane emulator).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LS_FN __host__ __device__ __forceinline__
#else
#define LS_FN static inline
#endif

// Every simulator buffer lives in device global memory.  The pointers come out of a table (cx.buf[]), so the compiler cannot know their
// address space and would emit FLAT loads / stores, which count in lgkmcnt as well as vmcnt: every LDS wait (each phase boundary) would
// then also wait for all outstanding global stores.  Casting to the global address space gives global_load / global_store (vmcnt only).
#if defined(__HIP_DEVICE_COMPILE__)
#define LS_GLOBAL __attribute__((address_space(1)))
#else
#define LS_GLOBAL
#endif
#define LS_G(T, ptr) ((LS_GLOBAL T*)(ptr))

// lane-strided loop over N items with a compile-time trip count: `for (k = lane; k < N; k += 64)` has a per-lane trip count, which the
// compiler turns into a vectorised monster with 64-bit address arithmetic per element; this form unrolls into ceil(N / 64) predicated bodies
#if defined(LS_EMU) || !defined(__HIPCC__)
#define LS_STRIDED(k, lane, N) for (int k = (lane); k < (N); k += 64)
#else
#define LS_STRIDED(k, lane, N) _Pragma("unroll") for (int it_ = 0, k = (lane); it_ < ((N) + 63) / 64; ++it_, k += 64) if (k < (N))
#endif

// The synchronisation between two phases of a ONE-WAVE workgroup (kernels A / B: __launch_bounds__(64), one robot per wave).  __syncthreads() is a
// workgroup-scope release + acquire: the compiler drops the s_barrier for a single wave but keeps `s_waitcnt lgkmcnt(0)` -- every LDS write of the
// phase acknowledged before the next phase issues anything.  A wave's LDS instructions execute in issue order, so between lanes of ONE wave that
// wait buys nothing: a wavefront-scope fence orders the memory operations for the compiler and emits no instruction, and the next phase's address
// arithmetic may start while the writes drain.  Measured (round 6, interleaved, profiles/r06_kernel_a_ab.txt): kernel A 0.1097 -> 0.1094 ms flat,
// 0.1200 -> 0.1186 ms stairs; 34 of 730 lgkmcnt(0) waits gone (most phases begin with an LDS read, which has to wait for its data anyway).
// -DLS_NO_WAVE_FENCE: the __syncthreads() form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LS_NO_WAVE_FENCE)
#define LS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define LS_WAVE_SYNC() __syncthreads()
#endif

// a value every lane of the wave holds identically (read from LDS, so the compiler cannot know): moved to a scalar register, which turns
// branches and loop bounds on it into scalar compares instead of per-lane compares with hoisted lane masks
#if defined(__HIP_DEVICE_COMPILE__)
#define LS_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define LS_UNIFORM(x) (x)
#endif

// One float of a table that no kernel writes, at an address that is the same for every active lane, through the SCALAR data cache: the
// request counts in lgkmcnt, not vmcnt, so it does not queue behind the vector stores the wave has in flight (vmcnt retires in order: a
// vector load late in a kernel waits for every store before it).  Not for data a kernel of the step writes: the scalar cache is not
// coherent with vector stores.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float ls_uniform_load(LS_GLOBAL const float* p) {
    const unsigned long long a = (unsigned long long)p;
    // (readfirstlane returns a SIGNED int: widen through unsigned, or the low word's bit 31 floods the high word)
    const unsigned long long u = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a);
    float v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(u) : "memory");
    return v;
}
// three consecutive floats the same way, one wait
__device__ __forceinline__ void ls_uniform_load3(LS_GLOBAL const float* p, float out[3]) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned long long u = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a);
    float v0, v1, v2;
    asm volatile("s_load_dword %0, %3, 0x0\n\ts_load_dword %1, %3, 0x4\n\ts_load_dword %2, %3, 0x8\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v0), "=&s"(v1), "=&s"(v2) : "s"(u) : "memory");
    out[0] = v0; out[1] = v1; out[2] = v2;
}
#else
static inline float ls_uniform_load(const float* p) { return *p; }
static inline void ls_uniform_load3(const float* p, float out[3]) { out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; }
#endif

struct alignas(8) LsF2 { float x, y; };   // one 8-byte global store
// float -> int as the GPU converts it (v_cvt_i32_f32: NaN -> 0, out-of-range values saturate); the C++ cast is undefined there and x86 returns
// INT_MIN, with which the lane emulator indexed the height grid when a robot's state was NaN (the GPU did not: tests/test_nonfinite_counter.py)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int ls_f2i(float v) { return (int)v; }
#else
static inline int ls_f2i(float v) { return v != v ? 0 : (v >= 2147483648.0f ? 2147483647 : (v <= -2147483648.0f ? (-2147483647 - 1) : (int)v)); }
#endif
LS_FN unsigned int ls_float_bits(float v) { union { float f; unsigned int u; } c; c.f = v; return c.u; }

struct V3 {
    float x, y, z;
};
LS_FN V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
LS_FN V3 v3p(const float* p) { V3 r = {p[0], p[1], p[2]}; return r; }
LS_FN void v3st(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
LS_FN V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
LS_FN V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
LS_FN V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
LS_FN V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
LS_FN float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LS_FN V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
LS_FN float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
// The simulator is built with -fno-hip-fp32-correctly-rounded-divide-sqrt (fast v_rcp / v_rsq based division in the dynamics);
// quotients that feed an integer truncation of the reference (grid indices, LR:1343) must stay IEEE-exact.  Under that flag
// __fdiv_rn / __fsqrt_rn are NOT exact either -- __clang_hip_math.h defines them as `x / y` and the native square root unless
// OCML_BASIC_ROUNDED_OPERATIONS is set, which round 2 relied on by mistake (an N = 4096 step put ~1 of its 2.3 M height samples into the
// neighbouring grid cell).  The fp64 operations are not affected by the flag, and the fp64 quotient / root of fp32 operands rounded to
// fp32 IS the correctly rounded fp32 result (53 >= 2 * 24 + 2 bits: no double rounding).
LS_FN float ls_div_exact(float a, float b) { return (float)((double)a / (double)b); }
LS_FN float ls_sqrt_exact(float a) { return (float)sqrt((double)a); }

// Hardware reciprocal / reciprocal square root (v_rcp_f32 / v_rsq_f32, 1 ulp) for the dynamics: a plain `a / b` compiles to an 8-instruction
// range-safe sequence (frexp / rcp / ldexp) even with fast division enabled, and the physics does ~60 of them per sub-step.  The
// operands here are O(1e-6 .. 1e4) by construction (guarded by fmaxf where they could vanish); the lane emulator and the oracle divide.
LS_FN float ls_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
LS_FN float ls_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}

// row-major 3x3
struct M3 {
    float m[9];
};
LS_FN V3 mul(const M3& R, V3 v) {
    return v3(R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z, R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z);
}
LS_FN M3 mul(const M3& A, const M3& B) {
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
LS_FN M3 m3p(const float* p) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
LS_FN void m3st(float* p, const M3& a) { for (int i = 0; i < 9; ++i) p[i] = a.m[i]; }

LS_FN M3 quat_to_R(const float* q) {  // xyzw
    float x = q[0], y = q[1], z = q[2], w = q[3];
    M3 R;
    R.m[0] = 1 - 2 * (y * y + z * z); R.m[1] = 2 * (x * y - z * w); R.m[2] = 2 * (x * z + y * w);
    R.m[3] = 2 * (x * y + z * w); R.m[4] = 1 - 2 * (x * x + z * z); R.m[5] = 2 * (y * z - x * w);
    R.m[6] = 2 * (x * z - y * w); R.m[7] = 2 * (y * z + x * w); R.m[8] = 1 - 2 * (x * x + y * y);
    return R;
}
LS_FN void R_to_quat(const M3& Rm, float* q) {
    const float* R = Rm.m;
    float tr = R[0] + R[4] + R[8];
    // same case selection as the textbook form (trace > 0, else the largest diagonal entry), but one rsqrt and the same products for
    // every case instead of a division per component
    const int m = (tr > 0) ? 0 : ((R[0] > R[4] && R[0] > R[8]) ? 1 : (R[4] > R[8] ? 2 : 3));
    const float dg = (m == 1) ? R[0] : (m == 2 ? R[4] : R[8]);
    const float t = (m == 0) ? tr + 1.0f : 1.0f + 2.0f * dg - tr;          // 4 * (selected component)^2
    const float is = 0.5f * ls_rsqrt(t);                                     // 1 / (4 * largest component)
    const float big = t * is;                                                // the largest component itself
    const float a = (R[7] - R[5]) * is, b = (R[2] - R[6]) * is, c = (R[3] - R[1]) * is;      // antisymmetric part: w * (x, y, z) / big
    const float d = (R[1] + R[3]) * is, e = (R[2] + R[6]) * is, f = (R[5] + R[7]) * is;      // symmetric part:   (xy, xz, yz) / big
    // selects, not stores through a computed index (that would put q into scratch memory on the GPU)
    q[0] = (m == 1) ? big : (m == 0 ? a : (m == 2 ? d : e));
    q[1] = (m == 2) ? big : (m == 0 ? b : (m == 1 ? d : f));
    q[2] = (m == 3) ? big : (m == 0 ? c : (m == 1 ? e : f));
    q[3] = (m == 0) ? big : (m == 1 ? a : (m == 2 ? b : c));
}
// joint angles are a few radians at most: the hardware v_sin_f32 / v_cos_f32 (abs error ~1e-6) replace ocml's ~60-instruction
// range-reduced sinf / cosf on the device; the lane emulator and the oracle use libm
LS_FN void ls_sincos_joint(float th, float& s, float& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    s = __sinf(th); c = __cosf(th);
#else
    s = sinf(th); c = cosf(th);
#endif
}
LS_FN M3 axis_angle_R(V3 a, float th) {
    float c, s;
    ls_sincos_joint(th, s, c);
    float t = 1 - c;
    M3 R;
    R.m[0] = c + t * a.x * a.x; R.m[1] = t * a.x * a.y - s * a.z; R.m[2] = t * a.x * a.z + s * a.y;
    R.m[3] = t * a.x * a.y + s * a.z; R.m[4] = c + t * a.y * a.y; R.m[5] = t * a.y * a.z - s * a.x;
    R.m[6] = t * a.x * a.z - s * a.y; R.m[7] = t * a.y * a.z + s * a.x; R.m[8] = c + t * a.z * a.z;
    return R;
}

// ---- the reference's quaternion helpers (isaacgym.torch_utils / legged_gym.utils.math), same op order as the torch code ----
LS_FN V3 quat_rotate_inverse(const float* q, V3 v) {  // torch_utils.quat_rotate_inverse
    float w = q[3];
    V3 qv = v3(q[0], q[1], q[2]);
    float s = 2.0f * (w * w) - 1.0f;
    V3 c = cross(qv, v);
    float d = dot(qv, v);
    return v3(v.x * s - c.x * w * 2.0f + qv.x * d * 2.0f, v.y * s - c.y * w * 2.0f + qv.y * d * 2.0f, v.z * s - c.z * w * 2.0f + qv.z * d * 2.0f);
}
LS_FN V3 quat_apply(const float* q, V3 v) {  // torch_utils.quat_apply
    V3 qv = v3(q[0], q[1], q[2]);
    V3 t = cross(qv, v) * 2.0f;
    V3 u = cross(qv, t);
    return v3(v.x + q[3] * t.x + u.x, v.y + q[3] * t.y + u.y, v.z + q[3] * t.z + u.z);
}
LS_FN V3 quat_apply_yaw(const float* q, V3 v) {  // MTH:38-42
    float n = sqrtf(q[2] * q[2] + q[3] * q[3]);
    if (n < 1e-9f) n = 1e-9f;
    float qy[4] = {0.0f, 0.0f, q[2] / n, q[3] / n};
    return quat_apply(qy, v);
}
LS_FN float wrap_to_pi(float a) {  // MTH:45-48 with torch.remainder semantics
    const float two_pi = 6.2831855f;
    float r = fmodf(a, two_pi);
    if (r != 0.0f && r < 0.0f) r += two_pi;
    if (r > 3.1415927f) r -= two_pi;
    return r;
}
LS_FN void quat_from_euler_xyz(float roll, float pitch, float yaw, float* q) {
    float cy = cosf(yaw * 0.5f), sy = sinf(yaw * 0.5f), cr = cosf(roll * 0.5f), sr = sinf(roll * 0.5f);
    float cp = cosf(pitch * 0.5f), sp = sinf(pitch * 0.5f);
    q[3] = cy * cr * cp + sy * sr * sp;
    q[0] = cy * sr * cp - sy * cr * sp;
    q[1] = cy * cr * sp + sy * sr * cp;
    q[2] = sy * cr * cp - cy * sr * sp;
}

// ---- spatial (Pluecker) algebra, [angular; linear], world-aligned axes about the base origin ----
struct S6 {
    V3 a, l;
};
LS_FN S6 s6(V3 a, V3 l) { S6 r = {a, l}; return r; }
LS_FN S6 s6p(const float* p) { return s6(v3p(p), v3p(p + 3)); }
LS_FN void s6st(float* p, S6 v) { v3st(p, v.a); v3st(p + 3, v.l); }
LS_FN S6 operator+(S6 x, S6 y) { return s6(x.a + y.a, x.l + y.l); }
LS_FN S6 operator-(S6 x, S6 y) { return s6(x.a - y.a, x.l - y.l); }
LS_FN S6 operator*(S6 x, float s) { return s6(x.a * s, x.l * s); }
LS_FN float dot(S6 x, S6 y) { return dot(x.a, y.a) + dot(x.l, y.l); }
LS_FN S6 crm(S6 v, S6 m) { return s6(cross(v.a, m.a), cross(v.a, m.l) + cross(v.l, m.a)); }   // motion cross product
LS_FN S6 crf(S6 v, S6 f) { return s6(cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)); }   // force cross product
LS_FN S6 m6v(const float* I, S6 v) {  // row-major 6x6 times spatial vector
    float x[6] = {v.a.x, v.a.y, v.a.z, v.l.x, v.l.y, v.l.z}, o[6];
    for (int i = 0; i < 6; ++i) {
        float s = 0;
        for (int k = 0; k < 6; ++k) s += I[6 * i + k] * x[k];
        o[i] = s;
    }
    return s6(v3(o[0], o[1], o[2]), v3(o[3], o[4], o[5]));
}

// ---- Philox4x32-10 (include/lsim.h RNG spec) ----
LS_FN uint32_t ls_mulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
LS_FN void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = ls_mulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        uint32_t hi1 = ls_mulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
LS_FN float u32_to_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// all four uniforms of block (idx >> 2)
LS_FN void ls_u01x4(uint32_t seed, uint32_t rank, uint32_t env, uint32_t step, uint32_t tag, uint32_t block, float out[4]) {
    uint32_t c[4] = {env, step, tag, block};
    philox4x32_10(c, seed, rank);
    for (int i = 0; i < 4; ++i) out[i] = u32_to_u01(c[i]);
}
LS_FN float ls_u01(uint32_t seed, uint32_t rank, uint32_t env, uint32_t step, uint32_t tag, uint32_t idx) {
    uint32_t c[4] = {env, step, tag, idx >> 2};
    philox4x32_10(c, seed, rank);
    return u32_to_u01(c[idx & 3u]);
}
// isaacgym.torch_utils.torch_rand_float: span formed on the host in double, then fp32 (upper-lower)*u + lower
LS_FN float rand_range(float u, float lo, float hi) { return (hi - lo) * u + lo; }
