// ls_physics.h -- one 5 ms articulated-body sub-step of one robot, executed by one wavefront.
// Replaces gym.simulate()/fetch_results()/refresh_dof_state_tensor() (LR:149-152, PhysX in the reference).
//
// Algorithm (same model as oracle/orc_physics.c, different -- structured -- solver):
//   the floating base couples four independent 3-dof chains, so
//     M = [ Mbb  Mb1 .. Mb4 ]      Schur complement on the base:  Sb = Mbb - sum_l Mbl Mll^-1 Mlb
//         [ M1b  M11        ]      every solve with M costs four 3x3 and one 6x6 Cholesky solve.
//         [ ..        ..    ]
//   All spatial quantities are expressed in world-aligned axes about the base origin, so composite
//   inertias and bias forces accumulate up the tree by plain addition (no frame transforms).
// Lane roles per phase:  K: leg chain (lanes 0-3)   B: body (0-16)   L: leg (0-3)   E: matrix element (0-41)
//                        P: collision point (0-55)  R: constraint row (0-59)        V: generalized velocity (0-17)
#pragma once
#include <stddef.h>

#include "ls_shared.h"

LS_FN int ls_leg_of_body(int b) { return (b - 1) >> 2; }   // b >= 1
// Constraint-row slots (lane = slot): contact k owns slots 3k .. 3k+2 (normal, t1, t2), joint-limit row i owns slot LS_LIM0 + i.
// Fixed slots give every slot a static row type, so the Gauss-Seidel sweep is specialised per type with no per-row masks or selects.
#define LS_LIM0 (3 * LS_MAXC)
LS_FN bool ls_slot_active(const WaveShared& sh, int slot) { return slot < 3 * sh.nc || (slot >= LS_LIM0 && slot < LS_LIM0 + sh.nlim); }
LS_FN int ls_depth_of_body(int b) { return (b - 1) & 3; }  // 0 hip, 1 thigh, 2 calf, 3 foot

// ---- phase K: forward kinematics, motion subspaces, twists, bias accelerations (lane = leg; lane 4 = base)
LS_FN void ph_kinematics(WaveShared& sh, int lane) {
    if (lane == 4) {
        M3 R0 = quat_to_R(sh.root + 3);
        m3st(sh.R[0], R0);
        sh.p[0][0] = sh.p[0][1] = sh.p[0][2] = 0.0f;
        s6st(sh.V[0], s6(v3p(sh.root + 10), v3p(sh.root + 7)));
        s6st(sh.Ab[0], s6(v3(0, 0, 0), v3(0, 0, 0)));
    }
    if (lane >= 4) return;
    M3 R = quat_to_R(sh.root + 3);
    V3 p = v3(0, 0, 0);
    S6 V = s6(v3p(sh.root + 10), v3p(sh.root + 7));
    S6 Ab = s6(v3(0, 0, 0), v3(0, 0, 0));
    for (int k = 0; k < 4; ++k) {
        int b = 1 + 4 * lane + k;
        const LsBodyLds& bd = sh.body[b];
        p = p + mul(R, v3p(bd.jpos));
        if (k < 3) {
            int d = 3 * lane + k;
            V3 ax = v3p(bd.axis);
            V3 aw = mul(R, ax);
            R = mul(R, axis_angle_R(ax, sh.q[d]));
            S6 S = s6(aw, cross(p, aw));
            s6st(sh.S[d], S);
            S6 vj = S * sh.qd[d];
            V = V + vj;
            Ab = Ab + crm(V, vj);
        }
        m3st(sh.R[b], R);
        v3st(sh.p[b], p);
        s6st(sh.V[b], V);
        s6st(sh.Ab[b], Ab);
    }
}

#if !defined(LS_EMU)
// ---- wave collective: the same kinematics, one MATRIX ELEMENT per lane instead of one leg per lane (the per-leg form above runs a 590-
//      instruction program on 4 of the 64 lanes).  lane = 16 * leg + 4 * iq + j: row iq and column j of the 3 x 3 rotation (iq = 3 repeats
//      row 0, j = 3 idles), so a matrix row lives in one quad and a vector component in one quad of the leg's 16-lane DPP row:
//        child(i, j) = c par(i, j) + (1 - c) a_j (x . a) + s (x x a)_j,  x = row i of the parent     Rodrigues' formula for one element; the
//                                                            row comes in by three quad_perm broadcasts, no LDS round trip, any joint axis a
//        cross(u, v)_i = u_{i+1} v_{i+2} - u_{i+2} v_{i+1}   the neighbouring components by row rotations (three components on four quads: quad 3
//                                                            copies component 0 so that quad 2 finds its successor; what quad 3 itself
//                                                            derives from rotations is wrong and is repaired from quad 0 where it is re-used)
//      Results are those of ph_kinematics up to fp32 summation order (tests: HIP against the oracle, tests/test_gpu_parity.py).
template <int CTRL> __device__ __forceinline__ float ls_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float ls_rot1(float v) { return ls_dpp<0x120 + 12>(v); }   // component i + 1: from lane + 4 (row_ror:12)
__device__ __forceinline__ float ls_rot2(float v) {                                    // component i + 2 = i - 1 (mod 3): from lane - 4 (row_ror:4);
    const float t = ls_dpp<0x120 + 4>(v);                                              // quad 0 would read quad 3 (the copy of component 0): it takes
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(t), __float_as_int(v), 0x120 + 8, 0xF, 0x1, false));   // component 2 from lane + 8
}
__device__ __forceinline__ float ls_fix_quad3(float v) {                               // quad 3 <- quad 0 of the same row
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x120 + 12, 0xF, 0x8, false));
}
__device__ __forceinline__ float ls_cross_i(float a1, float a2, float b1, float b2) { return fmaf(a1, b2, -(a2 * b1)); }

LS_FN void wc_kinematics(WaveShared& sh, int lane) {
    const int l = lane >> 4, iq = (lane >> 2) & 3, j = lane & 3;
    const int i = iq == 3 ? 0 : iq, jj = j < 3 ? j : 0;
    const bool elem = iq < 3 && j < 3, comp = iq < 3 && j == 0;
    // per-lane bases: body 1 + 4 l of every array, so that level k is a compile-time offset
    const LsBodyLds* bl = &sh.body[1 + 4 * l];
    float* Rl = &sh.R[1 + 4 * l][3 * i + jj];
    float* pl = &sh.p[1 + 4 * l][i];
    float* Vl_ = &sh.V[1 + 4 * l][i];
    float* Al_ = &sh.Ab[1 + 4 * l][i];
    float* Sl = &sh.S[3 * l][i];
    const float* ql = &sh.q[3 * l];
    const float* qdl = &sh.qd[3 * l];
    // base rotation element R0(i, j) from the quaternion (x, y, z, w): diag 1 - 2 (|v|^2 - v_i^2), off-diag 2 (v_i v_j - w eps_ijk v_k)
    const float qx = sh.root[3], qy = sh.root[4], qz = sh.root[5], qw = sh.root[6];
    const int kk = (i == jj) ? 0 : 3 - i - jj;
    const float vi = sh.root[3 + i], vj = sh.root[3 + jj], vk = sh.root[3 + kk];
    const float n2 = qx * qx + qy * qy + qz * qz;
    const float weps = ((jj - i + 3) % 3 == 1) ? qw : -qw;
    float Re = (i == jj) ? 1.0f - 2.0f * (n2 - vi * vi) : 2.0f * (vi * vj - weps * vk);
    if (j == 3) Re = 0.0f;
    float pe = 0.0f;                                   // component i of the body origin (relative to the base origin, world axes)
    float Va = sh.root[10 + i], Vl = sh.root[7 + i];   // twist (angular, linear at the base origin)
    float Aa = 0.0f, Al = 0.0f;                        // bias acceleration
    if (l == 0) {                                      // body 0
        if (elem) sh.R[0][3 * i + j] = Re;
        if (comp) { sh.p[0][i] = 0.0f; sh.V[0][i] = Va; sh.V[0][3 + i] = Vl; sh.Ab[0][i] = 0.0f; sh.Ab[0][3 + i] = 0.0f; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x0 = ls_dpp<0x00>(Re), x1 = ls_dpp<0x55>(Re), x2 = ls_dpp<0xAA>(Re);   // row i of the parent rotation
        pe = fmaf(x0, bl[k].jpos[0], fmaf(x1, bl[k].jpos[1], fmaf(x2, bl[k].jpos[2], pe)));
        float aw = 0.0f, c = 0.0f;
        if (k < 3) {
            const float a0 = bl[k].axis[0], a1 = bl[k].axis[1], a2 = bl[k].axis[2];
            aw = fmaf(x0, a0, fmaf(x1, a1, x2 * a2));                                       // (x . a): component i of the axis in world axes
            const float w0 = ls_cross_i(x1, x2, a1, a2), w1 = ls_cross_i(x2, x0, a2, a0), w2 = ls_cross_i(x0, x1, a0, a1);   // x x a
            const float aj = jj == 0 ? a0 : (jj == 1 ? a1 : a2), wj = jj == 0 ? w0 : (jj == 1 ? w1 : w2);
            float sn, cs;
            ls_sincos_joint(ql[k], sn, cs);
            Re = fmaf(cs, Re, fmaf((1.0f - cs) * aj, aw, sn * wj));
            if (j == 3) Re = 0.0f;
            c = ls_cross_i(ls_rot1(pe), ls_rot2(pe), ls_rot1(aw), ls_rot2(aw));             // (p x aw)_i
            c = ls_fix_quad3(c);
            const float qd = qdl[k];
            const float ja = aw * qd, jl = c * qd;                                           // joint twist S qd
            Va += ja; Vl += jl;
            const float Va1 = ls_rot1(Va), Va2 = ls_rot2(Va), Vl1 = ls_rot1(Vl), Vl2 = ls_rot2(Vl);
            const float ja1 = ls_rot1(ja), ja2 = ls_rot2(ja), jl1 = ls_rot1(jl), jl2 = ls_rot2(jl);
            Aa += ls_cross_i(Va1, Va2, ja1, ja2);                                            // crm(V, vj): (w x ja ; w x jl + v x ja)
            Al += ls_cross_i(Va1, Va2, jl1, jl2) + ls_cross_i(Vl1, Vl2, ja1, ja2);
        }
        if (elem) Rl[9 * k] = Re;
        if (comp) {
            if (k < 3) { Sl[6 * k] = aw; Sl[6 * k + 3] = c; }
            pl[3 * k] = pe; Vl_[6 * k] = Va; Vl_[6 * k + 3] = Vl; Al_[6 * k] = Aa; Al_[6 * k + 3] = Al;
        }
    }
}
#endif

// ---- phase B: spatial inertia about the base origin and bias force of each body (lane = body)
LS_FN void ph_body_inertia(const LsCtx& cx, WaveShared& sh, int lane, bool apply_force) {
    if (lane >= LS_NB) return;
    const LsBodyLds& bd = sh.body[lane];
    float mass = bd.mass;
    V3 cl = v3p(bd.com);
    float sc = 1.0f;
    if (lane == 0) {  // payload / COM randomisation (LR:591-596); inertia scaled with the mass
        mass = bd.mass + sh.payload;
        cl = cl + v3p(sh.comd);
        sc = mass * ls_rcp(bd.mass);
    }
    M3 R = m3p(sh.R[lane]);
    V3 c = mul(R, cl) + v3p(sh.p[lane]);
    M3 Il;
    Il.m[0] = bd.inertia[0] * sc; Il.m[1] = bd.inertia[1] * sc; Il.m[2] = bd.inertia[2] * sc;
    Il.m[3] = Il.m[1]; Il.m[4] = bd.inertia[3] * sc; Il.m[5] = bd.inertia[4] * sc;
    Il.m[6] = Il.m[2]; Il.m[7] = Il.m[5]; Il.m[8] = bd.inertia[5] * sc;
    M3 Rt;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt.m[3 * i + j] = R.m[3 * j + i];
    M3 Iw = mul(mul(R, Il), Rt);
    float* I6 = sh.u.I6[lane];
    float cx9[9] = {0, -c.z, c.y, c.z, 0, -c.x, -c.y, c.x, 0};
    float cv[3] = {c.x, c.y, c.z};
    float cc = dot(c, c);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            I6[6 * i + j] = Iw.m[3 * i + j] + mass * ((i == j ? cc : 0.0f) - cv[i] * cv[j]);
            I6[6 * i + 3 + j] = mass * cx9[3 * i + j];
            I6[6 * (3 + i) + j] = mass * cx9[3 * j + i];
            I6[6 * (3 + i) + 3 + j] = (i == j) ? mass : 0.0f;
        }
#if LS_I6_STRIDE > 36
    // padding floats of the row: they lie inside constraint rows Y that the solver may read as "holds no row yet" slots (times a zero impulse): finite
    for (int k = 36; k < LS_I6_STRIDE; ++k) I6[k] = 0.0f;
#endif
    S6 V = s6p(sh.V[lane]);
    S6 a = s6p(sh.Ab[lane]) - s6(v3(0, 0, 0), v3p(cx.cfg.gravity));
    S6 F = m6v(I6, a) + crf(V, m6v(I6, V));
    if (lane == 0) {
        v3st(sh.com0, c);
        if (apply_force) {  // body-local disturbance force at the base COM (LR:844)
            V3 fw = mul(R, v3p(sh.pend));
            F = F - s6(cross(c, fw), fw);
        }
    }
    s6st(sh.Fb[lane], F);
}

// 3x3 SPD Cholesky and solve.  L = (l00, l10, l11, l20, l21, l22, 1/l00, 1/l11, 1/l22): the reciprocal diagonal comes out of the
// rsqrt that forms the factor, so no solve ever divides
#define LS_CHOL3 9
LS_FN void chol3(const float* M /*m00,m10,m11,m20,m21,m22*/, float* L) {
    const float r0 = ls_rsqrt(M[0]);
    L[0] = M[0] * r0;
    L[1] = M[1] * r0;
    L[3] = M[3] * r0;
    const float d1 = M[2] - L[1] * L[1], r1 = ls_rsqrt(d1);
    L[2] = d1 * r1;
    L[4] = (M[4] - L[3] * L[1]) * r1;
    const float d2 = M[5] - L[3] * L[3] - L[4] * L[4], r2 = ls_rsqrt(d2);
    L[5] = d2 * r2;
    L[6] = r0; L[7] = r1; L[8] = r2;
}
LS_FN void chol3_solve(const float* L, float* b) {
    b[0] = b[0] * L[6];
    b[1] = (b[1] - L[1] * b[0]) * L[7];
    b[2] = (b[2] - L[3] * b[0] - L[4] * b[1]) * L[8];
    b[2] = b[2] * L[8];
    b[1] = (b[1] - L[4] * b[2]) * L[7];
    b[0] = (b[0] - L[1] * b[1] - L[3] * b[2]) * L[6];
}
// 6x6 Cholesky solve; lower factor row-major in a 36-float array with the RECIPROCAL of the diagonal stored on the diagonal
LS_FN void chol6_solve(const float* L, float* b) {
    for (int i = 0; i < 6; ++i) {
        float v = b[i];
        for (int k = 0; k < i; ++k) v -= L[6 * i + k] * b[k];
        b[i] = v * L[6 * i + i];
    }
    for (int i = 5; i >= 0; --i) {
        float v = b[i];
        for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * b[k];
        b[i] = v * L[6 * i + i];
    }
}

// ---- phase L1a: composite inertias / bias forces up each leg, one matrix ROW per lane (lane = 6 * leg + row):
//      F_k = Ic_k S_k (columns of Mbl) and the accumulated bias force at every level (parked in sh.G, which the Schur
//      phase only writes afterwards).  M_jk = S_j . F_k for ancestors j <= k and h_k = S_k . f_k are 6-lane reductions:
//      they are finished by ph_leg_block from LDS.
LS_FN void ph_leg_composite(WaveShared& sh, int lane) {
    if (lane >= 24) return;
    const int l = lane / 6, r = lane - 6 * l;
    float Ic[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    float fb = 0.0f;
    for (int k = 3; k >= 0; --k) {
        const int b = 1 + 4 * l + k;
        const float* I6 = sh.u.I6[b] + 6 * r;
        for (int c = 0; c < 6; ++c) Ic[c] += I6[c];
        fb += sh.Fb[b][r];
        if (k < 3) {
            const float* S = sh.S[3 * l + k];
            float f = 0.0f;
            for (int c = 0; c < 6; ++c) f += Ic[c] * S[c];
            sh.Mbl[l][3 * r + k] = f;
            sh.G[l][6 * k + r] = fb;
        }
    }
    sh.legF[l][r] = fb;
}
// ---- phase L1b: the 3x3 leg block M_ll (6 entries, parked in sh.lam until the Schur phase factors it) and h_l
//      (lane = 9 * leg + entry; entries 0-5 = m00,m10,m11,m20,m21,m22, entries 6-8 = h_hip,h_thigh,h_calf)
LS_FN void ph_leg_block(WaveShared& sh, int lane) {
    if (lane >= 36) return;
    const int l = lane / 9, e = lane - 9 * l;
    if (e < 6) {
        const int j = (e == 2 || e == 4) ? 1 : (e == 5 ? 2 : 0);           // ancestor (row of S)
        const int k = (e == 0) ? 0 : (e <= 2 ? 1 : 2);                      // column F_k
        const float* S = sh.S[3 * l + j];
        float m = 0.0f;
        for (int r = 0; r < 6; ++r) m += S[r] * sh.Mbl[l][3 * r + k];
        sh.lam[6 * l + e] = m;
    } else {
        const int k = e - 6;
        const float* S = sh.S[3 * l + k];
        float h = 0.0f;
        for (int r = 0; r < 6; ++r) h += S[r] * sh.G[l][6 * k + r];
        sh.hl[l][k] = h;
    }
}
// ---- phase L2: Cholesky of M_ll (every lane of the leg redundantly, lane c == 0 publishes it) and
//      G_l = Mll^-1 Mlb^T (3x6) from the stored columns (lane = 6 * leg + column)
LS_FN void ph_leg_schur(WaveShared& sh, int lane) {
    if (lane >= 24) return;
    int l = lane / 6, c = lane - 6 * l;
    float L[LS_CHOL3];
    chol3(sh.lam + 6 * l, L);
    if (c == 0) for (int e = 0; e < LS_CHOL3; ++e) sh.Lll[l][e] = L[e];
    float col[3] = {sh.Mbl[l][3 * c], sh.Mbl[l][3 * c + 1], sh.Mbl[l][3 * c + 2]};   // (F_h[c], F_t[c], F_c[c])
    chol3_solve(L, col);
    sh.G[l][c] = col[0]; sh.G[l][6 + c] = col[1]; sh.G[l][12 + c] = col[2];
}

// ---- phase E: base Schur complement Sb = sum_b I6[b] - sum_l Mbl_l G_l and total base bias force (lane = element)
LS_FN void ph_base_assemble(WaveShared& sh, int lane) {
    if (lane < 36) {
        float s = 0.0f;
        for (int b = 0; b < LS_NB; ++b) s += sh.u.I6[b][lane];
        int r = lane / 6, c = lane - 6 * r;
        for (int l = 0; l < 4; ++l)
            s -= sh.Mbl[l][3 * r] * sh.G[l][c] + sh.Mbl[l][3 * r + 1] * sh.G[l][6 + c] + sh.Mbl[l][3 * r + 2] * sh.G[l][12 + c];
        sh.Sb[lane] = s;
    } else if (lane < 42) {
        int k = lane - 36;
        float s = sh.Fb[0][k];
        for (int l = 0; l < 4; ++l) s += sh.legF[l][k];
        sh.hb[k] = s;
    }
}

// ---- phase: Cholesky of the 6x6 Schur complement and its explicit inverse (lane = column of the inverse)
//      Lanes 0-5 all run the factorisation (free in SIMT) and lane c then solves for column c of the inverse, so the two
//      later users (free base acceleration, one solve per constraint row) do 36 FMAs instead of a 12-division substitution.
LS_FN void ph_base_factor(WaveShared& sh, int lane) {
    if (lane >= 6) return;
    float A[36];
    for (int e = 0; e < 36; ++e) A[e] = sh.Sb[e];
    for (int j = 0; j < 6; ++j) {
        float d = A[6 * j + j];
        for (int k = 0; k < j; ++k) d -= A[6 * j + k] * A[6 * j + k];
        const float rd = ls_rsqrt(fmaxf(d, 1e-12f));
        A[6 * j + j] = rd;                      // reciprocal of the diagonal entry (chol6_solve multiplies by it)
        for (int i = j + 1; i < 6; ++i) {
            float v = A[6 * i + j];
            for (int k = 0; k < j; ++k) v -= A[6 * i + k] * A[6 * j + k];
            A[6 * i + j] = v * rd;
        }
    }
    float col[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int r = 0; r < 6; ++r) col[r] = (r == lane) ? 1.0f : 0.0f;
    chol6_solve(A, col);
    for (int r = 0; r < 6; ++r) sh.Sinv[6 * r + lane] = col[r];
}

// ---- free velocity: three small phases
//      1. lanes 0-3 (leg): y_l = Mll^-1 (tau_l - h_l);  lanes 8-13 (base row r): s_r = -h_b[r] - sum_l (G_l^T (tau_l - h_l))[r]
//         (Mbl Mll^-1 = G^T, so the base right-hand side needs no leg solve and both run in the same phase)
//      2. lanes 0-5: a_b = Sb^-1 s      3. ph_free_finish
LS_FN void ph_free_leg(WaveShared& sh, int lane) {
    if (lane < 4) {
        float y[3];
        for (int k = 0; k < 3; ++k) y[k] = sh.tau[3 * lane + k] - sh.hl[lane][k];
        chol3_solve(sh.Lll[lane], y);
        for (int k = 0; k < 3; ++k) sh.yl[lane][k] = y[k];
    } else if (lane >= 8 && lane < 14) {
        const int r = lane - 8;
        float s = -sh.hb[r];
        for (int l = 0; l < 4; ++l)
            for (int k = 0; k < 3; ++k) s -= sh.G[l][6 * k + r] * (sh.tau[3 * l + k] - sh.hl[l][k]);
        sh.rb[r] = s;
    }
}
LS_FN void ph_free_base(WaveShared& sh, int lane) {  // lane = row of a_b = Sb^-1 s
    if (lane >= 6) return;
    float a = 0.0f;
    for (int r = 0; r < 6; ++r) a += sh.Sinv[6 * lane + r] * sh.rb[r];
    sh.ab[lane] = a;
}
LS_FN void ph_free_finish(WaveShared& sh, int lane, float dt) {  // lane = generalized velocity index
    if (lane >= LS_NV) return;
    float v, a;
    if (lane < 6) {
        v = (lane < 3) ? sh.root[10 + lane] : sh.root[7 + lane - 3];
        a = sh.ab[lane];
    } else {
        int d = lane - 6, l = d / 3, k = d % 3;
        v = sh.qd[d];
        a = sh.yl[l][k];
        for (int c = 0; c < 6; ++c) a -= sh.G[l][6 * k + c] * sh.ab[c];
    }
    sh.vfree[lane] = v + dt * a;
}

// ---- terrain surface under a world point: triangulated height grid, diagonal (i,j)-(i+1,j+1)
LS_FN void ls_terrain_query(const LsCtx& cx, float x, float y, float& h, V3& n) {
    const lsim_config& c = cx.cfg;
    if (c.mesh_type == 0) { h = 0.0f; n = v3(0, 0, 1); return; }
    LS_GLOBAL const int16_t* g = LS_G(const int16_t, cx.buf[LSIM_BUF_HEIGHT_GRID]);
    float hs = c.horizontal_scale, vs = c.vertical_scale;
    const float ihs = ls_rcp(hs);
    float gx = (x + c.border_size) * ihs, gy = (y + c.border_size) * ihs;
    float fi = clampf(floorf(gx), 0.0f, (float)(c.grid_rows - 2)), fj = clampf(floorf(gy), 0.0f, (float)(c.grid_cols - 2));
    int i = ls_f2i(fi), j = ls_f2i(fj);
    float u = clampf(gx - fi, 0.0f, 1.0f), v = clampf(gy - fj, 0.0f, 1.0f);
    float h00 = g[i * c.grid_cols + j] * vs, h10 = g[(i + 1) * c.grid_cols + j] * vs;
    float h01 = g[i * c.grid_cols + j + 1] * vs, h11 = g[(i + 1) * c.grid_cols + j + 1] * vs;
    float dhx, dhy;
    if (u >= v) { dhx = h10 - h00; dhy = h11 - h10; h = h00 + u * dhx + v * dhy; }
    else { dhx = h11 - h01; dhy = h01 - h00; h = h00 + v * dhy + u * dhx; }
    float nx = -dhx * ihs, ny = -dhy * ihs, inv = ls_rsqrt(nx * nx + ny * ny + 1.0f);
    n = v3(nx * inv, ny * inv, inv);
}

LS_FN void ls_tangent_basis(V3 n, V3& t1, V3& t2) {
    V3 ref = (fabsf(n.x) > 0.9f) ? v3(0, 1, 0) : v3(1, 0, 0);
    t1 = cross(n, ref);
    t1 = t1 * ls_rsqrt(dot(t1, t1));
    t2 = cross(n, t1);
}

// closest point of triangle (a,b,c) to p (Ericson, Real-Time Collision Detection 5.1.5)
LS_FN V3 ls_closest_on_triangle(V3 p, V3 a, V3 b, V3 c) {
    V3 ab = b - a, ac = c - a, ap = p - a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.0f && d2 <= 0.0f) return a;
    V3 bp = p - b;
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.0f && d4 <= d3) return b;
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) return a + ab * (d1 * ls_rcp(fmaxf(d1 - d3, 1e-20f)));
    V3 cp = p - c;
    float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.0f && d5 <= d6) return c;
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) return a + ac * (d2 * ls_rcp(fmaxf(d2 - d6, 1e-20f)));
    float va = d3 * d6 - d5 * d4;
    if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) return b + (c - b) * ((d4 - d3) * ls_rcp(fmaxf((d4 - d3) + (d5 - d6), 1e-20f)));
    float denom = ls_rcp(fmaxf(va + vb + vc, 1e-20f));
    return a + ab * (vb * denom) + ac * (vc * denom);
}

// world position of a vertex of the reference's triangle mesh from its packed word (LSIM_BUF_TERRAIN_MESH)
LS_FN V3 ls_mesh_vertex(const lsim_config& c, int word, int a, int b) {
    float dx = (float)(((word >> 16) & 3) - 1), dy = (float)(((word >> 18) & 3) - 1);
    return v3(((float)a + dx) * c.horizontal_scale - c.border_size, ((float)b + dy) * c.horizontal_scale - c.border_size,
              (float)(int16_t)(word & 0xFFFF) * c.vertical_scale);
}

// signed distance (negative inside the ground) and contact normal of a sphere centre against the terrain.
//   fast path: plane of the grid triangle under the point (exact where no vertex of the 4x4 block is displaced);
//   wall path: closest point over the triangles of the 3x3 cells around the point in the displaced mesh (TER:72-75).  Only
//   features within reach = radius + contact_offset can make a contact, so cells whose bounding box (grown by reach in x/y,
//   and in +z) excludes the centre are skipped, as are triangles whose plane lies more than reach below the centre.
// In pieces, because the GPU spreads the wall path's 18 (cell, triangle) candidates of a point over the lanes of the wave (wc_wall_contacts
// below) while the lane emulator and the single-point form walk them in order: the same arithmetic per candidate, the same winner.
//
// ls_terrain_fast: everything up to the decision.  true: (dist, n) are final; false: the point needs the wall path around cell (gi, gj).
LS_FN bool ls_terrain_fast(const LsCtx& cx, V3 cw, float radius, float& dist, V3& n, int& gi, int& gj) {
    const lsim_config& c = cx.cfg;
    gi = 0; gj = 0;
    if (c.mesh_type == 0) { dist = cw.z; n = v3(0, 0, 1); return true; }
    LS_GLOBAL const int* mesh = LS_G(const int, cx.buf[LSIM_BUF_TERRAIN_MESH]);
    const float hs = c.horizontal_scale, vs = c.vertical_scale;
    const float ihs = ls_rcp(hs);
    float gx = (cw.x + c.border_size) * ihs, gy = (cw.y + c.border_size) * ihs;
    float fi = clampf(floorf(gx), 0.0f, (float)(c.grid_rows - 2)), fj = clampf(floorf(gy), 0.0f, (float)(c.grid_cols - 2));
    int i = ls_f2i(fi), j = ls_f2i(fj);
    gi = i; gj = j;
    const int w00 = mesh[i * c.grid_cols + j];
    {   // nothing within the 4 x 4 vertex block reaches up to the sphere (ls_api_impl.h: bits 24-31 = the block's highest vertex above this
        // one, rounded up): no contact, whatever the faces look like -- decided after this one load
        const int dz = (int)((unsigned int)w00 >> 24);
        const float top = ((float)(int16_t)(w00 & 0xFFFF) + (float)(dz * LSIM_MESH_DZ_UNIT)) * vs;
        if (dz < 255 && cw.z - top > radius + c.contact_offset) { dist = cw.z - top; n = v3(0, 0, 1); return true; }
    }
#if defined(LS_NO_WALLS)   // A/B experiments only
    const bool walls = false;
#else
    const bool walls = (w00 & (1 << 20)) != 0;
#endif
    if (walls) return false;
    float h00 = (float)(int16_t)(w00 & 0xFFFF) * vs, h10 = (float)(int16_t)(mesh[(i + 1) * c.grid_cols + j] & 0xFFFF) * vs;
    float h01 = (float)(int16_t)(mesh[i * c.grid_cols + j + 1] & 0xFFFF) * vs;
    float h11 = (float)(int16_t)(mesh[(i + 1) * c.grid_cols + j + 1] & 0xFFFF) * vs;
    float u = clampf(gx - fi, 0.0f, 1.0f), v = clampf(gy - fj, 0.0f, 1.0f);
    float dhx, dhy, h;
    if (u >= v) { dhx = h10 - h00; dhy = h11 - h10; h = h00 + u * dhx + v * dhy; }
    else { dhx = h11 - h01; dhy = h01 - h00; h = h00 + v * dhy + u * dhx; }
    float nx = -dhx * ihs, ny = -dhy * ihs, inv = ls_rsqrt(nx * nx + ny * ny + 1.0f);
    n = v3(nx * inv, ny * inv, inv);
    dist = (cw.z - h) * inv;
    return true;
}
// one candidate cell of the wall path: cell number `cell` (0..8, row-major) of the 3 x 3 block around (gi, gj), its two triangles
// (ind0, ind3, ind1) then (ind0, ind2, ind3).  Returns the smaller squared distance of cw to them (the first on a tie) -- 1e30 when the cell
// lies outside the grid or out of reach, or both triangles are collapsed or have their plane out of reach below the centre -- with the
// closest point q and the unit face normal fn.
LS_FN float ls_wall_cell(const LsCtx& cx, V3 cw, float reach, int gi, int gj, int cell, V3& q, V3& fn) {
    const lsim_config& c = cx.cfg;
    LS_GLOBAL const int* mesh = LS_G(const int, cx.buf[LSIM_BUF_TERRAIN_MESH]);
    q = v3(0, 0, 0); fn = v3(0, 0, 1);
    float best = 1e30f;
    const int ci = gi - 1 + cell / 3, cj = gj - 1 + cell % 3;
    if (ci < 0 || cj < 0 || ci > c.grid_rows - 2 || cj > c.grid_cols - 2) return best;
    LS_GLOBAL const int* row = mesh + ci * c.grid_cols + cj;
    V3 p00 = ls_mesh_vertex(c, row[0], ci, cj), p10 = ls_mesh_vertex(c, row[c.grid_cols], ci + 1, cj);
    V3 p01 = ls_mesh_vertex(c, row[1], ci, cj + 1), p11 = ls_mesh_vertex(c, row[c.grid_cols + 1], ci + 1, cj + 1);
    float xlo = fminf(fminf(p00.x, p10.x), fminf(p01.x, p11.x)), xhi = fmaxf(fmaxf(p00.x, p10.x), fmaxf(p01.x, p11.x));
    float ylo = fminf(fminf(p00.y, p10.y), fminf(p01.y, p11.y)), yhi = fmaxf(fmaxf(p00.y, p10.y), fmaxf(p01.y, p11.y));
    float zhi = fmaxf(fmaxf(p00.z, p10.z), fmaxf(p01.z, p11.z));
    if (cw.x < xlo - reach || cw.x > xhi + reach || cw.y < ylo - reach || cw.y > yhi + reach || cw.z > zhi + reach) return best;
    // rolled on purpose: two inlined copies of the triangle query per cell cost code and registers
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
        V3 a = p00, b = t == 0 ? p11 : p10, cc = t == 0 ? p01 : p11;
        V3 nt = cross(b - a, cc - a);
        float nl2 = dot(nt, nt);
        if (nl2 < 1e-16f) continue;                                   // collapsed triangle
        float nl = sqrtf(nl2);
        if (dot(nt, cw - a) > reach * nl) continue;                   // plane out of reach below the centre
        V3 qt = ls_closest_on_triangle(cw, a, b, cc);
        V3 dq = cw - qt;
        float d2 = dot(dq, dq);
        if (d2 < best) { best = d2; q = qt; fn = nt * ls_rcp(nl); }
    }
    return best;
}
// from the winning candidate (best = its squared distance; > 1e29: none) to (dist, n)
LS_FN void ls_wall_finish(V3 cw, float radius, float best, V3 bq, V3 bn, float& dist, V3& n) {
    if (best > 1e29f) { dist = 1.0f + radius; n = v3(0, 0, 1); return; }
    float d = sqrtf(best);
    V3 dq = cw - bq;
    float side = dot(bn, dq);
    if (side < -1e-6f) { dist = -d; n = bn; }                  // centre behind the face: inside the ground
    else if (d > 1e-6f) { dist = d; n = dq * ls_rcp(d); }
    else { dist = 0.0f; n = bn; }
}
// the wall path of one point, cells in order (the first strict minimum wins)
LS_FN void ls_wall_serial(const LsCtx& cx, V3 cw, float radius, int gi, int gj, float& dist, V3& n) {
    const float reach = radius + cx.cfg.contact_offset;
    float best = 1e30f;
    V3 bq = v3(0, 0, 0), bn = v3(0, 0, 1);
    // rolled on purpose: nine inlined copies of the cell query cost 24 KB of code and 160 VGPRs
#pragma unroll 1
    for (int cell = 0; cell < 9; ++cell) {
        V3 q, fn;
        const float d2 = ls_wall_cell(cx, cw, reach, gi, gj, cell, q, fn);
        if (d2 < best) { best = d2; bq = q; bn = fn; }
    }
    ls_wall_finish(cw, radius, best, bq, bn, dist, n);
}
LS_FN void ls_terrain_contact(const LsCtx& cx, V3 cw, float radius, float& dist, V3& n) {
    int gi, gj;
    if (!ls_terrain_fast(cx, cw, radius, dist, n, gi, gj)) ls_wall_serial(cx, cw, radius, gi, gj, dist, n);
}

// ---- the collision point's constants (lane = point): re-read from the cache-resident model every sub-step rather than held in registers through
//      the solver (the register peak), but at the TOP of the sub-step, so that the load latency hides behind the dynamics phases
LS_FN void ph_collide_prefetch(const LsCtx& cx, LaneRegs& r, int lane) {
    const lsim_collision_point& cp = cx.model.points[lane < cx.model.num_collision_points ? lane : 0];
    r.cp_body = cp.body;
    r.cp_r = cp.radius;
    for (int k = 0; k < 3; ++k) r.cp_pos[k] = cp.pos[k];
}
// ---- phase P: narrow phase, one collision point per lane
#if !defined(LS_EMU) && defined(__HIP_DEVICE_COMPILE__)
// Wall path of the points in `mask` (lane = point; `mine`: this lane is one of them), all lanes of the wave at work: a point's nine cells
// walked by its own lane while the lanes without a wall wait was the slowest thing a wave could meet (stairs: the 1-2 % of the waves with
// feet at a riser ended 40 us after the median wave; 0.131 ms per launch against 0.110 on the flat task).  Seven points per round: their
// owners post (centre, reach, cell), lane 9 o + k evaluates cell k of owner o, the owners pick the first strict minimum of their nine
// results -- the order and the arithmetic of ls_wall_serial.  Scratch: the spatial-inertia array, dead between the composite pass and the
// row build.  With most of the wave in the wall path (a robot lying on a staircase) the rounds stop paying: every lane walks its own cells.
#define LS_WALL_OWNERS 7
__device__ __forceinline__ void wc_wall_contacts(const LsCtx& cx, WaveShared& sh, int lane, bool mine, unsigned long long mask, V3 cw, float radius, int gi, int gj,
                                                 float& dist, V3& n) {
    float* own = &sh.u.I6[0][0];                  // [7][8]: cw.x, cw.y, cw.z, reach, gi, gj
    float* res = own + 8 * LS_WALL_OWNERS;        // [63][8]: d2, q.xyz, fn.xyz
    static_assert(sizeof(sh.u.I6) >= (8 * LS_WALL_OWNERS + 9 * LS_WALL_OWNERS * 8) * sizeof(float), "scratch of the wall path");
    const int m = __popcll(mask);
    if (m > 5 * LS_WALL_OWNERS) {                 // wave-uniform
        if (mine) ls_wall_serial(cx, cw, radius, gi, gj, dist, n);
        return;
    }
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    const float reach = radius + cx.cfg.contact_offset;
    for (int c0 = 0; c0 < m; c0 += LS_WALL_OWNERS) {
        const bool posting = mine && rank >= c0 && rank < c0 + LS_WALL_OWNERS;
        if (posting) {
            float* o = own + 8 * (rank - c0);
            o[0] = cw.x; o[1] = cw.y; o[2] = cw.z; o[3] = reach; o[4] = __int_as_float(gi); o[5] = __int_as_float(gj);
        }
        LS_WAVE_SYNC();
        {
            const int o = lane / 9, k = lane - 9 * o;
            if (lane < 9 * LS_WALL_OWNERS && c0 + o < m) {
                const float* od = own + 8 * o;
                V3 q, fn;
                const float d2 = ls_wall_cell(cx, v3(od[0], od[1], od[2]), od[3], __float_as_int(od[4]), __float_as_int(od[5]), k, q, fn);
                float* r = res + 8 * lane;
                r[0] = d2; r[1] = q.x; r[2] = q.y; r[3] = q.z; r[4] = fn.x; r[5] = fn.y; r[6] = fn.z;
            }
        }
        LS_WAVE_SYNC();
        if (posting) {
            const float* r = res + 8 * 9 * (rank - c0);
            float best = 1e30f;
            int win = 0;
            for (int k = 0; k < 9; ++k) { const float d2 = r[8 * k]; if (d2 < best) { best = d2; win = k; } }
            const float* w = r + 8 * win;
            ls_wall_finish(cw, radius, best, v3(w[1], w[2], w[3]), v3(w[4], w[5], w[6]), dist, n);
        }
        LS_WAVE_SYNC();
    }
}
#endif
// returns whether any point of the wave took the wall path (GPU: wave-uniform; lane emulator: this lane's point)
LS_FN bool ph_collide(const LsCtx& cx, WaveShared& sh, LaneRegs& r, int lane) {
    r.cp_active = 0;
    const bool has = lane < cx.model.num_collision_points;
    const int b = r.cp_body;
    const float cp_r = r.cp_r;
    V3 pw = mul(m3p(sh.R[b]), v3(r.cp_pos[0], r.cp_pos[1], r.cp_pos[2])) + v3p(sh.p[b]);
    const V3 cw = v3(sh.root[0] + pw.x, sh.root[1] + pw.y, sh.root[2] + pw.z);
    float d = 1.0f;
    V3 n = v3(0, 0, 1);
    int gi = 0, gj = 0;
    const bool walls = has && !ls_terrain_fast(cx, cw, cp_r, d, n, gi, gj);
#if !defined(LS_EMU) && defined(__HIP_DEVICE_COMPILE__) && !defined(LS_SERIAL_WALLS)
    const unsigned long long mask = __ballot(walls);
    if (mask != 0ull) wc_wall_contacts(cx, sh, lane, walls, mask, cw, cp_r, gi, gj, d, n);      // wave-uniform branch
    const bool any_walls = mask != 0ull;
#else
    if (walls) ls_wall_serial(cx, cw, cp_r, gi, gj, d, n);
    const bool any_walls = walls;
#endif
    if (!has) return any_walls;
    float dist = d - cp_r;
    if (dist < cx.cfg.contact_offset) {
        r.cp_active = 1;
        r.cp_dist = dist;
        v3st(r.cp_n, n);
        v3st(r.cp_x, pw - n * cp_r);
    }
    return any_walls;
}

// ---- wave collective: ordered compaction of the active points into at most LS_MAXC contacts
#if defined(LS_EMU)
static inline void wc_compact_contacts(WaveShared& sh, LaneRegs* L) {
    int nc = 0, nact = 0;
    for (int lane = 0; lane < 64; ++lane)
        if (L[lane].cp_active) {
            ++nact;
            if (nc >= LS_MAXC) continue;
            sh.cbody[nc] = L[lane].cp_body; sh.cdist[nc] = L[lane].cp_dist;
            for (int k = 0; k < 3; ++k) { sh.cn[nc][k] = L[lane].cp_n[k]; sh.cpos[nc][k] = L[lane].cp_x[k]; }
            ++nc;
        }
    sh.nc = nc;
    sh.nact = nact;
    sh.nact_max = nact > sh.nact_max ? nact : sh.nact_max;
}
#else
LS_FN void wc_compact_contacts(WaveShared& sh, LaneRegs& r, int lane) {
    unsigned long long m = __ballot(r.cp_active != 0);
    int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (r.cp_active && rank < LS_MAXC) {
        sh.cbody[rank] = r.cp_body; sh.cdist[rank] = r.cp_dist;
        for (int k = 0; k < 3; ++k) { sh.cn[rank][k] = r.cp_n[k]; sh.cpos[rank][k] = r.cp_x[k]; }
    }
    if (lane == 0) {
        int n = __popcll(m);
        sh.nc = n < LS_MAXC ? n : LS_MAXC;
        sh.nact = n;
        sh.nact_max = n > sh.nact_max ? n : sh.nact_max;
    }
}
#endif

// ---- phase: joint position AND velocity limits as ONE two-sided ("boxed") row per joint: L <= qd_j <= U with
//      [L, U] = [-vmax, vmax]  cut by  qd >= -gap_lo / dt (within 0.1 rad of the lower stop)  /  qd <= gap_hi / dt (upper stop); a
//      penetrated stop pushes back with erp, capped at 1 rad/s.  A joint gets its row when its free velocity violates a bound or comes within
//      LS_LIMIT_MARGIN of the velocity limit of it; because the limit impulses of one joint move its neighbours on the same leg by tens of
//      rad/s, a joint that VIOLATES a bound -- whose row will carry an impulse -- also gives its two neighbours on the leg their rows.
//      [Until round 2 a joint merely within the margin did that too: 4.6 of 5.3 limit rows per solve never carried an impulse.  Rows for the
//      needing joints alone fail tests/test_physics_invariants.py's saturated-motor case: the kicked neighbours run into the 1.5 x safety clamp,
//      which is not momentum-neutral.]  [Clamping the joint velocity after the solve instead leaves the reaction of a saturated motor torque
//      on the base and spins up a robot in free flight.]
//      The rows are laid out in DESCENDING joint order (calf, thigh, hip of the last leg first: leaf to root within a leg).  The sweep relaxes
//      slots in ascending order, and a Gauss-Seidel pass that goes leaf to root leaves about a third of the residue of the opposite order on a
//      saturated leg (round 4, oracle: with TGS's single pass per iteration the joint speed exceeded 1.1 x its limit in 2.7 % instead of 19 % of
//      the steps of the saturated-motor invariant).
//      GPU: lane = joint, ordered compaction by ballot; lane emulator: lane 0 walks the joints (same order, same result)
#define LS_LIMIT_MARGIN 0.2f    // a joint "needs" its row when the free velocity is within this fraction of vmax of a bound (or beyond it)
LS_FN bool ls_joint_limit_bounds(const LsCtx& cx, const WaveShared& sh, int j, float dt, float& Lb, float& Ub, bool& violates) {
    const float idt = ls_rcp(dt);
    const float lo = sh.q[j] - sh.jc_lo[j], hi = sh.jc_hi[j] - sh.q[j];
    const float vmax = sh.jc_vmax[j], vf = sh.vfree[6 + j];
    Lb = -vmax; Ub = vmax;
    if (lo < 0.1f) Lb = fmaxf(Lb, lo >= 0.0f ? -lo * idt : fminf(1.0f, cx.cfg.erp * (-lo) * idt));
    if (hi < 0.1f) Ub = fminf(Ub, hi >= 0.0f ? hi * idt : -fminf(1.0f, cx.cfg.erp * (-hi) * idt));
    if (Ub < Lb) Ub = Lb;                                   // both stops violated at once cannot happen; keep the box well formed
    violates = fminf(vf - Lb, Ub - vf) < 0.0f;
    return fminf(vf - Lb, Ub - vf) < LS_LIMIT_MARGIN * vmax;
}
#if !defined(LS_EMU)
LS_FN void wc_limits(const LsCtx& cx, WaveShared& sh, int lane, float dt) {
    bool need = false, viol = false;
    float Lb = 0.0f, Ub = 0.0f;
    if (lane < 12) need = ls_joint_limit_bounds(cx, sh, lane, dt, Lb, Ub, viol);
    const unsigned long long mviol = __ballot(viol);
    const bool has = need || (lane < 12 && ((mviol >> (3 * (lane / 3))) & 7ull) != 0ull);     // own need, or a violating joint on this leg
    const unsigned long long m = __ballot(has);
    if (has) {      // rows in DESCENDING joint order (leaf to root within a leg: see the comment above): rank = rows of the joints above this one
        const int rank = __popcll(m >> (lane + 1));
        sh.limdof[rank] = lane; sh.limvt[rank] = Lb; sh.limrng[rank] = Ub - Lb;
    }
    if (lane == 0) { const int n = __popcll(m); sh.nlim = n; sh.nrows = 3 * sh.nc + n; }
}
#endif
LS_FN void ph_limits(const LsCtx& cx, WaveShared& sh, int lane, float dt) {
    if (lane != 0) return;
    int n = 0;
    for (int leg = 3; leg >= 0; --leg) {
        float Lb[3], Ub[3];
        bool need[3], viol[3], any_viol = false;
        for (int k = 0; k < 3; ++k) { need[k] = ls_joint_limit_bounds(cx, sh, 3 * leg + k, dt, Lb[k], Ub[k], viol[k]); any_viol = any_viol || viol[k]; }
        for (int k = 2; k >= 0; --k) {
            if (!(need[k] || any_viol)) continue;
            sh.limdof[n] = 3 * leg + k; sh.limvt[n] = Lb[k]; sh.limrng[n] = Ub[k] - Lb[k]; ++n;
        }
    }
    sh.nlim = n;
    sh.nrows = 3 * sh.nc + n;
}

// ---- phase R1: constraint row Jacobian, Y = M^-1 J^T (structured solve), right-hand side (lane = row)
template <bool TGS> LS_FN void ph_rows(const LsCtx& cx, WaveShared& sh, LaneRegs& r, int lane, float dt) {
    r.row_kind = -1;
    r.row_rng = __builtin_inff();
    if (!ls_slot_active(sh, lane)) return;
    const lsim_config& c = cx.cfg;
    float vt, rng = __builtin_inff();     // rng: width of a two-sided row's velocity interval (joint limits), +inf for one-sided rows
    float tg_a = 0.0f, tg_b = 0.0f;       // TGS: what the row's target is re-derived from in every sub-iteration
    int leg;
    float Jb[6] = {0, 0, 0, 0, 0, 0}, Jl[3] = {0, 0, 0};
    V3 d = v3(0, 0, 0);
    if (lane < 3 * sh.nc) {
        int k = lane / 3, a = lane - 3 * k;
        V3 n = v3p(sh.cn[k]), t1, t2;
        ls_tangent_basis(n, t1, t2);
        d = (a == 0) ? n : (a == 1 ? t1 : t2);
        V3 x = v3p(sh.cpos[k]);
        S6 f = s6(cross(x, d), d);
        s6st(Jb, f);
        int b = sh.cbody[k];
        leg = (b == 0) ? -1 : ls_leg_of_body(b);
        if (b > 0) {
            int depth = ls_depth_of_body(b);
            int nd = depth > 2 ? 3 : depth + 1;
            for (int j = 0; j < nd; ++j) Jl[j] = dot(s6p(sh.S[3 * leg + j]), f);
        }
        r.row_kind = a;
        if (a == 0) {
            float dist = sh.cdist[k];
            tg_a = dist;
            const float idt = ls_rcp(dt);
            if (dist >= 0.0f) vt = -dist * idt;
            else vt = fminf(c.max_depenetration_velocity, c.erp * fmaxf(-dist - c.contact_slop, 0.0f) * idt);
        } else vt = 0.0f;
    } else {
        int i = lane - LS_LIM0;
        int j = sh.limdof[i];
        leg = j / 3;
        Jl[j - 3 * leg] = 1.0f;
        vt = sh.limvt[i];                 // lower velocity bound; the row is two-sided: the upper bound is limrng above it
        rng = sh.limrng[i];
        if (TGS) {                        // the bounds move with the joint angle: hand over the distances to the stops and the velocity limit
            rng = sh.jc_vmax[j];
            tg_a = sh.q[j] - sh.jc_lo[j];
            tg_b = sh.jc_hi[j] - sh.q[j];
        }
        r.row_kind = 3;
    }
    r.row_leg = leg;
    for (int k = 0; k < 3; ++k) r.Jl[k] = Jl[k];
    if (lane < 3 * LS_MAXC) v3st(sh.u.c.dirs[lane], d);
    // M^-1 J^T without the base-coupling leg terms (ls_shared.h, Y): y = Mll^-1 Jl, a = Jb - Mbl y (kept in r.Jb), z = Sb^-1 a
    float y[3] = {0, 0, 0}, av[6];
    for (int k = 0; k < 6; ++k) av[k] = Jb[k];
    if (leg >= 0) {
        for (int k = 0; k < 3; ++k) y[k] = Jl[k];
        chol3_solve(sh.Lll[leg], y);
        for (int k = 0; k < 6; ++k) av[k] -= sh.Mbl[leg][3 * k] * y[0] + sh.Mbl[leg][3 * k + 1] * y[1] + sh.Mbl[leg][3 * k + 2] * y[2];
    }
    for (int k = 0; k < 6; ++k) r.Jb[k] = av[k];
    float* Y = sh.u.c.Y[lane];
    for (int k = 0; k < 6; ++k) {
        float s = 0.0f;
        for (int cc = 0; cc < 6; ++cc) s += sh.Sinv[6 * k + cc] * av[cc];
        Y[k] = s;
    }
    for (int l = 0; l < 4; ++l)
        for (int k = 0; k < 3; ++k) Y[6 + 3 * l + k] = (l == leg) ? y[k] : 0.0f;
    float jv = 0.0f;
    for (int k = 0; k < 6; ++k) jv += Jb[k] * sh.vfree[k];
    if (leg >= 0) for (int k = 0; k < 3; ++k) jv += Jl[k] * sh.vfree[6 + 3 * leg + k];
    if (TGS) {                            // the iterations start from the free velocity; the targets move with them
        r.tg_a = tg_a; r.tg_b = tg_b;
        r.brow = jv;
    } else r.brow = jv - vt;
    r.row_rng = rng;
}

// ---- phase R2 (lane emulator only; the GPU fuses it into wc_delassus_pgs): Delassus row W_i. = J_i Y^T (lane = row i)
#if defined(LS_EMU)
LS_FN void ph_delassus(WaveShared& sh, LaneRegs& r, int lane) {
    if (!ls_slot_active(sh, lane)) return;
    const int leg = r.row_leg;
    const int lo = leg >= 0 ? 6 + 3 * leg : 6;
    const float jl0 = leg >= 0 ? r.Jl[0] : 0.0f, jl1 = leg >= 0 ? r.Jl[1] : 0.0f, jl2 = leg >= 0 ? r.Jl[2] : 0.0f;
    float wd = 0.0f;
    for (int j = 0; j < LS_MAXR; ++j) {
        if (ls_slot_active(sh, j)) {
            const float* Y = sh.u.c.Y[j];
            float w = 0.0f;
            for (int k = 0; k < 6; ++k) w += r.Jb[k] * Y[k];
            w += jl0 * Y[lo] + jl1 * Y[lo + 1] + jl2 * Y[lo + 2];
            if (j == lane) { w += 1e-6f; wd = w; }   // constraint-force mixing keeps the diagonal positive
            r.W[j] = w;
        }
    }
    r.wdiag = wd;
}
#endif

// TGS (wc_delassus_tgs / wc_tgs below): where the solver parks what the integrator needs
#define LS_VEL0 LS_MAXR
LS_FN float* ls_tgs_base_twist(WaveShared& sh, int s) { return sh.Sb + 6 * s; }     // [LSIM_MAX_POSITION_ITERATIONS][6] over Sb + Sinv
LS_FN float* ls_tgs_vel_sum(WaveShared& sh) { return &sh.legF[0][0]; }             // [18]: sum over the sub-iterations (legF is dead since the base assembly)
static_assert(LS_VEL0 + LS_NV <= 64, "velocity lanes");
static_assert(6 * LSIM_MAX_POSITION_ITERATIONS <= 72, "base twists of the sub-iterations in Sb + Sinv");
static_assert(offsetof(WaveShared, Sinv) == offsetof(WaveShared, Sb) + 36 * sizeof(float), "Sb and Sinv are one 72-float block");

// ---- wave collective: projected Gauss-Seidel sweep in impulse space
//      w_i = b_i + sum_j W_ij lam_j is kept up to date by every lane; rows are relaxed in order r = 0..R-1.
#if defined(LS_EMU)
static inline void wc_pgs(WaveShared& sh, LaneRegs* L, int iters) {
    const int R = LS_MAXR;
    float lam[LS_MAXR], w[LS_MAXR];
    for (int i = 0; i < R; ++i) { lam[i] = 0.0f; w[i] = ls_slot_active(sh, i) ? L[i].brow : 0.0f; }
    for (int it = 0; it < iters; ++it)
        for (int r = 0; r < R; ++r) {
            if (!ls_slot_active(sh, r)) continue;
            float nl = lam[r] - w[r] / L[r].wdiag;
            int kind = L[r].row_kind;
            if (kind == 0) nl = fmaxf(nl, 0.0f);
            else if (kind == 3) nl = fmaxf(nl, 0.0f) + fminf(nl + L[r].row_rng / L[r].wdiag, 0.0f);   // boxed: lower bound pushes up, upper bound down
            else { float lim = sh.mu * lam[r - kind]; nl = clampf(nl, -lim, lim); }
            const float old = lam[r];
            lam[r] = nl;
            for (int i = 0; i < R; ++i) if (ls_slot_active(sh, i)) w[i] = fmaf(L[i].W[r], nl, fmaf(-L[i].W[r], old, w[i]));   // the GPU form's two FMAs
        }
    for (int i = 0; i < R; ++i) if (ls_slot_active(sh, i)) sh.lam[i] = lam[i];
}
#else
LS_FN float ls_readlane(float v, int srclane) {  // srclane is wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}
// One Gauss-Seidel relaxation of slot R.  The impulses are wave-uniform data and live in SCALAR registers (sl[R]); a lane only owns the
// residual w of its row.  Every lane evaluates the candidate of slot R from its own w (only lane R's is real), readlane broadcasts it,
// and all residuals move by W[R] * (new - old) as two FMAs: - W[R] * old before the broadcast, + W[R] * new after it.  5-6 vector
// instructions per row (a per-lane impulse register with readlane / writelane hand-offs needed 7-9; at four waves per SIMD the
// sweep's time is its instruction count).  The clamp is specialised by the slot's static row type:
//   normal    lam >= 0
//   friction  |lam| <= mu * lam_n, lam_n = sl[first slot of the contact]
//   limit     two-sided velocity interval [L, L + rng]: lam = raw - clamp(raw, -rng / d, 0)   (push up at L, push down at the upper bound)
enum { LS_ROW_NORMAL = 0, LS_ROW_FRICTION = 1, LS_ROW_LIMIT = 3 };
template <int R, int KIND> __device__ __forceinline__ void ls_pgs_row(const float (&W)[LS_MAXR], float (&sl)[LS_MAXR], float fric_box, float inv_d,
                                                                     float neg_rng_d, float& w) {
    const float old = sl[R];
    const float raw = fmaf(-w, inv_d, old);
    w = fmaf(-W[R], old, w);
    float nl;
    if constexpr (KIND == LS_ROW_NORMAL) nl = fmaxf(raw, 0.0f);
    else if constexpr (KIND == LS_ROW_FRICTION) nl = __builtin_amdgcn_fmed3f(raw, -fric_box, fric_box);
    else nl = raw - __builtin_amdgcn_fmed3f(raw, neg_rng_d, 0.0f);
    const float s_nl = ls_readlane(nl, R);
    sl[R] = s_nl;
    w = fmaf(W[R], s_nl, w);
}
// contacts K0 .. nc-1 (three slots each), then limit rows I0 .. nlim-1: compile-time recursion, one uniform branch per contact / limit row
template <int K0> __device__ __forceinline__ void ls_pgs_contacts(int nc, const float (&W)[LS_MAXR], float (&sl)[LS_MAXR], float cf, float inv_d, float& w) {
    if constexpr (K0 < LS_MAXC) {
        if (K0 < nc) {
            ls_pgs_row<3 * K0, LS_ROW_NORMAL>(W, sl, 0.0f, inv_d, 0.0f, w);
            const float box = cf * sl[3 * K0];
            ls_pgs_row<3 * K0 + 1, LS_ROW_FRICTION>(W, sl, box, inv_d, 0.0f, w);
            ls_pgs_row<3 * K0 + 2, LS_ROW_FRICTION>(W, sl, box, inv_d, 0.0f, w);
            ls_pgs_contacts<K0 + 1>(nc, W, sl, cf, inv_d, w);
        }
    }
}
// joint-limit slots three per branch; a slot past nlim in the last triple is relaxed too, which is a no-op: its lane holds w = 0, 1/d = 0 and
// an impulse of 0, so the candidate is 0 and every residual moves by W * 0 (W of a stale slot is finite: see ph_load_a on the row array's tail)
template <int I0> __device__ __forceinline__ void ls_pgs_limits(int nlim, const float (&W)[LS_MAXR], float (&sl)[LS_MAXR], float inv_d, float neg_rng_d, float& w) {
    if constexpr (I0 < LSIM_NUM_DOF) {
        if (I0 < nlim) {
            ls_pgs_row<LS_LIM0 + I0, LS_ROW_LIMIT>(W, sl, 0.0f, inv_d, neg_rng_d, w);
            ls_pgs_row<LS_LIM0 + I0 + 1, LS_ROW_LIMIT>(W, sl, 0.0f, inv_d, neg_rng_d, w);
            ls_pgs_row<LS_LIM0 + I0 + 2, LS_ROW_LIMIT>(W, sl, 0.0f, inv_d, neg_rng_d, w);
            ls_pgs_limits<I0 + 3>(nlim, W, sl, inv_d, neg_rng_d, w);
        }
    }
}
// W[j] = a_lane . z_j + Jl_lane . y_j[leg_lane] for slots j = J0 .. END-1 while j < cnt (compile-time recursion; the active slots of a
// range are contiguous); y_j is zero on every leg but row j's own, so no leg comparison is needed
template <int J0> __device__ __forceinline__ void ls_delassus_row(const WaveShared& sh, int lane, int lo, const float (&jb)[6], float jl0, float jl1, float jl2,
                                                                  float (&W)[LS_MAXR], float& wd) {
    const float* Y = sh.u.c.Y[J0];
    float w = 0.0f;
    for (int k = 0; k < 6; ++k) w += jb[k] * Y[k];
    w += jl0 * Y[lo] + jl1 * Y[lo + 1] + jl2 * Y[lo + 2];
    if (J0 == lane) { w += 1e-6f; wd = w; }   // constraint-force mixing keeps the diagonal positive
    W[J0] = w;
}
// three slots per branch (a contact's rows, a leg's limit rows: the slot counts are multiples of three), so that the LDS reads of three
// rows are in flight together instead of one round trip per row
template <int J0, int END> __device__ __forceinline__ void ls_delassus_rows(const WaveShared& sh, int cnt, int lane, int lo, const float (&jb)[6],
                                                                           float jl0, float jl1, float jl2, float (&W)[LS_MAXR], float& wd) {
    if constexpr (J0 < END) {
        if (J0 < cnt) {
            ls_delassus_row<J0>(sh, lane, lo, jb, jl0, jl1, jl2, W, wd);
            ls_delassus_row<J0 + 1>(sh, lane, lo, jb, jl0, jl1, jl2, W, wd);
            ls_delassus_row<J0 + 2>(sh, lane, lo, jb, jl0, jl1, jl2, W, wd);
            ls_delassus_rows<J0 + 3, END>(sh, cnt, lane, lo, jb, jl0, jl1, jl2, W, wd);
        }
    }
}
// sum over slots [R0, R0 + N) of Y[r][lane] * lam_r in slot order, every read issued before the first use; lanes past the 18 velocities idle
template <int R0, int N> __device__ __forceinline__ float ls_apply_rows(const WaveShared& sh, const float (&sl)[LS_MAXR], int lane, float acc) {
    if (lane >= LS_NV) return acc;
    float y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = sh.u.c.Y[R0 + i][lane];
#pragma unroll
    for (int i = 0; i < N; ++i) acc += y[i] * sl[R0 + i];
    return acc;
}
// net contact force on body `lane`: contacts K0 .. nc-1, one uniform branch per contact (compile-time recursion); within a contact the three
// axes are selects, no branch on the body test
template <int K0> __device__ __forceinline__ void ls_contact_force(const WaveShared& sh, const float (&sl)[LS_MAXR], int nc, int lane, float idt, V3& f) {
    if constexpr (K0 < LS_MAXC) {
        if (K0 < nc) {
            const bool mine = sh.cbody[K0] == lane;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const V3 g = f + v3p(sh.u.c.dirs[3 * K0 + a]) * (sl[3 * K0 + a] * idt);
                f.x = mine ? g.x : f.x; f.y = mine ? g.y : f.y; f.z = mine ? g.z : f.z;      // per component: a select of whole structs goes through memory
            }
            ls_contact_force<K0 + 1>(sh, sl, nc, lane, idt, f);
        }
    }
}
// ---- Delassus build on the matrix core (north star: "MFMA ... for the small dense Jacobian / mass-matrix contractions where rocprof shows it
// wins over scalar FMA").  W = J' Y^T with J'_i = [a_i (6) | Jl_i in the slots of its own leg (12)] and Y_j in the same layout is a dense
// R x 18 by 18 x R product.  With at most 4 contacts and 4 limit rows (slots 0..11 and LS_LIM0..LS_LIM0+3) the active rows fit ONE 16 x 16 tile:
// five v_mfma_f32_16x16x4_f32 (inner dimension 18 padded to 20) replace up to 16 x 9 FMAs + as many LDS operand reads per lane.  Around
// them: J' goes to LDS as dense 18-vectors (A operand: lane (i, k) reads J'[i][4 s + k]), Y is read where it is (B operand), and the D tile
// returns through LDS so that lane = slot ends up with its row in registers, which is what the sweep consumes.  The scratch lives in the
// kinematics arrays (R .. Fb), dead between ph_rows and the next sub-step's kinematics.  tools/micro/delassus_mfma.hip measures the two forms
// side by side at kernel A's occupancy (profiles/r03_delassus_mfma.json: 1.8 vs 3.3-5.0 us per build for 6-16 rows); for more rows the
// D tiles need 4-9 KB of LDS the kernel does not have at 16 robots per CU, so the FMA rows above stay for those (9 % of the sub-steps).
#define LS_DELASSUS_MFMA_NC 4
#if !defined(LS_EMU)
typedef float ls_v4f_t __attribute__((ext_vector_type(4)));
LS_FN int ls_tile_row_of_slot(int slot) { return slot < 12 ? slot : (slot >= LS_LIM0 && slot < LS_LIM0 + 4 ? 12 + slot - LS_LIM0 : -1); }
// vk >= 0 (TGS): the lane is a VELOCITY lane -- its "row" is the unit vector of generalized velocity vk, W[j] = Y[j][vk] (see wc_delassus_tgs)
__device__ __forceinline__ void ls_delassus_mfma16(WaveShared& sh, int lane, bool act, int leg, const float (&jb)[6], float jl0, float jl1, float jl2,
                                                   float (&W)[LS_MAXR], float& wd, int vk = -1) {
    float* Jd = &sh.R[0][0];                 // [16][18]
    float* Wt = Jd + 16 * 18;                // [16][17]
    static_assert(sizeof(sh.R) + sizeof(sh.p) + sizeof(sh.S) + sizeof(sh.V) + sizeof(sh.Ab) + sizeof(sh.Fb) >= (16 * 18 + 16 * 17) * sizeof(float), "scratch of the MFMA Delassus build");
    const int t = ls_tile_row_of_slot(lane);
    if (t >= 0) {                            // jb / jl are zero for an inactive slot: its tile row is zero
        float* d = Jd + 18 * t;
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = jb[k];
#pragma unroll
        for (int l = 0; l < 4; ++l) { d[6 + 3 * l] = leg == l ? jl0 : 0.0f; d[7 + 3 * l] = leg == l ? jl1 : 0.0f; d[8 + 3 * l] = leg == l ? jl2 : 0.0f; }
    }
    LS_WAVE_SYNC();
    const int r16 = lane & 15, kq = lane >> 4;
    const int bslot = r16 < 12 ? r16 : LS_LIM0 + r16 - 12;          // B rows: Y of the slot that tile column r16 stands for
    ls_v4f_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int k = 4 * s + kq;
        const int kk = k < LS_NV ? k : 0;
        const float a = Jd[18 * r16 + kk], b = sh.u.c.Y[bslot][kk];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k < LS_NV ? a : 0.0f, k < LS_NV ? b : 0.0f, acc, 0, 0, 0);
    }
    // D[i = 4 kq + r][j = r16]; the constraint-force-mixing term on the diagonal
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[17 * (4 * kq + r) + r16] = acc[r] + ((4 * kq + r) == r16 ? 1e-6f : 0.0f);
    LS_WAVE_SYNC();
    // a row lane reads its row of the D tile, a velocity lane column vk of Y at the 16 slots of the tile: one address pattern, 16 reads
    const float* row = vk >= 0 ? &sh.u.c.Y[0][vk] : Wt + 17 * (t >= 0 ? t : 0);
    const float* rowl = vk >= 0 ? &sh.u.c.Y[LS_LIM0][vk] : row + 12;
    const int stride = vk >= 0 ? LS_NV : 1;
#pragma unroll
    for (int j = 0; j < 12; ++j) W[j] = row[j * stride];
#pragma unroll
    for (int j = 0; j < 4; ++j) W[LS_LIM0 + j] = rowl[j * stride];
    W[LS_LIM0 + 4] = 0.0f; W[LS_LIM0 + 5] = 0.0f;     // the sweep relaxes limit slots in triples: slots past nlim must hold something finite
    if (act) wd = row[t];
    LS_WAVE_SYNC();                          // the scratch is the next phase's to overwrite only after every lane has read its row
}
template <int I>
__device__ __forceinline__ void ls_delassus_mfma32_row(WaveShared& sh, int lane, int vk, float (&W)[LS_MAXR]) {
    float* Jd = &sh.R[0][0];
    const int r16 = lane & 15, kq = lane >> 4;
    ls_v4f_t accA = {0.0f, 0.0f, 0.0f, 0.0f}, accB = accA;           // tiles (I, 0) and (I, 1)
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int k = 4 * s + kq;
        const bool in = k < LS_NV;
        const int kk = in ? k : 0;
        const float a = in ? Jd[18 * (16 * I + r16) + kk] : 0.0f;
        const float b0 = in ? sh.u.c.Y[r16][kk] : 0.0f, b1 = in ? sh.u.c.Y[16 + r16][kk] : 0.0f;
        accA = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, accB, 0, 0, 0);
    }
    LS_WAVE_SYNC();                      // every lane has read its A operands of this tile row: its half of the scratch now takes D tiles
    float* Wt = Jd + 288 * I;             // [16][17]
    const bool mine = vk < 0 && (lane >> 4) == I && lane < 32;
    // D[i = 4 kq + r][j = r16]; the constraint-force-mixing term on the diagonal (tile (I, I))
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[17 * (4 * kq + r) + r16] = accA[r] + ((I == 0 && (4 * kq + r) == r16) ? 1e-6f : 0.0f);
    LS_WAVE_SYNC();
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float t = Wt[17 * r16 + j]; W[j] = mine ? t : W[j]; }
    LS_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[17 * (4 * kq + r) + r16] = accB[r] + ((I == 1 && (4 * kq + r) == r16) ? 1e-6f : 0.0f);
    LS_WAVE_SYNC();
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float t = Wt[17 * r16 + j]; W[16 + j] = mine ? t : W[16 + j]; }
    LS_WAVE_SYNC();
}
// Round 6, BUILT, CORRECT (GPU physics suite green with it) AND NOT SHIPPED (-DLS_DELASSUS_MFMA32 enables it): kernel A 0.1096 -> 0.1148 ms flat, 0.1194 ->
// 0.1230 stairs, interleaved on one lease (profiles/r06_kernel_a_ab.txt).  The micro-benchmark's 3.5 against 6.4-7.8 us per build assumed LDS for the whole D
// matrix; squeezed into the 582 free floats the transposition needs ten barriers and 64 register selects per lane (conditional stores into W in divergent
// branches sent the whole array to scratch memory), on exactly the waves the launch waits for.  The FMA rows stay for more than 16 rows.
// The same for up to 32 rows -- every contact count, up to 8 limit rows (slots 0 .. 31, tile row = slot): 2 x 2 tiles, 20 MFMAs.  The robots that
// need it are the ones with five or more contacts, i.e. the SLOWEST waves of a launch that is one round of waves and ends with its slowest (DESIGN.md
// section 6): the FMA rows cost them 6.4-7.8 us per build against 3.5 on the matrix core (tools/micro/delassus_mfma, profiles/r03_delassus_mfma.json).
// Round 3 stopped at 16 rows because the D tiles of 32 rows need 4 KB of LDS.  Here the scratch (the kinematics arrays, 582 floats) holds J' [32][18]
// (576 floats) and is then reused HALF BY HALF: tile row 1 first (its A operands, rows 16 .. 31 of J', are then dead: 288 floats take one [16][17] D tile at a
// time), then tile row 0 the same way in the lower half -- only two accumulator tiles (8 registers) are alive at any time, the kernel sits at its 128-register
// limit (four tiles alive pushed the scalar-register spills out of the vector file into scratch memory).
__device__ __forceinline__ void ls_delassus_mfma32(WaveShared& sh, int lane, bool act, int leg, const float (&jb)[6], float jl0, float jl1, float jl2,
                                                   float (&W)[LS_MAXR], float& wd, int vk = -1) {
    float* Jd = &sh.R[0][0];                 // [32][18]; each half [16][18] later one [16][17] D tile
    static_assert(sizeof(sh.R) + sizeof(sh.p) + sizeof(sh.S) + sizeof(sh.V) + sizeof(sh.Ab) + sizeof(sh.Fb) >= 32 * 18 * sizeof(float), "scratch of the 32-row MFMA Delassus build");
    static_assert(16 * 17 <= 16 * 18 && LS_LIM0 == 24 && LS_MAXR == 36, "a D tile fits where half of J' was; slots 0 .. 31 are tile rows 0 .. 31");
    if (lane < 32) {                         // jb / jl are zero for an inactive slot: its tile row is zero
        float* d = Jd + 18 * lane;
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = jb[k];
#pragma unroll
        for (int l = 0; l < 4; ++l) { d[6 + 3 * l] = leg == l ? jl0 : 0.0f; d[7 + 3 * l] = leg == l ? jl1 : 0.0f; d[8 + 3 * l] = leg == l ? jl2 : 0.0f; }
    }
    // velocity lanes: column vk of Y at the 32 slots (no D involved); lanes that hold no row: something finite
    if (vk >= 0) {
#pragma unroll
        for (int j = 0; j < 32; ++j) W[j] = sh.u.c.Y[j][vk];
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) W[j] = 0.0f;          // row lanes: filled below; lanes that hold no row: something finite
    }
    LS_WAVE_SYNC();
    ls_delassus_mfma32_row<1>(sh, lane, vk, W);      // tile row 1 first: its half of J' is the first to die
    ls_delassus_mfma32_row<0>(sh, lane, vk, W);
    W[32] = 0.0f; W[33] = 0.0f; W[34] = 0.0f; W[35] = 0.0f;     // the sweep relaxes limit slots in triples: slots past nlim must hold something finite
    if (act) {                                 // (act implies lane < 32 here: slot < 3 nc <= 24 or a limit slot < LS_LIM0 + 8); the diagonal entry by selects
        float d = W[0];
#pragma unroll
        for (int j = 1; j < 32; ++j) d = lane == j ? W[j] : d;
        wd = d;
    }
}
#endif

// GPU form: Delassus row and sweep fused so that the 36-entry row lives in registers only between here and the end of
// the sweep (written unconditionally: no liveness across sub-steps); rows relaxed in slot order, impulse broadcast by readlane.
LS_FN void wc_delassus_pgs(WaveShared& sh, const LaneRegs& rg, int lane, int iters, float dt) {
    const int nc = LS_UNIFORM(sh.nc), nlim = LS_UNIFORM(sh.nlim);
    const bool act = ls_slot_active(sh, lane);
    const int leg = rg.row_leg;
    const int lo = (act && leg >= 0) ? 6 + 3 * leg : 6;
    const bool has_leg = act && leg >= 0;
    const float jl0 = has_leg ? rg.Jl[0] : 0.0f, jl1 = has_leg ? rg.Jl[1] : 0.0f, jl2 = has_leg ? rg.Jl[2] : 0.0f;
    float jb[6];
    for (int k = 0; k < 6; ++k) jb[k] = act ? rg.Jb[k] : 0.0f;
    float W[LS_MAXR];                   // entries of inactive slots stay unset: the sweep never touches them (36 zero-fills saved per sub-step)
    float wd = 1.0f;
#if !defined(LS_NO_DELASSUS_MFMA)
    if (nc <= LS_DELASSUS_MFMA_NC && nlim <= 4) {
        ls_delassus_mfma16(sh, lane, act, leg, jb, jl0, jl1, jl2, W, wd);      // <= 16 rows (91 % of the sub-steps): one 16 x 16 MFMA tile
    } else
#if defined(LS_DELASSUS_MFMA32)      /* measured slower: see ls_delassus_mfma32 */
    if (nlim <= 8) {
        ls_delassus_mfma32(sh, lane, act, leg, jb, jl0, jl1, jl2, W, wd);      // <= 32 rows: 2 x 2 tiles (round 6)
    } else
#endif
#endif
    {
        ls_delassus_rows<0, LS_LIM0>(sh, 3 * nc, lane, lo, jb, jl0, jl1, jl2, W, wd);
        ls_delassus_rows<LS_LIM0, LS_MAXR>(sh, LS_LIM0 + nlim, lane, lo, jb, jl0, jl1, jl2, W, wd);
    }
    float w = act ? rg.brow : 0.0f;
    const float inv_d = act ? ls_rcp(wd) : 0.0f;
    const float cf = sh.mu;                                                 // only the friction slots use it
    const float neg_rng_d = act ? -(rg.row_rng * inv_d) : 0.0f;            // only the limit slots use it (their range is finite)
    float sl[LS_MAXR] = {};                                                 // the impulses: wave-uniform, scalar registers
    for (int it = 0; it < iters; ++it) {
        // the counts are re-materialised every sweep: loop-invariant, the 20 branch conditions on them would be hoisted out of the loop
        // as 20 lane masks in 40 scalar registers, which pushes the impulses out of the scalar file (spills through v_readlane)
        int nc_it = nc, nlim_it = nlim;
        asm volatile("" : "+s"(nc_it), "+s"(nlim_it));
        ls_pgs_contacts<0>(nc_it, W, sl, cf, inv_d, w);
        ls_pgs_limits<0>(nlim_it, W, sl, inv_d, neg_rng_d, w);
    }
    // ---- v+ = vfree + M^-1 J^T lam (lane = generalized velocity index) and the net contact force per body (lane = body, LR:944), here at the
    //      end of the solver phase with the impulses still in scalar registers: no impulse array in LDS, no extra phase, and one straight-line
    //      block per contact / limit-row count, so that all the LDS reads of a block are in flight together (a loop over the active rows paid
    //      one LDS round trip per remainder row: 7.8 us of kernel A for two dozen FMAs)
    const float idt = ls_rcp(dt);
    float acc = 0.0f;
    V3 f = v3(0, 0, 0);
    switch (nc) {
#define LS_C(N) case N: acc = ls_apply_rows<0, 3 * N>(sh, sl, lane, acc); break;
        LS_C(1) LS_C(2) LS_C(3) LS_C(4) LS_C(5) LS_C(6) LS_C(7) LS_C(8)
#undef LS_C
        default: break;
    }
    if (lane < LS_NB) ls_contact_force<0>(sh, sl, nc, lane, idt, f);
    switch ((nlim + 2) / 3) {       // whole triples: the impulse of a slot past nlim is 0 and its stale row finite
#define LS_L(T) case T: acc = ls_apply_rows<LS_LIM0, 3 * T>(sh, sl, lane, acc); break;
        LS_L(1) LS_L(2) LS_L(3) LS_L(4)
#undef LS_L
        default: break;
    }
    static_assert(LS_MAXC == 8 && LSIM_NUM_DOF == 12, "cases above");
    if (lane < LS_NV) {
        if (lane < 6) sh.ab[lane] = acc;
        sh.vnew[lane] = sh.vfree[lane] + acc;
    }
    if (lane < LS_NB) v3st(sh.cf[lane], f);
}

// ==== Temporal Gauss-Seidel (lsim_config.solver_type 1: what the reference configures, LRC:245-247) =========================================
// External forces act once, over the whole dt (the free velocity, as for PGS); the sim_dt is then split into nsub = num_position_iterations
// sub-iterations of h = dt / nsub.  Each one (i) relaxes every row ONCE, in slot order, against a target re-derived from the configuration
// reached so far (contact gap, distance of the joint to its stops), impulses accumulated and projected as totals, (ii) advances gaps, joints
// and the base pose by h of the velocity it arrived at.  J and M^-1 J^T stay those of the start of the step (oracle/orc_physics.c, TGS
// branch: same scheme, dense fp64; its comment says why the free acceleration is not ramped over the sub-iterations).  In this kernel's form a lane owns the VELOCITY u_i = J_i v of its row instead of a residual: a relaxation of slot R moves every
// u_i by W_iR (new - old) -- the same two FMAs as in the PGS sweep -- and (ii)'s target and (iii)'s gap update are per-lane arithmetic on u_i
// (the gap of a normal row advances by h u_i, a limit row's joint angle too: its J is a unit vector).
// Lanes LS_VEL0 .. LS_VEL0 + 17 carry the generalized velocity itself the same way: lane LS_VEL0 + k holds "row" e_k, i.e. W[j] = Y[j][k], so
// the sweep's FMAs keep the velocity change of the impulses up to date for free (these lanes idle in the PGS form) -- the apply pass at
// the end of the PGS solver disappears, and the base twist after each sub-iteration, which the pose update needs, is one LDS store.
LS_FN void wc_delassus_tgs(const LsCtx& cx, WaveShared& sh, const LaneRegs& rg, int lane, int nsub, float dt) {
    const lsim_config& c = cx.cfg;
    const int nc = LS_UNIFORM(sh.nc), nlim = LS_UNIFORM(sh.nlim);
    const bool act = ls_slot_active(sh, lane);
    const int vk = (lane >= LS_VEL0 && lane < LS_VEL0 + LS_NV) ? lane - LS_VEL0 : -1;
    const int leg = rg.row_leg;
    const bool has_leg = act && leg >= 0;
    int lo = has_leg ? 6 + 3 * leg : 6;
    float jl0 = has_leg ? rg.Jl[0] : 0.0f, jl1 = has_leg ? rg.Jl[1] : 0.0f, jl2 = has_leg ? rg.Jl[2] : 0.0f;
    float jb[6];
    for (int k = 0; k < 6; ++k) jb[k] = act ? rg.Jb[k] : 0.0f;
    if (vk >= 0) {     // the FMA rows evaluate jb . Y[0..5] + jl . Y[lo..lo+2]: pick out component vk (lo + 2 must stay inside the 18-vector)
        for (int k = 0; k < 6; ++k) jb[k] = vk == k ? 1.0f : 0.0f;
        lo = vk < 6 ? 6 : (vk < 15 ? vk : 15);
        jl0 = (vk >= 6 && vk - lo == 0) ? 1.0f : 0.0f; jl1 = vk - lo == 1 ? 1.0f : 0.0f; jl2 = vk - lo == 2 ? 1.0f : 0.0f;
    }
    float W[LS_MAXR];
    float wd = 1.0f;
#if !defined(LS_NO_DELASSUS_MFMA)
    if (nc <= LS_DELASSUS_MFMA_NC && nlim <= 4) {
        ls_delassus_mfma16(sh, lane, act, leg, jb, jl0, jl1, jl2, W, wd, vk);
    } else
#if defined(LS_DELASSUS_MFMA32)      /* measured slower: see ls_delassus_mfma32 */
    if (nlim <= 8) {
        ls_delassus_mfma32(sh, lane, act, leg, jb, jl0, jl1, jl2, W, wd, vk);
    } else
#endif
#endif
    {
        ls_delassus_rows<0, LS_LIM0>(sh, 3 * nc, lane, lo, jb, jl0, jl1, jl2, W, wd);
        ls_delassus_rows<LS_LIM0, LS_MAXR>(sh, LS_LIM0 + nlim, lane, lo, jb, jl0, jl1, jl2, W, wd);
    }
    const float inv_d = act ? ls_rcp(wd) : 0.0f;
    const float cf = sh.mu;
    const float hs = dt * ls_rcp((float)nsub), ihs = (float)nsub * ls_rcp(dt);
    const bool is_normal = rg.row_kind == 0, is_limit = rg.row_kind == 3;     // -1 on every lane that holds no row
    float u = act ? rg.brow : 0.0f;                                           // row velocity J v, starting from J vfree (velocity lanes: the change of v_k by the impulses)
    float ga = rg.tg_a, gb = rg.tg_b;
    const float vmax = rg.row_rng;
    float vsum = 0.0f;
    float sl[LS_MAXR] = {};                                                   // the accumulated impulses: wave-uniform, scalar registers
    for (int s = 0; s < nsub; ++s) {
        // targets from the configuration reached so far (every lane evaluates both forms; selects, no branches)
        const float tn = ga >= 0.0f ? -ga * ihs : fminf(c.max_depenetration_velocity, c.erp * fmaxf(-ga - c.contact_slop, 0.0f) * ihs);
        float Lb = -vmax, Ub = vmax;
        if (ga < 0.1f) Lb = fmaxf(Lb, ga >= 0.0f ? -ga * ihs : fminf(1.0f, c.erp * (-ga) * ihs));
        if (gb < 0.1f) Ub = fminf(Ub, gb >= 0.0f ? gb * ihs : -fminf(1.0f, c.erp * (-gb) * ihs));
        Ub = fmaxf(Ub, Lb);
        const float tgt = is_normal ? tn : (is_limit ? Lb : 0.0f);
        const float neg_rng_d = is_limit ? -((Ub - Lb) * inv_d) : 0.0f;
        float w = u - tgt;
        int nc_it = nc, nlim_it = nlim;                                       // re-materialised per sweep (see wc_delassus_pgs)
        asm volatile("" : "+s"(nc_it), "+s"(nlim_it));
        ls_pgs_contacts<0>(nc_it, W, sl, cf, inv_d, w);
        ls_pgs_limits<0>(nlim_it, W, sl, inv_d, neg_rng_d, w);
        u = w + tgt;
        const float adv = hs * u;
        ga += adv; gb -= adv;
        if (vk >= 0) {
            vsum += u;
            if (vk < 6) ls_tgs_base_twist(sh, s)[vk] = u;
        }
    }
    // lsim_config.tgs_limit_passes: velocity-level passes over the limit rows alone, against the bounds of the configuration reached; the
    // contact impulses stay as they are, every lane's row velocity (and the velocity lanes) follow through W as in the sweeps above.  The
    // positions are done: only the velocity the step hands on changes.
    if (nlim > 0) {
        for (int it = 0; it < c.tgs_limit_passes; ++it) {
            float Lb = -vmax, Ub = vmax;
            if (ga < 0.1f) Lb = fmaxf(Lb, ga >= 0.0f ? -ga * ihs : fminf(1.0f, c.erp * (-ga) * ihs));
            if (gb < 0.1f) Ub = fminf(Ub, gb >= 0.0f ? gb * ihs : -fminf(1.0f, c.erp * (-gb) * ihs));
            Ub = fmaxf(Ub, Lb);
            const float tgt = is_limit ? Lb : 0.0f;
            const float neg_rng_d = is_limit ? -((Ub - Lb) * inv_d) : 0.0f;
            float w = u - tgt;
            int nlim_it = nlim;
            asm volatile("" : "+s"(nlim_it));
            ls_pgs_limits<0>(nlim_it, W, sl, inv_d, neg_rng_d, w);
            u = w + tgt;
        }
    }
    const float idt = ls_rcp(dt);
    V3 f = v3(0, 0, 0);
    if (lane < LS_NB) ls_contact_force<0>(sh, sl, nc, lane, idt, f);
    if (vk >= 0) {
        if (vk < 6) sh.ab[vk] = u;                      // base twist change: the integrator applies the coupling - G_l dvb to the joints
        sh.vnew[vk] = sh.vfree[vk] + u;
        ls_tgs_vel_sum(sh)[vk] = vsum;
    }
    if (lane < LS_NB) v3st(sh.cf[lane], f);
}
#endif

#if defined(LS_EMU)
// lane emulator: the same scheme with the rows' data in arrays (rows relaxed in slot order, fp32, the GPU form's two FMAs per update)
static inline void wc_tgs(const LsCtx& cx, WaveShared& sh, LaneRegs* L, int nsub, float dt) {
    const lsim_config& c = cx.cfg;
    const int R = LS_MAXR;
    const float hs = dt * ls_rcp((float)nsub), ihs = (float)nsub * ls_rcp(dt);
    float lam[LS_MAXR], u[LS_MAXR], ga[LS_MAXR], gb[LS_MAXR], w[LS_MAXR], tgt[LS_MAXR], nrd[LS_MAXR], dv[LS_NV], vsum[LS_NV];
    bool act[LS_MAXR];
    for (int i = 0; i < R; ++i) {
        act[i] = ls_slot_active(sh, i);
        lam[i] = 0.0f; w[i] = 0.0f; tgt[i] = 0.0f; nrd[i] = 0.0f;
        u[i] = act[i] ? L[i].brow : 0.0f;
        ga[i] = act[i] ? L[i].tg_a : 0.0f; gb[i] = act[i] ? L[i].tg_b : 0.0f;
    }
    for (int k = 0; k < LS_NV; ++k) { dv[k] = 0.0f; vsum[k] = 0.0f; }
    for (int s = 0; s < nsub; ++s) {
        for (int i = 0; i < R; ++i) {
            if (!act[i]) continue;
            const int kind = L[i].row_kind;
            float t = 0.0f;
            nrd[i] = 0.0f;
            if (kind == 0) t = ga[i] >= 0.0f ? -ga[i] * ihs : fminf(c.max_depenetration_velocity, c.erp * fmaxf(-ga[i] - c.contact_slop, 0.0f) * ihs);
            else if (kind == 3) {
                const float vmax = L[i].row_rng;
                float Lb = -vmax, Ub = vmax;
                if (ga[i] < 0.1f) Lb = fmaxf(Lb, ga[i] >= 0.0f ? -ga[i] * ihs : fminf(1.0f, c.erp * (-ga[i]) * ihs));
                if (gb[i] < 0.1f) Ub = fminf(Ub, gb[i] >= 0.0f ? gb[i] * ihs : -fminf(1.0f, c.erp * (-gb[i]) * ihs));
                Ub = fmaxf(Ub, Lb);
                t = Lb;
                nrd[i] = (Ub - Lb) / L[i].wdiag;
            }
            tgt[i] = t;
            w[i] = u[i] - t;
        }
        for (int r = 0; r < R; ++r) {
            if (!act[r]) continue;
            float nl = lam[r] - w[r] / L[r].wdiag;
            const int kind = L[r].row_kind;
            if (kind == 0) nl = fmaxf(nl, 0.0f);
            else if (kind == 3) nl = fmaxf(nl, 0.0f) + fminf(nl + nrd[r], 0.0f);
            else { float lim = sh.mu * lam[r - kind]; nl = clampf(nl, -lim, lim); }
            const float old = lam[r];
            lam[r] = nl;
            for (int i = 0; i < R; ++i) if (act[i]) w[i] = fmaf(L[i].W[r], nl, fmaf(-L[i].W[r], old, w[i]));
            for (int k = 0; k < LS_NV; ++k) dv[k] = fmaf(sh.u.c.Y[r][k], nl, fmaf(-sh.u.c.Y[r][k], old, dv[k]));
        }
        for (int i = 0; i < R; ++i) {
            if (!act[i]) continue;
            u[i] = w[i] + tgt[i];
            const float adv = hs * u[i];
            ga[i] += adv; gb[i] -= adv;
        }
        for (int k = 0; k < LS_NV; ++k) vsum[k] += dv[k];
        for (int k = 0; k < 6; ++k) ls_tgs_base_twist(sh, s)[k] = dv[k];
    }
    if (sh.nlim > 0)        // lsim_config.tgs_limit_passes: the limit rows alone, velocity level, bounds of the configuration reached
        for (int it = 0; it < c.tgs_limit_passes; ++it) {
            for (int i = 0; i < R; ++i) {
                if (!act[i]) continue;
                float t = 0.0f;
                nrd[i] = 0.0f;
                if (L[i].row_kind == 3) {
                    const float vmax = L[i].row_rng;
                    float Lb = -vmax, Ub = vmax;
                    if (ga[i] < 0.1f) Lb = fmaxf(Lb, ga[i] >= 0.0f ? -ga[i] * ihs : fminf(1.0f, c.erp * (-ga[i]) * ihs));
                    if (gb[i] < 0.1f) Ub = fminf(Ub, gb[i] >= 0.0f ? gb[i] * ihs : -fminf(1.0f, c.erp * (-gb[i]) * ihs));
                    Ub = fmaxf(Ub, Lb);
                    t = Lb;
                    nrd[i] = (Ub - Lb) / L[i].wdiag;
                }
                tgt[i] = t;
                w[i] = u[i] - t;
            }
            for (int r = 0; r < R; ++r) {
                if (!act[r] || L[r].row_kind != 3) continue;
                float nl = lam[r] - w[r] / L[r].wdiag;
                nl = fmaxf(nl, 0.0f) + fminf(nl + nrd[r], 0.0f);
                const float old = lam[r];
                lam[r] = nl;
                for (int i = 0; i < R; ++i) if (act[i]) w[i] = fmaf(L[i].W[r], nl, fmaf(-L[i].W[r], old, w[i]));
                for (int k = 0; k < LS_NV; ++k) dv[k] = fmaf(sh.u.c.Y[r][k], nl, fmaf(-sh.u.c.Y[r][k], old, dv[k]));
            }
            for (int i = 0; i < R; ++i) if (act[i]) u[i] = w[i] + tgt[i];
        }
    for (int i = 0; i < R; ++i) if (act[i]) sh.lam[i] = lam[i];
    for (int k = 0; k < LS_NV; ++k) {
        if (k < 6) sh.ab[k] = dv[k];
        sh.vnew[k] = sh.vfree[k] + dv[k];
        ls_tgs_vel_sum(sh)[k] = vsum[k];
    }
}
#endif

// ---- phase V (lane emulator; the GPU does this at the end of wc_delassus_pgs): constrained velocity v+ = vfree + M^-1 J^T lam from the
//      stored rows (lane = generalized velocity index): the base part
//      dvb = sum_r z_r lam_r is final here (and parked in sh.ab, dead since ph_free_finish); a joint gets its own-leg part here and the
//      coupling term - G_l dvb in ph_integrate, once dvb is complete
LS_FN void ph_apply_impulses(WaveShared& sh, int lane) {
    if (lane >= LS_NV) return;
    float acc = 0.0f;
    const int nc = LS_UNIFORM(sh.nc), nlim = LS_UNIFORM(sh.nlim);
    for (int r = 0; r < 3 * nc; ++r) acc += sh.u.c.Y[r][lane] * sh.lam[r];
    for (int r = LS_LIM0; r < LS_LIM0 + nlim; ++r) acc += sh.u.c.Y[r][lane] * sh.lam[r];
    if (lane < 6) sh.ab[lane] = acc;
    sh.vnew[lane] = sh.vfree[lane] + acc;
}

// ---- phase B (lane emulator; GPU: end of wc_delassus_pgs): net contact force per body, world frame (LR:944) (lane = body)
LS_FN void ph_contact_forces(WaveShared& sh, int lane, float dt) {
    if (lane >= LS_NB) return;
    V3 f = v3(0, 0, 0);
    const float idt = ls_rcp(dt);
    const int nc = LS_UNIFORM(sh.nc);
    for (int k = 0; k < nc; ++k)
        if (sh.cbody[k] == lane)
            for (int a = 0; a < 3; ++a) f = f + v3p(sh.u.c.dirs[3 * k + a]) * (sh.lam[3 * k + a] * idt);
    v3st(sh.cf[lane], f);
}

// ---- phase: integrate (lanes 0-11 joints, lane 12 base)
LS_FN void ph_integrate(const LsCtx& cx, WaveShared& sh, int lane, float dt) {
    if (lane < 12) {
        float lim = 1.5f * sh.jc_vmax[lane];     // the limit itself is a constraint row; this only bounds solver residue
        const int l = lane / 3, k = lane - 3 * l;
        float vj = sh.vnew[6 + lane];
        for (int c = 0; c < 6; ++c) vj -= sh.G[l][6 * k + c] * sh.ab[c];      // - G_l dvb: the base impulse response on this joint (ph_apply_impulses)
        float v = clampf(vj, -lim, lim);
        sh.q[lane] += dt * v;
        sh.qd[lane] = v;
    } else if (lane == 12) {
        V3 w = v3p(sh.vnew), vl = v3p(sh.vnew + 3);
        {   // body velocity caps of the asset options (LRC:229-230; PhysX clamps there too): a safety net, never reached by a sane robot
            const float wn = sqrtf(dot(w, w)), ln = sqrtf(dot(vl, vl));
            if (cx.cfg.max_angular_velocity > 0.0f && wn > cx.cfg.max_angular_velocity) w = w * (cx.cfg.max_angular_velocity * ls_rcp(wn));
            if (cx.cfg.max_linear_velocity > 0.0f && ln > cx.cfg.max_linear_velocity) vl = vl * (cx.cfg.max_linear_velocity * ls_rcp(ln));
        }
        V3 dp = vl * dt;
        V3 vo = vl + cross(w, dp);   // velocity of the base origin after it moved by dp
        float* q = sh.root + 3;
        float qn = ls_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        float q0 = q[0] * qn, q1 = q[1] * qn, q2 = q[2] * qn, q3 = q[3] * qn;
        float h = 0.5f * dt;
        float n0 = q0 + h * (w.x * q3 + w.y * q2 - w.z * q1);
        float n1 = q1 + h * (-w.x * q2 + w.y * q3 + w.z * q0);
        float n2 = q2 + h * (w.x * q1 - w.y * q0 + w.z * q3);
        float n3 = q3 + h * (-w.x * q0 - w.y * q1 - w.z * q2);
        float nn = ls_rsqrt(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
        q[0] = n0 * nn; q[1] = n1 * nn; q[2] = n2 * nn; q[3] = n3 * nn;
        sh.root[0] += dp.x; sh.root[1] += dp.y; sh.root[2] += dp.z;
        v3st(sh.root + 7, vo);
        v3st(sh.root + 10, w);
    }
}

// ---- TGS form of the integrator: the configuration advanced with the sub-iterations, so the joint angles take the SUM of the velocities of
//      the sub-iterations -- sub-iteration s runs at  vfree + (impulse part after its sweep) -- and the base pose is stepped through the n
//      twists the solver parked (lanes 0-11 joints, lane 12 base)
LS_FN void ph_integrate_tgs(const LsCtx& cx, WaveShared& sh, int lane, float dt, int nsub) {
    const float hs = dt * ls_rcp((float)nsub);
    if (lane < 12) {
        const float lim = 1.5f * sh.jc_vmax[lane];     // bounds solver residue on the velocity the step hands on (as the PGS form)
        const int l = lane / 3, k = lane - 3 * l;
        const float* vs = ls_tgs_vel_sum(sh);
        float vj = sh.vnew[6 + lane], sj = vs[6 + lane];
        for (int c = 0; c < 6; ++c) { const float g = sh.G[l][6 * k + c]; vj -= g * sh.ab[c]; sj -= g * vs[c]; }   // - G_l dvb: base impulse response
        sh.q[lane] += dt * sh.vfree[6 + lane] + hs * sj;
        sh.qd[lane] = clampf(vj, -lim, lim);
    } else if (lane == 12) {
        const V3 wf = v3p(sh.vfree), vf = v3p(sh.vfree + 3);
        float* q = sh.root + 3;
        const float qn = ls_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        float q0 = q[0] * qn, q1 = q[1] * qn, q2 = q[2] * qn, q3 = q[3] * qn;
        V3 dp = v3(0, 0, 0), w = wf, vl = vf;
        const float h = 0.5f * hs;
        // q <- (I + h Omega(w_s)) q is linear in q: normalising once after the last sub-iteration gives what normalising after each one gives
        auto sub_iteration = [&](const float* tb) {
            w = wf + v3p(tb);
            vl = vf + v3p(tb + 3);                // velocity of the point of the base that sat at the base origin at the start of the step
            dp = dp + (vl + cross(w, dp)) * hs;
            const float n0 = q0 + h * (w.x * q3 + w.y * q2 - w.z * q1);
            const float n1 = q1 + h * (-w.x * q2 + w.y * q3 + w.z * q0);
            const float n2 = q2 + h * (w.x * q1 - w.y * q0 + w.z * q3);
            const float n3 = q3 + h * (-w.x * q0 - w.y * q1 - w.z * q2);
            q0 = n0; q1 = n1; q2 = n2; q3 = n3;
        };
        if (nsub == 4) {                          // the reference's setting: unrolled, all 24 twist components read before the dependent chain starts
            float tb[24];
            for (int k = 0; k < 24; ++k) tb[k] = ls_tgs_base_twist(sh, 0)[k];
            for (int s = 0; s < 4; ++s) sub_iteration(tb + 6 * s);
        } else {
            for (int s = 0; s < nsub; ++s) sub_iteration(ls_tgs_base_twist(sh, s));
        }
        {
            const float nn = ls_rsqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
            q0 *= nn; q1 *= nn; q2 *= nn; q3 *= nn;
        }
        // the twist the step hands on is the solver's FINAL one (sh.ab): the last sub-iteration's, plus what the limit-only velocity passes
        // (lsim_config.tgs_limit_passes) did to the base afterwards -- the joints take the same change through - G_l dvb above
        w = wf + v3p(sh.ab);
        vl = vf + v3p(sh.ab + 3);
        {   // body velocity caps of the asset options (LRC:229-230), on the velocity the step hands on
            const float wn = sqrtf(dot(w, w)), ln = sqrtf(dot(vl, vl));
            if (cx.cfg.max_angular_velocity > 0.0f && wn > cx.cfg.max_angular_velocity) w = w * (cx.cfg.max_angular_velocity * ls_rcp(wn));
            if (cx.cfg.max_linear_velocity > 0.0f && ln > cx.cfg.max_linear_velocity) vl = vl * (cx.cfg.max_linear_velocity * ls_rcp(ln));
        }
        q[0] = q0; q[1] = q1; q[2] = q2; q[3] = q3;
        sh.root[0] += dp.x; sh.root[1] += dp.y; sh.root[2] += dp.z;
        v3st(sh.root + 7, vl + cross(w, dp));
        v3st(sh.root + 10, w);
    }
}

// ---- phase B: world state of every body (rigid_body_states, LR:938) from the kinematics phase (lane = body)
// at_com (lsim_config.lin_vel_at_com): the linear velocity is the one of the body's centre of mass (PhysX's getLinearVelocity), i.e. taken at
// p + R c instead of at the link origin p; the base's c carries the per-env COM displacement (LR:1025-1028)
LS_FN V3 ls_body_com_local(const WaveShared& sh, int body) {
    V3 cl = v3p(sh.body[body].com);
    if (body == 0) cl = cl + v3p(sh.comd);
    return cl;
}
LS_FN void ph_body_states(WaveShared& sh, int lane, float* o /* [13]: this body's row of the rigid-body state tensor (position, quaternion, linear, angular velocity) */,
                          bool at_com) {
    if (lane >= LS_NB) return;
    V3 p = v3p(sh.p[lane]);
    S6 V = s6p(sh.V[lane]);
    V3 at = p;
    if (at_com) at = p + mul(m3p(sh.R[lane]), ls_body_com_local(sh, lane));
    V3 vel = V.l + cross(V.a, at);
    o[0] = sh.root[0] + p.x; o[1] = sh.root[1] + p.y; o[2] = sh.root[2] + p.z;
    float q4[4];
    if (lane == 0) { for (int k = 0; k < 4; ++k) q4[k] = sh.root[3 + k]; }
    else R_to_quat(m3p(sh.R[lane]), q4);
    for (int k = 0; k < 4; ++k) o[3 + k] = q4[k];
    v3st(o + 7, vel);
    v3st(o + 10, V.a);
}
