/*
 * orc_physics.c -- TEST INFRASTRUCTURE, not part of the product.
 *
 * CPU twin of the build's articulated-body sub-step (replaces gym.simulate(), LR:149, i.e. closed-source
 * PhysX).  PARITY UNPINNED: there is no reference source, test or golden vector for the dynamics; this
 * file is an independent, deliberately naive (dense, double precision, generic kinematic tree)
 * statement of the same mathematical model the HIP kernel implements with a structured solver, so that
 * the two can be compared (tests/test_physics_*.py) and checked against physical invariants.
 *
 * Model (DESIGN.md "Physics"):
 *   - floating base + 12 revolute joints, 17 bodies (feet are fixed children of the calves), URDF constants
 *     from lsim_robot_model; per-env payload mass / COM shift on the base (LR:591-596).
 *   - generalized velocity v = [base twist (w, v_O) in world axes about the base origin O; qd(12)].
 *   - M(q) by composite rigid bodies, bias h(q,v) by recursive Newton-Euler, both in world-aligned
 *     Pluecker coordinates about O.
 *   - semi-implicit Euler: v_free = v + dt M^-1 (tau - h); contacts / joint limits either as a velocity-level LCP
 *     solved by projected Gauss-Seidel in impulse space on W = J M^-1 J^T, q+ = q (+) dt v+ (lsim_config.solver_type 0), or by
 *     the Temporal Gauss-Seidel scheme the reference configures (solver_type 1, LRC:245-248; see the branch below).
 *   - collision: sphere-swept points (lsim_collision_point) against the triangulated height grid.
 */
#include <math.h>
#include <string.h>
#include <stdlib.h>

#include "orc_internal.h"

#define NB LSIM_NUM_BODIES
#define NV 18
#define MAXC LSIM_MAX_CONTACTS
#define MAXR (3 * MAXC + LSIM_NUM_DOF)

typedef struct { double R[9]; double p[3]; } xf_t;

static void v3cross(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static double v3dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void m3v(const double R[9], const double v[3], double o[3]) {
    for (int i = 0; i < 3; ++i) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
static void m3m(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double a = 0; for (int k = 0; k < 3; ++k) a += A[3 * i + k] * B[3 * k + j];
        C[3 * i + j] = a;
    }
}
static void quat_to_R(const double q[4], double R[9]) { /* xyzw */
    double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
static void R_to_quat(const double R[9], double q[4]) {
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) { double s = sqrt(tr + 1.0) * 2; q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; }
    else if (R[0] > R[4] && R[0] > R[8]) { double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; }
    else if (R[4] > R[8]) { double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s; }
    else { double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s; }
}
static void axis_angle_R(const double a[3], double th, double R[9]) {
    double c = cos(th), s = sin(th), t = 1 - c;
    R[0] = c + t * a[0] * a[0]; R[1] = t * a[0] * a[1] - s * a[2]; R[2] = t * a[0] * a[2] + s * a[1];
    R[3] = t * a[0] * a[1] + s * a[2]; R[4] = c + t * a[1] * a[1]; R[5] = t * a[1] * a[2] - s * a[0];
    R[6] = t * a[0] * a[2] - s * a[1]; R[7] = t * a[1] * a[2] + s * a[0]; R[8] = c + t * a[2] * a[2];
}

/* spatial (6D) helpers, ordering [angular; linear] */
static void crm(const double v[6], const double m[6], double o[6]) { /* motion cross product v x m */
    double a[3], b[3], c[3];
    v3cross(v, m, a); v3cross(v, m + 3, b); v3cross(v + 3, m, c);
    for (int i = 0; i < 3; ++i) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
static void crf(const double v[6], const double f[6], double o[6]) { /* force cross product v x* f */
    double a[3], b[3], c[3];
    v3cross(v, f, a); v3cross(v + 3, f + 3, b); v3cross(v, f + 3, c);
    for (int i = 0; i < 3; ++i) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}
static void m6v(const double I[36], const double v[6], double o[6]) {
    for (int i = 0; i < 6; ++i) { double a = 0; for (int k = 0; k < 6; ++k) a += I[6 * i + k] * v[k]; o[i] = a; }
}
static double v6dot(const double a[6], const double b[6]) { double s = 0; for (int i = 0; i < 6; ++i) s += a[i] * b[i]; return s; }

/* forward kinematics relative to the base origin, world axes */
static void kinematics(const lsim_robot_model* m, const double Rb[9], const double q[12], xf_t X[NB], double axis_w[NB][3]) {
    memcpy(X[0].R, Rb, sizeof(double) * 9);
    X[0].p[0] = X[0].p[1] = X[0].p[2] = 0;
    axis_w[0][0] = axis_w[0][1] = axis_w[0][2] = 0;
    for (int i = 1; i < NB; ++i) {
        const lsim_body* b = &m->bodies[i];
        const xf_t* P = &X[b->parent];
        double jp[3] = {b->joint_pos[0], b->joint_pos[1], b->joint_pos[2]}, off[3];
        m3v(P->R, jp, off);
        for (int k = 0; k < 3; ++k) X[i].p[k] = P->p[k] + off[k];
        if (b->dof >= 0) {
            double ax[3] = {b->joint_axis[0], b->joint_axis[1], b->joint_axis[2]}, Rj[9];
            axis_angle_R(ax, q[b->dof], Rj);
            m3m(P->R, Rj, X[i].R);
            m3v(P->R, ax, axis_w[i]);
        } else {
            memcpy(X[i].R, P->R, sizeof(double) * 9);
            axis_w[i][0] = axis_w[i][1] = axis_w[i][2] = 0;
        }
    }
}

static void spatial_inertia(double mass, const double c[3], const double Ic[9], double I6[36]) {
    memset(I6, 0, sizeof(double) * 36);
    double cx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
    double cc = v3dot(c, c);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        I6[6 * i + j] = Ic[3 * i + j] + mass * ((i == j ? cc : 0.0) - c[i] * c[j]);
        I6[6 * i + 3 + j] = mass * cx[3 * i + j];
        I6[6 * (3 + i) + j] = mass * cx[3 * j + i];
        I6[6 * (3 + i) + 3 + j] = (i == j) ? mass : 0.0;
    }
}

/* ---- terrain contact -----------------------------------------------------------------------------------------------
 * The reference collides against the triangle mesh isaacgym.terrain_utils.convert_heightfield_to_trimesh builds from the
 * height grid with slope_treshold (TER:72-75): per grid cell two triangles, diagonal (i,j)-(i+1,j+1); where the height step
 * to a neighbour exceeds the threshold the LOWER vertex is moved one cell towards the higher one (vertical wall).
 * orc_vertex_moves restates that published vertex rule; the query is an independent (double precision, projection +
 * edge-distance) statement of "signed distance of a point to the displaced mesh". */
static void orc_vertex_move(const orc_sim* s, int i, int j, int* mx_out, int* my_out) {
    const lsim_config* c = &s->cfg;
    const int16_t* g = ORC_I16(s, LSIM_BUF_HEIGHT_GRID);
    const int R = c->grid_rows, C = c->grid_cols;
    *mx_out = *my_out = 0;
    if (c->mesh_type != 2 || c->slope_threshold <= 0.0f) return;
    const double thr = (double)c->slope_threshold * c->horizontal_scale / c->vertical_scale;
    double h = g[i * C + j];
    int mx = 0, my = 0, mc = 0;
    if (i + 1 < R && g[(i + 1) * C + j] - h > thr) mx += 1;
    if (i >= 1 && g[(i - 1) * C + j] - h > thr) mx -= 1;
    if (j + 1 < C && g[i * C + j + 1] - h > thr) my += 1;
    if (j >= 1 && g[i * C + j - 1] - h > thr) my -= 1;
    if (i + 1 < R && j + 1 < C && g[(i + 1) * C + j + 1] - h > thr) mc += 1;
    if (i >= 1 && j >= 1 && g[(i - 1) * C + j - 1] - h > thr) mc -= 1;
    *mx_out = mx + (mx == 0 ? mc : 0);
    *my_out = my + (my == 0 ? mc : 0);
}
void orc_build_mesh_cache(orc_sim* s) {
    const lsim_config* c = &s->cfg;
    if (c->mesh_type == 0 || c->grid_rows <= 0) return;
    const int R = c->grid_rows, C = c->grid_cols;
    s->vmove = (int8_t*)calloc((size_t)R * C, 2);
    s->cell_walls = (uint8_t*)calloc((size_t)R * C, 1);
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            int mx, my;
            orc_vertex_move(s, i, j, &mx, &my);
            s->vmove[2 * ((size_t)i * C + j)] = (int8_t)mx;
            s->vmove[2 * ((size_t)i * C + j) + 1] = (int8_t)my;
        }
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            int any = 0;
            for (int a = i - 1; a <= i + 2 && !any; ++a)
                for (int b = j - 1; b <= j + 2; ++b) {
                    if (a < 0 || b < 0 || a >= R || b >= C) continue;
                    if (s->vmove[2 * ((size_t)a * C + b)] || s->vmove[2 * ((size_t)a * C + b) + 1]) { any = 1; break; }
                }
            s->cell_walls[(size_t)i * C + j] = (uint8_t)any;
        }
}
static void mesh_vertex(const orc_sim* s, int a, int b, double v[3]) {
    const lsim_config* c = &s->cfg;
    int mx = s->vmove[2 * ((size_t)a * c->grid_cols + b)], my = s->vmove[2 * ((size_t)a * c->grid_cols + b) + 1];
    v[0] = (a + mx) * (double)c->horizontal_scale - c->border_size;
    v[1] = (b + my) * (double)c->horizontal_scale - c->border_size;
    v[2] = ORC_I16(s, LSIM_BUF_HEIGHT_GRID)[a * c->grid_cols + b] * (double)c->vertical_scale;
}
static double seg_closest(const double p[3], const double a[3], const double b[3], double q[3]) {
    double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ap[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
    double l2 = v3dot(ab, ab), t = l2 > 0 ? v3dot(ap, ab) / l2 : 0.0;
    if (t < 0) t = 0; if (t > 1) t = 1;
    for (int k = 0; k < 3; ++k) q[k] = a[k] + t * ab[k];
    double d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
    return v3dot(d, d);
}
/* squared distance and closest point of a triangle; returns 0 for collapsed triangles */
static int tri_closest(const double p[3], const double a[3], const double b[3], const double c[3], double q[3], double nrm[3], double* d2) {
    double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, n[3];
    v3cross(ab, ac, n);
    double nl2 = v3dot(n, n);
    if (nl2 < 1e-16) return 0;
    double nl = sqrt(nl2);
    for (int k = 0; k < 3; ++k) nrm[k] = n[k] / nl;
    double ap[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
    double dist = v3dot(ap, nrm), proj[3];
    for (int k = 0; k < 3; ++k) proj[k] = p[k] - dist * nrm[k];
    /* barycentric test of the projection */
    double v0[3] = {proj[0] - a[0], proj[1] - a[1], proj[2] - a[2]};
    double d00 = v3dot(ab, ab), d01 = v3dot(ab, ac), d11 = v3dot(ac, ac), d20 = v3dot(v0, ab), d21 = v3dot(v0, ac);
    double den = d00 * d11 - d01 * d01;
    double v = (d11 * d20 - d01 * d21) / den, w = (d00 * d21 - d01 * d20) / den;
    if (v >= 0 && w >= 0 && v + w <= 1) { memcpy(q, proj, sizeof(proj)); *d2 = dist * dist; return 1; }
    double q1[3], q2[3], q3[3];
    double e1 = seg_closest(p, a, b, q1), e2 = seg_closest(p, b, c, q2), e3 = seg_closest(p, c, a, q3);
    if (e1 <= e2 && e1 <= e3) { memcpy(q, q1, sizeof(q1)); *d2 = e1; }
    else if (e2 <= e3) { memcpy(q, q2, sizeof(q2)); *d2 = e2; }
    else { memcpy(q, q3, sizeof(q3)); *d2 = e3; }
    return 1;
}
static int cell_has_walls(const orc_sim* s, int i, int j) { return s->cell_walls[(size_t)i * s->cfg.grid_cols + j]; }
/* signed distance of world point cw to the terrain surface (negative inside) and contact normal */
void orc_terrain_contact(const orc_sim* s, const double cw[3], double radius, double* dist, double n[3]) {
    const lsim_config* c = &s->cfg;
    if (c->mesh_type == 0) { *dist = cw[2]; n[0] = n[1] = 0; n[2] = 1; return; }
    double hs = c->horizontal_scale, vs = c->vertical_scale;
    double gx = (cw[0] + c->border_size) / hs, gy = (cw[1] + c->border_size) / hs;
    double fi = floor(gx), fj = floor(gy);
    /* (written so that a NaN position -- a robot that blew up, counted in LSIM_BUF_NONFINITE -- lands in cell 0 instead of indexing with (int)NaN) */
    if (!(fi >= 0)) fi = 0; if (fi > c->grid_rows - 2) fi = c->grid_rows - 2;
    if (!(fj >= 0)) fj = 0; if (fj > c->grid_cols - 2) fj = c->grid_cols - 2;
    int i = (int)fi, j = (int)fj;
    const int16_t* g = ORC_I16(s, LSIM_BUF_HEIGHT_GRID);
    double h00 = g[i * c->grid_cols + j] * vs, h10 = g[(i + 1) * c->grid_cols + j] * vs;
    double h01 = g[i * c->grid_cols + j + 1] * vs, h11 = g[(i + 1) * c->grid_cols + j + 1] * vs;
    if (!cell_has_walls(s, i, j)) { /* plane of the grid triangle under the point */
        double u = gx - fi, v = gy - fj, h, dhx, dhy;
        if (u < 0) u = 0; if (u > 1) u = 1; if (v < 0) v = 0; if (v > 1) v = 1;
        if (u >= v) { dhx = h10 - h00; dhy = h11 - h10; h = h00 + u * dhx + v * dhy; }
        else { dhx = h11 - h01; dhy = h01 - h00; h = h00 + v * dhy + u * dhx; }
        double nx = -dhx / hs, ny = -dhy / hs, inv = 1.0 / sqrt(nx * nx + ny * ny + 1.0);
        n[0] = nx * inv; n[1] = ny * inv; n[2] = inv;
        *dist = (cw[2] - h) * inv;
        return;
    }
    /* only features within reach = radius + contact_offset can make a contact: cells whose bounding box grown by reach (x, y, +z)
     * excludes the centre are skipped, and triangles whose plane lies more than reach below the centre */
    const double reach = radius + (double)c->contact_offset;
    double best = 1e30, bq[3] = {0, 0, 0}, bn[3] = {0, 0, 1};
    for (int ci = i - 1; ci <= i + 1; ++ci)
        for (int cj = j - 1; cj <= j + 1; ++cj) {
            if (ci < 0 || cj < 0 || ci > c->grid_rows - 2 || cj > c->grid_cols - 2) continue;
            double p00[3], p10[3], p01[3], p11[3], q[3], nt[3], d2;
            mesh_vertex(s, ci, cj, p00); mesh_vertex(s, ci + 1, cj, p10); mesh_vertex(s, ci, cj + 1, p01); mesh_vertex(s, ci + 1, cj + 1, p11);
            int out = cw[2] > fmax(fmax(p00[2], p10[2]), fmax(p01[2], p11[2])) + reach;
            for (int k = 0; k < 2; ++k) {
                double lo = fmin(fmin(p00[k], p10[k]), fmin(p01[k], p11[k])), hi = fmax(fmax(p00[k], p10[k]), fmax(p01[k], p11[k]));
                if (cw[k] < lo - reach || cw[k] > hi + reach) out = 1;
            }
            if (out) continue;
            const double* tri[2][2] = {{p11, p01}, {p10, p11}};
            for (int t = 0; t < 2; ++t) {
                if (!tri_closest(cw, p00, tri[t][0], tri[t][1], q, nt, &d2)) continue;
                double ap[3] = {cw[0] - p00[0], cw[1] - p00[1], cw[2] - p00[2]};
                if (v3dot(nt, ap) > reach) continue;
                if (d2 < best) { best = d2; memcpy(bq, q, sizeof(q)); memcpy(bn, nt, sizeof(nt)); }
            }
        }
    if (best > 1e29) { *dist = 1.0 + radius; n[0] = n[1] = 0; n[2] = 1; return; }
    double d = sqrt(best), dq[3] = {cw[0] - bq[0], cw[1] - bq[1], cw[2] - bq[2]};
    double side = v3dot(bn, dq);
    if (side < -1e-6) { *dist = -d; memcpy(n, bn, sizeof(bn)); }
    else if (d > 1e-6) { *dist = d; for (int k = 0; k < 3; ++k) n[k] = dq[k] / d; }
    else { *dist = 0; memcpy(n, bn, sizeof(bn)); }
}

static void tangent_basis(const double n[3], double t1[3], double t2[3]) {
    double ref[3] = {1, 0, 0};
    if (fabs(n[0]) > 0.9) { ref[0] = 0; ref[1] = 1; }
    v3cross(n, ref, t1);
    double l = sqrt(v3dot(t1, t1));
    for (int k = 0; k < 3; ++k) t1[k] /= l;
    v3cross(n, t1, t2);
}

static void base_mass_props(const orc_sim* s, int e, double* mass, double com[3], double I[6]) {
    const lsim_body* b = &s->model.bodies[0];
    double pay = ORC_F(s, LSIM_BUF_PAYLOAD)[e];
    *mass = b->mass + pay;
    for (int k = 0; k < 3; ++k) com[k] = b->com[k] + ORC_F(s, LSIM_BUF_COM_DISPLACEMENT)[3 * e + k];
    double sc = *mass / b->mass;
    for (int k = 0; k < 6; ++k) I[k] = b->inertia[k] * sc;
}

/* lin_vel_at_com: the root state tensor carries the linear velocity of the base's centre of mass; the sub-steps work on the link
   origin's.  dir = -1: v_origin = v_com - w x (R c) (before the first sub-step); the way back is row 0 of orc_refresh_body_states. */
void orc_root_lin_vel_to_origin(orc_sim* s, int e) {
    float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
    double qb[4] = {root[3], root[4], root[5], root[6]}, Rb[9], cl[3], cw[3], w[3] = {root[10], root[11], root[12]}, wxr[3];
    double qn = sqrt(qb[0] * qb[0] + qb[1] * qb[1] + qb[2] * qb[2] + qb[3] * qb[3]);
    for (int k = 0; k < 4; ++k) qb[k] /= qn;
    quat_to_R(qb, Rb);
    for (int k = 0; k < 3; ++k) cl[k] = s->model.bodies[0].com[k] + ORC_F(s, LSIM_BUF_COM_DISPLACEMENT)[3 * e + k];
    m3v(Rb, cl, cw);
    v3cross(w, cw, wxr);
    for (int k = 0; k < 3; ++k) root[7 + k] = (float)(root[7 + k] - wxr[k]);
}

void orc_refresh_body_states(orc_sim* s, int e) {
    const lsim_robot_model* m = &s->model;
    const float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
    const float* dof = ORC_F(s, LSIM_BUF_DOF_STATE) + 24 * e;
    float* out = ORC_F(s, LSIM_BUF_RIGID_BODY_STATES) + 13 * NB * e;
    double qb[4] = {root[3], root[4], root[5], root[6]}, Rb[9], q[12], qd[12];
    quat_to_R(qb, Rb);
    for (int j = 0; j < 12; ++j) { q[j] = dof[2 * j]; qd[j] = dof[2 * j + 1]; }
    xf_t X[NB]; double aw[NB][3], V[NB][6];
    kinematics(m, Rb, q, X, aw);
    for (int k = 0; k < 3; ++k) { V[0][k] = root[10 + k]; V[0][3 + k] = root[7 + k]; }
    for (int i = 1; i < NB; ++i) {
        const lsim_body* b = &m->bodies[i];
        memcpy(V[i], V[b->parent], sizeof(double) * 6);
        if (b->dof >= 0) {
            double lin[3]; v3cross(X[i].p, aw[i], lin);
            for (int k = 0; k < 3; ++k) { V[i][k] += aw[i][k] * qd[b->dof]; V[i][3 + k] += lin[k] * qd[b->dof]; }
        }
    }
    for (int i = 0; i < NB; ++i) {
        double qq[4], vel[3], wxp[3], at[3];
        R_to_quat(X[i].R, qq);
        if (i == 0) memcpy(qq, qb, sizeof(qq));
        for (int k = 0; k < 3; ++k) at[k] = X[i].p[k];
        if (s->cfg.lin_vel_at_com) { /* include/lsim.h lin_vel_at_com: the linear velocity of the body's centre of mass (PhysX getLinearVelocity) */
            double cl[3], cw[3];
            for (int k = 0; k < 3; ++k) cl[k] = m->bodies[i].com[k] + (i == 0 ? ORC_F(s, LSIM_BUF_COM_DISPLACEMENT)[3 * e + k] : 0.0);
            m3v(X[i].R, cl, cw);
            for (int k = 0; k < 3; ++k) at[k] += cw[k];
        }
        v3cross(V[i], at, wxp);
        for (int k = 0; k < 3; ++k) vel[k] = V[i][3 + k] + wxp[k];
        float* o = out + 13 * i;
        for (int k = 0; k < 3; ++k) o[k] = (float)(root[k] + X[i].p[k]);
        for (int k = 0; k < 4; ++k) o[3 + k] = (float)qq[k];
        for (int k = 0; k < 3; ++k) { o[7 + k] = (float)vel[k]; o[10 + k] = (float)V[i][k]; }
    }
}

static int cholesky(double* A, int n) { /* in place lower factor, row-major n x n */
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (d <= 0) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = A[i * n + j];
            for (int k = 0; k < j; ++k) v -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = v / d;
        }
    }
    return 0;
}
static void chol_solve(const double* L, int n, double* b) {
    for (int i = 0; i < n; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[i * n + k] * b[k]; b[i] = v / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double v = b[i]; for (int k = i + 1; k < n; ++k) v -= L[k * n + i] * b[k]; b[i] = v / L[i * n + i]; }
}

void orc_physics_substep(orc_sim* s, int e, const float tau_f[12], int apply_force) {
    const lsim_config* c = &s->cfg;
    const lsim_robot_model* m = &s->model;
    const double dt = c->sim_dt;
    float* root = ORC_F(s, LSIM_BUF_ROOT_STATES) + 13 * e;
    float* dof = ORC_F(s, LSIM_BUF_DOF_STATE) + 24 * e;
    float* cfo = ORC_F(s, LSIM_BUF_CONTACT_FORCES) + 3 * NB * e;
    double p0[3] = {root[0], root[1], root[2]}, qb[4] = {root[3], root[4], root[5], root[6]};
    double qn = sqrt(qb[0] * qb[0] + qb[1] * qb[1] + qb[2] * qb[2] + qb[3] * qb[3]);
    for (int k = 0; k < 4; ++k) qb[k] /= qn;
    double q[12], v[NV];
    for (int j = 0; j < 12; ++j) { q[j] = dof[2 * j]; v[6 + j] = dof[2 * j + 1]; }
    for (int k = 0; k < 3; ++k) { v[k] = root[10 + k]; v[3 + k] = root[7 + k]; }

    /* 1. kinematics, motion subspaces */
    double Rb[9]; quat_to_R(qb, Rb);
    xf_t X[NB]; double aw[NB][3], S[NB][6];
    kinematics(m, Rb, q, X, aw);
    for (int i = 0; i < NB; ++i) {
        memset(S[i], 0, sizeof(S[i]));
        if (m->bodies[i].dof >= 0) { for (int k = 0; k < 3; ++k) S[i][k] = aw[i][k]; v3cross(X[i].p, aw[i], S[i] + 3); }
    }
    /* 2. spatial inertias about O */
    double I6[NB][36], com_w[NB][3];
    for (int i = 0; i < NB; ++i) {
        const lsim_body* b = &m->bodies[i];
        double mass = b->mass, cl[3] = {b->com[0], b->com[1], b->com[2]}, Il[6];
        for (int k = 0; k < 6; ++k) Il[k] = b->inertia[k];
        if (i == 0) base_mass_props(s, e, &mass, cl, Il);
        double cw[3]; m3v(X[i].R, cl, cw);
        for (int k = 0; k < 3; ++k) com_w[i][k] = cw[k] + X[i].p[k];
        double Ilm[9] = {Il[0], Il[1], Il[2], Il[1], Il[3], Il[4], Il[2], Il[4], Il[5]}, T[9], Rt[9], Iw[9];
        for (int a = 0; a < 3; ++a) for (int bb = 0; bb < 3; ++bb) Rt[3 * a + bb] = X[i].R[3 * bb + a];
        m3m(X[i].R, Ilm, T); m3m(T, Rt, Iw);
        spatial_inertia(mass, com_w[i], Iw, I6[i]);
    }
    /* 3. velocities, bias accelerations, bias forces (RNEA with zero joint acceleration) */
    double V[NB][6], Ab[NB][6], F[NB][6];
    memcpy(V[0], v, sizeof(double) * 6);
    memset(Ab[0], 0, sizeof(Ab[0]));
    const double ag[6] = {0, 0, 0, c->gravity[0], c->gravity[1], c->gravity[2]};
    for (int i = 1; i < NB; ++i) {
        const lsim_body* b = &m->bodies[i];
        memcpy(V[i], V[b->parent], sizeof(V[i]));
        memcpy(Ab[i], Ab[b->parent], sizeof(Ab[i]));
        if (b->dof >= 0) {
            double vj[6], cc[6];
            for (int k = 0; k < 6; ++k) { vj[k] = S[i][k] * v[6 + b->dof]; V[i][k] += vj[k]; }
            crm(V[i], vj, cc);
            for (int k = 0; k < 6; ++k) Ab[i][k] += cc[k];
        }
    }
    for (int i = 0; i < NB; ++i) {
        double a[6], Ia[6], Iv[6], vIv[6];
        for (int k = 0; k < 6; ++k) a[k] = Ab[i][k] - ag[k];
        m6v(I6[i], a, Ia); m6v(I6[i], V[i], Iv); crf(V[i], Iv, vIv);
        for (int k = 0; k < 6; ++k) F[i][k] = Ia[k] + vIv[k];
    }
    /* external: body-local disturbance force at the base COM (LR:844), first sub-step only */
    if (apply_force) {
        float* pf = ORC_F(s, LSIM_BUF_PENDING_FORCE) + 3 * e;
        double fl[3] = {pf[0], pf[1], pf[2]}, fw[3], mo[3];
        m3v(Rb, fl, fw); v3cross(com_w[0], fw, mo);
        for (int k = 0; k < 3; ++k) { F[0][k] -= mo[k]; F[0][3 + k] -= fw[k]; }
        pf[0] = pf[1] = pf[2] = 0.0f;
    }
    for (int i = NB - 1; i >= 1; --i) for (int k = 0; k < 6; ++k) F[m->bodies[i].parent][k] += F[i][k];
    double h[NV];
    for (int k = 0; k < 6; ++k) h[k] = F[0][k];
    for (int i = 1; i < NB; ++i) if (m->bodies[i].dof >= 0) h[6 + m->bodies[i].dof] = v6dot(S[i], F[i]);
    /* 4. mass matrix (CRBA) */
    double Ic[NB][36];
    memcpy(Ic, I6, sizeof(Ic));
    for (int i = NB - 1; i >= 1; --i) for (int k = 0; k < 36; ++k) Ic[m->bodies[i].parent][k] += Ic[i][k];
    double M[NV * NV];
    memset(M, 0, sizeof(M));
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) M[a * NV + b] = Ic[0][6 * a + b];
    for (int i = 1; i < NB; ++i) {
        int d = m->bodies[i].dof;
        if (d < 0) continue;
        double Fi[6]; m6v(Ic[i], S[i], Fi);
        M[(6 + d) * NV + 6 + d] = v6dot(S[i], Fi);
        for (int k = 0; k < 6; ++k) { M[k * NV + 6 + d] = Fi[k]; M[(6 + d) * NV + k] = Fi[k]; }
        int j = m->bodies[i].parent;
        while (j > 0) {
            int dj = m->bodies[j].dof;
            if (dj >= 0) { double x = v6dot(S[j], Fi); M[(6 + dj) * NV + 6 + d] = x; M[(6 + d) * NV + 6 + dj] = x; }
            j = m->bodies[j].parent;
        }
    }
    double L[NV * NV];
    memcpy(L, M, sizeof(M));
    if (cholesky(L, NV) != 0) return; /* singular model: leave state untouched */
    /* 5. free velocity */
    double rhs[NV], vfree[NV];
    for (int k = 0; k < 6; ++k) rhs[k] = -h[k];
    for (int j = 0; j < 12; ++j) rhs[6 + j] = (double)tau_f[j] - h[6 + j];
    chol_solve(L, NV, rhs);
    for (int k = 0; k < NV; ++k) vfree[k] = v[k] + dt * rhs[k];

    /* 6. collision detection: sphere-swept points vs terrain */
    int nc = 0, nact = 0, cbody[MAXC];
    double cpos[MAXC][3], cn[MAXC][3], cdist[MAXC];
    for (int pidx = 0; pidx < m->num_collision_points; ++pidx) {
        const lsim_collision_point* cp = &m->points[pidx];
        double pl[3] = {cp->pos[0], cp->pos[1], cp->pos[2]}, pw[3];
        m3v(X[cp->body].R, pl, pw);
        for (int k = 0; k < 3; ++k) pw[k] += X[cp->body].p[k];
        double dsurf, n[3], cw[3] = {p0[0] + pw[0], p0[1] + pw[1], p0[2] + pw[2]};
        orc_terrain_contact(s, cw, cp->radius, &dsurf, n);
        double dist = dsurf - cp->radius;
        if (dist < c->contact_offset) {
            ++nact;
            if (nc >= MAXC) continue;            /* the cap: points are ordered by priority, overflow is dropped from the end */
            cbody[nc] = cp->body; cdist[nc] = dist;
            for (int k = 0; k < 3; ++k) { cn[nc][k] = n[k]; cpos[nc][k] = pw[k] - n[k] * cp->radius; }
            ++nc;
        }
    }
    {   /* diagnostic: contacts before the cap (LSIM_BUF_CONTACT_COUNT) */
        int32_t* cc = ORC_I32(s, LSIM_BUF_CONTACT_COUNT) + 2 * e;
        if (nact > cc[0]) cc[0] = nact;
        cc[1] = nact;
    }
    /* 7. constraint rows */
    int R = 0, rkind[MAXR], rcontact[MAXR];
    double J[MAXR][NV], vt[MAXR], dirs[MAXR][3];
    double mu = 0.5 * ((double)c->terrain_friction + (double)ORC_F(s, LSIM_BUF_FRICTION)[e]);
    for (int k = 0; k < nc; ++k) {
        double t1[3], t2[3];
        tangent_basis(cn[k], t1, t2);
        const double* dd[3] = {cn[k], t1, t2};
        for (int a = 0; a < 3; ++a) {
            double f[6];
            v3cross(cpos[k], dd[a], f);
            for (int x = 0; x < 3; ++x) { f[3 + x] = dd[a][x]; dirs[R][x] = dd[a][x]; }
            memset(J[R], 0, sizeof(J[R]));
            for (int x = 0; x < 6; ++x) J[R][x] = f[x];
            for (int j = cbody[k]; j > 0; j = m->bodies[j].parent)
                if (m->bodies[j].dof >= 0) J[R][6 + m->bodies[j].dof] = v6dot(S[j], f);
            rkind[R] = a; rcontact[R] = k;
            if (a == 0) {
                double d = cdist[k];
                if (d >= 0) vt[R] = -d / dt;
                else { double pen = -d - c->contact_slop; if (pen < 0) pen = 0; vt[R] = fmin((double)c->max_depenetration_velocity, c->erp * pen / dt); }
            } else vt[R] = 0;
            ++R;
        }
    }
    static const double LIMIT_MARGIN = 0.2;   /* same constant as LS_LIMIT_MARGIN of the kernels */
    double lim_L[12], lim_U[12], rrng[MAXR];
    int lim_need[12], lim_viol[12];
    for (int j = 0; j < 12; ++j) {
        /* joint position AND velocity limits as ONE row per joint: the admissible velocity interval is
         *   [L, U] = [-vmax, vmax]  intersected with  v >= -gap_lo/dt (near the lower stop)  /  v <= gap_hi/dt (near the upper stop),
         * penetrated stops push back with erp, capped at 1 rad/s.  Clamping the joint velocity after the solve
         * instead (as this code once did) leaves the reaction of the saturated motor torque on the base and pumps angular momentum into a
         * robot in free flight. */
        double lo = q[j] - m->dof_pos_lower[j], hi = m->dof_pos_upper[j] - q[j], vmax = m->dof_vel_limit[j], vf = vfree[6 + j];
        double Lb = -vmax, Ub = vmax;
        if (lo < 0.1) Lb = fmax(Lb, lo >= 0 ? -lo / dt : fmin(1.0, c->erp * (-lo) / dt));
        if (hi < 0.1) Ub = fmin(Ub, hi >= 0 ? hi / dt : -fmin(1.0, c->erp * (-hi) / dt));
        if (Ub < Lb) Ub = Lb;
        lim_L[j] = Lb; lim_U[j] = Ub;
        lim_need[j] = fmin(vf - Lb, Ub - vf) < LIMIT_MARGIN * vmax;
        lim_viol[j] = fmin(vf - Lb, Ub - vf) < 0.0;
    }
    for (int j = 11; j >= 0; --j) {
        /* Rows in DESCENDING joint order -- calf, thigh, hip of the last leg first -- i.e. leaf to root within a leg: a Gauss-Seidel pass in
         * that order leaves a third of the residue of the ascending order on a saturated leg (round 4: with TGS's single pass per
         * iteration, joint speed beyond 1.1 x the limit in 2.7 % instead of 19 % of the steps of tests/test_physics_invariants.py's
         * saturated-motor case).
         * a two-sided row L <= qd_j <= U for every joint whose free velocity violates a bound or comes within LIMIT_MARGIN of the velocity
         * limit of it, and for the neighbours on the leg of a joint that violates a bound (limit impulses of one joint move its neighbours
         * by tens of rad/s).  Until round 2 a joint merely within the margin pulled its neighbours in too: 4.6 of 5.3 limit rows never
         * carried an impulse. */
        int leg = j / 3;
        if (!(lim_need[j] || lim_viol[3 * leg] || lim_viol[3 * leg + 1] || lim_viol[3 * leg + 2])) continue;
        double sgn = 1;
        memset(J[R], 0, sizeof(J[R]));
        J[R][6 + j] = sgn;
        vt[R] = lim_L[j];
        rrng[R] = lim_U[j] - lim_L[j];
        rkind[R] = 3; rcontact[R] = -1;
        dirs[R][0] = dirs[R][1] = dirs[R][2] = 0;
        ++R;
    }
    if (c->solver_type == LSIM_SOLVER_TGS) {
    /* ---- solver_type 1: PhysX's Temporal Gauss-Seidel scheme, the solver every reference config selects (solver_type = 1,
     * num_position_iterations = 4, num_velocity_iterations = 0: LRC:245-248), restated from its published description
     * (Macklin, Storey, Lu, Terdiman, Chentanez, Jeschke, Mueller: "Small Steps in Physics Simulation", SCA 2019; PhysX SDK guide, "Temporal
     * Gauss-Seidel"): the step is split into N = position-iterations sub-iterations of h = dt / N; each integrates the unconstrained
     * acceleration over h, relaxes every constraint ONCE against the positional error of the configuration reached so far (gaps and joint
     * angles advance with the sub-iterations; the articulation's response -- J, M^-1 J^T -- stays that of the start of the step), and advances
     * the configuration by h.  Impulses accumulate over the step and are projected as totals (normal >= 0, friction box mu * normal).  No
     * velocity iterations.  External forces (gravity, bias forces, motor torques) act ONCE, over the whole dt, before the iterations -- the
     * iterations start from vfree, as the SDK computes its unconstrained velocities with the full step -- so an unconstrained robot takes
     * exactly the semi-implicit Euler step of the PGS branch.  [Round 3's first restatement ramped the free acceleration over the
     * sub-iterations, the paper's sub-stepping: positions then advance by dt v + 5/8 dt^2 a, which is not symplectic -- the free-flight
     * energy test drifted 19 % instead of 6.6 % -- and one relaxation per iteration of an error that GROWS with the iterations left a
     * larger velocity residue on stiff coupled rows.]  PhysX itself stays closed: this is parity-UNPINNED like the rest of the dynamics (header of this file); the
     * HIP kernel's TGS form (ls_physics.h: wc_delassus_tgs) is checked against this one. */
        const int NS = c->num_position_iterations < 1 ? 1 : c->num_position_iterations;
        const double hs = dt / NS;
        double Y[MAXR][NV], Wd[MAXR], lam[MAXR], vv[NV], gap[MAXC], qs[12], dp[3] = {0, 0, 0}, qcur[4] = {qb[0], qb[1], qb[2], qb[3]};
        for (int r = 0; r < R; ++r) {
            memcpy(Y[r], J[r], sizeof(Y[r])); chol_solve(L, NV, Y[r]);
            double a = 1e-6; for (int k = 0; k < NV; ++k) a += J[r][k] * Y[r][k];
            Wd[r] = a; lam[r] = 0;
        }
        memcpy(vv, vfree, sizeof(vv));    /* external forces act once, over the whole dt, before the iterations (see the comment above) */
        for (int k = 0; k < nc; ++k) gap[k] = cdist[k];
        for (int j = 0; j < 12; ++j) qs[j] = q[j];
        for (int sub = 0; sub < NS; ++sub) {
            for (int r = 0; r < R; ++r) {
                double tgt = 0, rng = 0;
                if (rkind[r] == 0) {
                    double d = gap[rcontact[r]];
                    if (d >= 0) tgt = -d / hs;
                    else { double pen = -d - c->contact_slop; if (pen < 0) pen = 0; tgt = fmin((double)c->max_depenetration_velocity, c->erp * pen / hs); }
                } else if (rkind[r] == 3) {
                    int j = -1; for (int k = 0; k < 12; ++k) if (J[r][6 + k] != 0) j = k;
                    double lo = qs[j] - m->dof_pos_lower[j], hi = m->dof_pos_upper[j] - qs[j], vmax = m->dof_vel_limit[j];
                    double Lb = -vmax, Ub = vmax;
                    if (lo < 0.1) Lb = fmax(Lb, lo >= 0 ? -lo / hs : fmin(1.0, c->erp * (-lo) / hs));
                    if (hi < 0.1) Ub = fmin(Ub, hi >= 0 ? hi / hs : -fmin(1.0, c->erp * (-hi) / hs));
                    if (Ub < Lb) Ub = Lb;
                    tgt = Lb; rng = Ub - Lb;
                }
                double w = -tgt; for (int k = 0; k < NV; ++k) w += J[r][k] * vv[k];
                double nl = lam[r] - w / Wd[r];
                if (rkind[r] == 0) { if (nl < 0) nl = 0; }
                else if (rkind[r] == 3) { double up = nl + rng / Wd[r]; nl = (nl > 0 ? nl : 0) + (up < 0 ? up : 0); }   /* two-sided, as the PGS sweep */
                else { double lim = mu * lam[r - rkind[r]]; if (nl > lim) nl = lim; if (nl < -lim) nl = -lim; }
                double dl = nl - lam[r];
                lam[r] = nl;
                for (int k = 0; k < NV; ++k) vv[k] += Y[r][k] * dl;
            }
            /* advance the configuration by h: contact gaps along their normals, joints, base pose */
            for (int r = 0; r < R; ++r) if (rkind[r] == 0) { double a = 0; for (int k = 0; k < NV; ++k) a += J[r][k] * vv[k]; gap[rcontact[r]] += hs * a; }
            for (int j = 0; j < 12; ++j) qs[j] += hs * vv[6 + j];
            {   /* vv[3..5] is the velocity of the point of the base that sat at the base origin at the start of the step */
                double wxd[3], w3[3] = {vv[0], vv[1], vv[2]};
                v3cross(w3, dp, wxd);
                for (int k = 0; k < 3; ++k) dp[k] += hs * (vv[3 + k] + wxd[k]);
                double dq[4] = { 0.5 * hs * ( w3[0] * qcur[3] + w3[1] * qcur[2] - w3[2] * qcur[1]),
                                 0.5 * hs * (-w3[0] * qcur[2] + w3[1] * qcur[3] + w3[2] * qcur[0]),
                                 0.5 * hs * ( w3[0] * qcur[1] - w3[1] * qcur[0] + w3[2] * qcur[3]),
                                 0.5 * hs * (-w3[0] * qcur[0] - w3[1] * qcur[1] - w3[2] * qcur[2]) };
                for (int k = 0; k < 4; ++k) qcur[k] += dq[k];     /* linear in q: normalised once, after the last sub-iteration */
            }
        }
        { double nn = 0; for (int k = 0; k < 4; ++k) nn += qcur[k] * qcur[k]; nn = sqrt(nn); for (int k = 0; k < 4; ++k) qcur[k] /= nn; }
        {   /* lsim_config.tgs_limit_passes: velocity-level Gauss-Seidel passes over the limit rows alone (contact impulses frozen), against
             * the bounds of the configuration reached: only the velocity the step hands on changes */
            for (int it = 0; it < c->tgs_limit_passes; ++it)
                for (int r = 0; r < R; ++r) {
                    if (rkind[r] != 3) continue;
                    int j = -1; for (int k = 0; k < 12; ++k) if (J[r][6 + k] != 0) j = k;
                    double lo = qs[j] - m->dof_pos_lower[j], hi = m->dof_pos_upper[j] - qs[j], vmax = m->dof_vel_limit[j];
                    double Lb = -vmax, Ub = vmax;
                    if (lo < 0.1) Lb = fmax(Lb, lo >= 0 ? -lo / hs : fmin(1.0, c->erp * (-lo) / hs));
                    if (hi < 0.1) Ub = fmin(Ub, hi >= 0 ? hi / hs : -fmin(1.0, c->erp * (-hi) / hs));
                    if (Ub < Lb) Ub = Lb;
                    double w = -Lb; for (int k = 0; k < NV; ++k) w += J[r][k] * vv[k];
                    double nl = lam[r] - w / Wd[r];
                    double up = nl + (Ub - Lb) / Wd[r]; nl = (nl > 0 ? nl : 0) + (up < 0 ? up : 0);
                    double dl = nl - lam[r];
                    lam[r] = nl;
                    for (int k = 0; k < NV; ++k) vv[k] += Y[r][k] * dl;
                }
        }
        double cf[NB][3];
        memset(cf, 0, sizeof(cf));
        for (int r = 0; r < R; ++r) if (rcontact[r] >= 0) for (int k = 0; k < 3; ++k) cf[cbody[rcontact[r]]][k] += lam[r] * dirs[r][k] / dt;
        for (int i = 0; i < NB; ++i) for (int k = 0; k < 3; ++k) cfo[3 * i + k] = (float)cf[i][k];
        {   /* the same safety nets as the PGS branch, on the velocities the step hands on (the configuration has already advanced with the
             * sub-iterations): body velocity caps of the asset options, joint speed at 1.5 x the limit for solver residue */
            double wn = sqrt(vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2]), ln = sqrt(vv[3] * vv[3] + vv[4] * vv[4] + vv[5] * vv[5]);
            if (c->max_angular_velocity > 0 && wn > c->max_angular_velocity) for (int k = 0; k < 3; ++k) vv[k] *= c->max_angular_velocity / wn;
            if (c->max_linear_velocity > 0 && ln > c->max_linear_velocity) for (int k = 3; k < 6; ++k) vv[k] *= c->max_linear_velocity / ln;
        }
        for (int j = 0; j < 12; ++j) {
            double lim = 1.5 * m->dof_vel_limit[j], vj = vv[6 + j];
            if (vj > lim) vj = lim;
            if (vj < -lim) vj = -lim;
            dof[2 * j] = (float)qs[j]; dof[2 * j + 1] = (float)vj;
        }
        double wxd[3], w3[3] = {vv[0], vv[1], vv[2]};
        v3cross(w3, dp, wxd);
        for (int k = 0; k < 3; ++k) { root[k] = (float)(p0[k] + dp[k]); root[7 + k] = (float)(vv[3 + k] + wxd[k]); root[10 + k] = (float)vv[k]; }
        for (int k = 0; k < 4; ++k) root[3 + k] = (float)qcur[k];
        return;
    }
    /* 8. Delassus operator and projected Gauss-Seidel */
    static const double CFM = 1e-6;
    double Y[MAXR][NV], W[MAXR][MAXR], b[MAXR], lam[MAXR];
    for (int r = 0; r < R; ++r) { memcpy(Y[r], J[r], sizeof(Y[r])); chol_solve(L, NV, Y[r]); }
    for (int r = 0; r < R; ++r) {
        for (int q2 = 0; q2 < R; ++q2) { double a = 0; for (int k = 0; k < NV; ++k) a += J[r][k] * Y[q2][k]; W[r][q2] = a; }
        W[r][r] += CFM;
        double a = 0; for (int k = 0; k < NV; ++k) a += J[r][k] * vfree[k];
        b[r] = a - vt[r];
        lam[r] = 0;
    }
    for (int it = 0; it < c->solver_iterations; ++it)
        for (int r = 0; r < R; ++r) {
            double w = b[r];
            for (int q2 = 0; q2 < R; ++q2) w += W[r][q2] * lam[q2];
            double nl = lam[r] - w / W[r][r];
            if (rkind[r] == 0) { if (nl < 0) nl = 0; }
            else if (rkind[r] == 3) { double up = nl + rrng[r] / W[r][r]; nl = (nl > 0 ? nl : 0) + (up < 0 ? up : 0); }   /* two-sided */
            else { double lim = mu * lam[r - rkind[r]]; if (nl > lim) nl = lim; if (nl < -lim) nl = -lim; }
            lam[r] = nl;
        }
    double vn[NV];
    memcpy(vn, vfree, sizeof(vn));
    for (int r = 0; r < R; ++r) for (int k = 0; k < NV; ++k) vn[k] += Y[r][k] * lam[r];
    /* 9. contact force report: net force per body, world frame (LR:944) */
    double cf[NB][3];
    memset(cf, 0, sizeof(cf));
    for (int r = 0; r < R; ++r) if (rcontact[r] >= 0) for (int k = 0; k < 3; ++k) cf[cbody[rcontact[r]]][k] += lam[r] * dirs[r][k] / dt;
    for (int i = 0; i < NB; ++i) for (int k = 0; k < 3; ++k) cfo[3 * i + k] = (float)cf[i][k];
    /* 10. integrate */
    {   /* body velocity caps of the asset options (LRC:229-230; PhysX clamps there too): a safety net, never reached by a sane robot */
        double wn = sqrt(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]), ln = sqrt(vn[3] * vn[3] + vn[4] * vn[4] + vn[5] * vn[5]);
        if (c->max_angular_velocity > 0 && wn > c->max_angular_velocity) for (int k = 0; k < 3; ++k) vn[k] *= c->max_angular_velocity / wn;
        if (c->max_linear_velocity > 0 && ln > c->max_linear_velocity) for (int k = 3; k < 6; ++k) vn[k] *= c->max_linear_velocity / ln;
    }
    for (int j = 0; j < 12; ++j) {
        double lim = 1.5 * m->dof_vel_limit[j];   /* the limit itself is a constraint row (step 7); this only bounds solver residue */
        if (vn[6 + j] > lim) vn[6 + j] = lim;
        if (vn[6 + j] < -lim) vn[6 + j] = -lim;
        q[j] += dt * vn[6 + j];
        dof[2 * j] = (float)q[j]; dof[2 * j + 1] = (float)vn[6 + j];
    }
    double dp[3] = {dt * vn[3], dt * vn[4], dt * vn[5]}, wxd[3];
    v3cross(vn, dp, wxd);
    double w[3] = {vn[0], vn[1], vn[2]};
    double dq[4] = { 0.5 * dt * ( w[0] * qb[3] + w[1] * qb[2] - w[2] * qb[1]),
                     0.5 * dt * (-w[0] * qb[2] + w[1] * qb[3] + w[2] * qb[0]),
                     0.5 * dt * ( w[0] * qb[1] - w[1] * qb[0] + w[2] * qb[3]),
                     0.5 * dt * (-w[0] * qb[0] - w[1] * qb[1] - w[2] * qb[2]) };
    double qq[4] = {qb[0] + dq[0], qb[1] + dq[1], qb[2] + dq[2], qb[3] + dq[3]};
    double nn = sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
    for (int k = 0; k < 3; ++k) {
        root[k] = (float)(p0[k] + dp[k]);
        root[7 + k] = (float)(vn[3 + k] + wxd[k]);
        root[10 + k] = (float)vn[k];
    }
    for (int k = 0; k < 4; ++k) root[3 + k] = (float)(qq[k] / nn);
}
