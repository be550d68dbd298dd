/* kp_oracle.c -- CPU fp64 restatement of the KinPoly rollout physics path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (kinpoly_amd/, libkinpoly_sim.so) may
 * include, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / reported CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" at the MuJoCo boundary.  The reference executes this
 * arithmetic inside MuJoCo 2.1.0 (binary dependency `mujoco-py<2.2,>=2.1`, not present in
 * /root/reference, cannot be run here).  Functions marked [MJ-ext] restate MuJoCo's published
 * algorithm (MuJoCo "Computation" chapter + engine sources as recalled); they are anchored on
 * the reference's call sites, not on golden vectors.  Functions marked [REF] restate Python in
 * the reference tree and ARE pinned by tests/golden fixtures generated from the reference.
 *
 * Reference call sites restated here:
 *   sim.forward()/sim.step()            uhc/envs/humanoid_im.py:527, mujoco_env.py:99-103   [MJ-ext]
 *   mj_fullM / data.qfrc_bias           uhc/envs/humanoid_im.py:422-426                     [MJ-ext]
 *   compute_desired_accel               uhc/envs/humanoid_im.py:418-431                     [REF]
 *   compute_torque                      uhc/envs/humanoid_im.py:433-480                     [REF]
 *   rfc_implicit                        uhc/envs/humanoid_im.py:497-504                     [REF]
 *   do_simulation                       uhc/envs/humanoid_im.py:506-533                     [REF]
 *
 * Plain C99, scalar, one env per call.  Build: see oracle/Makefile.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NB_MAX 24
#define NV_MAX 75
#define NQ_MAX 76
#define NM_MAX 1221
#define MAXCON 64
#define MAXEFC (MAXCON * 4 + 2 * 69)
#define MAXGEOM 8
#define MAXOBJ 2                        /* dynamic free objects per env (push: box + table) */
#define NVT_MAX (NV_MAX + 6 * MAXOBJ)   /* dofs of the humanoid + the active objects */
#define MJ_MINVAL 1e-15
#define MJ_MINIMP 0.0001
#define MJ_MAXIMP 0.9999

typedef struct {
    int nb, nv, nq, nu, nM, nvert;
    int body_parent[NB_MAX], body_subtree[NB_MAX];
    double body_pos[NB_MAX][3], body_ipos[NB_MAX][3], body_mass[NB_MAX], body_inertia[NB_MAX][6];
    double body_rbound[NB_MAX], body_invweight0[NB_MAX][2];
    double mesh_rbound[NB_MAX];     /* mjModel.geom_rbound of every hull's mesh geom (model compiler) */
    int planemesh_max; double planemesh_tol;   /* mjc_PlaneConvex: maxplanemesh = 3, tolplanemesh = 0.3 [MJ-ext] (blob field `planemesh`) */
    int dof_body[NV_MAX], dof_parent[NV_MAX], dof_madr[NV_MAX + 1];
    double dof_armature[NV_MAX], dof_invweight0[NV_MAX];
    double jnt_range[69][2];
    int jnt_limited[69];
    int vert_adr[NB_MAX + 1];
    double *verts; /* [nvert][3] body frame */
    int *vert_nbr_adr, *vert_nbr;   /* hull graph: neighbours (hull-local vertex ids) of vertex v are vert_nbr[vert_nbr_adr[v] .. vert_nbr_adr[v + 1]) */
    double kp[69], kd[69], torque_lim[69], a_scale[69];
    double timestep, gravity[3], solref[2], solimp[5], friction[3], margin, impratio, meaninertia;
    double rfc_scale, rfc_lim, base_rot[4];
    int nv_full;   /* dofs of the whole reference scene (humanoid + all objects): solver termination scale */
    int solver_iter;
    double solver_tol;
    int ls_iterations; double ls_tolerance;   /* mjOption.ls_iterations = 50, ls_tolerance = 0.01 (MuJoCo 2.1.0 defaults) */
    int ls_exact;                              /* 1: exact minimiser along the search direction instead of PrimalSearch (tests) */
    /* switches (tests) */
    int enable_contact, enable_limits;
} kpo_model;

/* static collision geometry of the active objects (chair / box / table / Can / step), world frame */
typedef struct { int type; double size[3], pos[3], mat[9], invw, rbound; int obj; /* owning dynamic object slot, -1 = static */ } kpo_geom;
/* a dynamic free object: inertial constants, body-frame geoms, free-joint state (qpos = pos + quat, qvel = world linear +
 * body-frame angular velocity, as MuJoCo's free joint) and the quantities of the last forward pass */
typedef struct {
    double mass, ipos[3], inertia[6], invw[2], arm;
    int ngeom; kpo_geom lgeom[MAXGEOM];
    double qpos[7], qvel[6];
    double xmat[9], xipos[3], Iw[9], M[36], bias[6];
} kpo_obj;

typedef struct {
    int ngeom, ngeom_static; kpo_geom geom[MAXGEOM];   /* world-frame geoms: static ones first, then those of the dynamic objects */
    int nobj; kpo_obj obj[MAXOBJ];                      /* both survive kpo_reset */
    double qpos[NQ_MAX], qvel[NV_MAX];
    double ctrl[69], qfrc_applied[NV_MAX];
    /* derived (state of the last forward pass) */
    double xpos[NB_MAX][3], xquat[NB_MAX][4], xmat[NB_MAX][9], xipos[NB_MAX][3];
    double xaxis[NV_MAX][3]; /* world axis of every rotational dof */
    double subtree_com[3];   /* of the root body: reference point of the c-frame */
    double cinert[NB_MAX][10], crb[NB_MAX][10];
    double cdof[NV_MAX][6], cdof_dot[NV_MAX][6];
    double cvel[NB_MAX][6], cacc[NB_MAX][6], cfrc[NB_MAX][6];
    double qM[NM_MAX], qLD[NM_MAX], qLDiagInv[NV_MAX];
    double qfrc_bias[NV_MAX], qfrc_smooth[NVT_MAX], qacc_smooth[NVT_MAX], qacc[NVT_MAX];   /* [humanoid 75 | 6 per object] */
    double qacc_warmstart[NVT_MAX], qfrc_constraint[NVT_MAX];
    /* contacts */
    int ncon;
    int con_body[MAXCON], con_b2[MAXCON];   /* entity carrying the vertex (0..23 hull, 24+k object k) / the surface (-1 world, 24+k) */
    double con_pos[MAXCON][3], con_dist[MAXCON], con_frame[MAXCON][9], con_invw2[MAXCON];
    /* constraint rows */
    int nefc;
    double efc_J[MAXEFC][NVT_MAX], efc_aref[MAXEFC], efc_D[MAXEFC], efc_force[MAXEFC];
    int solver_niter;
} kpo_data;

/* floating-point operation counter (multiplications + additions + divisions + square roots / trigonometric calls, counted at the loop
 * bodies below) of everything executed since kpo_flops_reset(): the "algorithmic FLOPs of the reference formulation" bench.py reports
 * (dense mj_fullM + Cholesky stable-PD, sparse L'DL, dense Newton Hessian -- the arithmetic MuJoCo + the reference's Python execute,
 * not the matrix-free passes of the HIP kernel). */
static double kpo_flops = 0.0;
#define FL(n) (kpo_flops += (double)(n))
void kpo_flops_reset(void) { kpo_flops = 0.0; }
double kpo_flops_get(void) { return kpo_flops; }

/* ------------------------------------------------------------------ small math */
static void v3_cross(double *r, const double *a, const double *b) {
    FL(9);
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
static double v3_dot(const double *a, const double *b) { FL(5); return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void quat_mul(double *r, const double *a, const double *b) { /* r = a (x) b, (w,x,y,z) */
    FL(28);
    double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat_normalize(double *q) { /* mju_normalize4: zero quat -> identity */
    FL(12);
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < MJ_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
    else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static void quat2mat(double *m, const double *q) {
    FL(37);
    double w = q[0], x = q[1], y = q[2], z = q[3];
    m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
    m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
    m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void mat_mulvec(double *r, const double *m, const double *v) {
    FL(15);
    double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
/* spatial: inert = [Ixx Iyy Izz Ixy Ixz Iyz | hx hy hz | m], v = [ang; lin]  (mju_mulInertVec) */
static void inert_mulvec(double *f, const double *I, const double *v) {
    FL(42);
    const double *w = v, *l = v + 3, *h = I + 6;
    double m = I[9];
    f[0] = I[0] * w[0] + I[3] * w[1] + I[4] * w[2] + (h[1] * l[2] - h[2] * l[1]);
    f[1] = I[3] * w[0] + I[1] * w[1] + I[5] * w[2] + (h[2] * l[0] - h[0] * l[2]);
    f[2] = I[4] * w[0] + I[5] * w[1] + I[2] * w[2] + (h[0] * l[1] - h[1] * l[0]);
    f[3] = m * l[0] - (h[1] * w[2] - h[2] * w[1]);
    f[4] = m * l[1] - (h[2] * w[0] - h[0] * w[2]);
    f[5] = m * l[2] - (h[0] * w[1] - h[1] * w[0]);
}
static void cross_motion(double *r, const double *v, const double *s) { /* mju_crossMotion */
    FL(3);
    double a[3], b[3], c[3];
    v3_cross(a, v, s); v3_cross(b, v, s + 3); v3_cross(c, v + 3, s);
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
    r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(double *r, const double *v, const double *f) { /* mju_crossForce */
    FL(3);
    double a[3], b[3], c[3];
    v3_cross(a, v, f); v3_cross(b, v + 3, f + 3); v3_cross(c, v, f + 3);
    r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
    r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}

/* ------------------------------------------------------------------ KPM blob loader */
typedef struct { char name[32]; uint32_t dtype, pad; uint64_t count, off; } kpm_entry;
static const void *kpm_find(const unsigned char *buf, const char *name, uint64_t *count, int dtype) {
    uint32_t n; memcpy(&n, buf + 8, 4);
    for (uint32_t i = 0; i < n; i++) {
        kpm_entry e; memcpy(&e, buf + 12 + 56 * i, 56);
        if (!strncmp(e.name, name, 32)) {
            if ((int)e.dtype != dtype) return NULL;
            if (count) *count = e.count;
            return buf + e.off;
        }
    }
    return NULL;
}
#define LOADF(dst, nm, cnt) do { uint64_t c_; const void *p_ = kpm_find(buf, nm, &c_, 0); \
    if (!p_ || c_ != (uint64_t)(cnt)) { fprintf(stderr, "kpo: bad field %s\n", nm); free(buf); free(m); return NULL; } \
    memcpy(dst, p_, 8 * (cnt)); } while (0)
#define LOADI(dst, nm, cnt) do { uint64_t c_; const void *p_ = kpm_find(buf, nm, &c_, 1); \
    if (!p_ || c_ != (uint64_t)(cnt)) { fprintf(stderr, "kpo: bad field %s\n", nm); free(buf); free(m); return NULL; } \
    memcpy(dst, p_, 4 * (cnt)); } while (0)

kpo_model *kpo_model_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *buf = malloc(sz);
    if (fread(buf, 1, sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
    fclose(f);
    uint32_t magic; memcpy(&magic, buf, 4);
    if (magic != 0x314D504B) { free(buf); return NULL; }
    kpo_model *m = calloc(1, sizeof(*m));
    int dims[9]; LOADI(dims, "dims", 9);
    m->nb = dims[0]; m->nv = dims[1]; m->nq = dims[2]; m->nu = dims[3]; m->nM = dims[4]; m->nvert = dims[5];
    if (m->nb != NB_MAX || m->nv != NV_MAX || m->nM != NM_MAX) { fprintf(stderr, "kpo: unexpected dims\n"); free(buf); free(m); return NULL; }
    LOADI(m->body_parent, "body_parent", m->nb); LOADI(m->body_subtree, "body_subtree", m->nb);
    LOADF(m->body_pos, "body_pos", 3 * m->nb); LOADF(m->body_ipos, "body_ipos", 3 * m->nb);
    LOADF(m->body_mass, "body_mass", m->nb); LOADF(m->body_inertia, "body_inertia", 6 * m->nb);
    LOADF(m->body_rbound, "body_rbound", m->nb); LOADF(m->body_invweight0, "body_invweight0", 2 * m->nb);
    { uint64_t c; const void *p = kpm_find(buf, "mesh_rbound", &c, 0), *q;
      if (!p || c != (uint64_t)m->nb) { fprintf(stderr, "kpo: blob has no mesh_rbound (recompile the model, KPM version 7)\n"); free(buf); free(m); return NULL; }
      memcpy(m->mesh_rbound, p, 8 * m->nb);
      double pm[2] = {3.0, 0.3};
      q = kpm_find(buf, "planemesh", &c, 0); if (q && c == 2) memcpy(pm, q, 16);
      m->planemesh_max = (int)pm[0]; m->planemesh_tol = pm[1]; }
    LOADI(m->dof_body, "dof_body", m->nv); LOADI(m->dof_parent, "dof_parent", m->nv);
    LOADI(m->dof_madr, "dof_madr", m->nv + 1);
    LOADF(m->dof_armature, "dof_armature", m->nv); LOADF(m->dof_invweight0, "dof_invweight0", m->nv);
    LOADF(m->jnt_range, "jnt_range", 2 * m->nu); LOADI(m->jnt_limited, "jnt_limited", m->nu);
    LOADI(m->vert_adr, "vert_adr", m->nb + 1);
    m->verts = malloc(sizeof(double) * 3 * m->nvert);
    { uint64_t c; const void *p = kpm_find(buf, "verts", &c, 0); memcpy(m->verts, p, 8 * 3 * m->nvert); }
    { uint64_t c, c2; const void *p = kpm_find(buf, "vert_nbr_adr", &c, 1), *q = kpm_find(buf, "vert_nbr", &c2, 1);
      if (!p || !q || c != (uint64_t)m->nvert + 1) { fprintf(stderr, "kpo: blob has no hull graph (recompile the model, KPM version 6)\n"); free(buf); free(m->verts); free(m); return NULL; }
      m->vert_nbr_adr = malloc(4 * c); memcpy(m->vert_nbr_adr, p, 4 * c); m->vert_nbr = malloc(4 * (c2 ? c2 : 1)); memcpy(m->vert_nbr, q, 4 * c2); }
    LOADF(m->kp, "kp", m->nu); LOADF(m->kd, "kd", m->nu); LOADF(m->torque_lim, "torque_lim", m->nu);
    LOADF(m->a_scale, "a_scale", m->nu);
    double opt[26]; LOADF(opt, "opt", 26);
    m->timestep = opt[0]; memcpy(m->gravity, opt + 1, 24); memcpy(m->solref, opt + 4, 16);
    memcpy(m->solimp, opt + 6, 40); memcpy(m->friction, opt + 11, 24); m->margin = opt[14];
    m->impratio = opt[15]; m->meaninertia = opt[16]; m->rfc_scale = opt[17]; m->rfc_lim = opt[18];
    memcpy(m->base_rot, opt + 19, 32); m->solver_iter = (int)opt[23]; m->solver_tol = opt[24]; m->nv_full = (int)opt[25];
    m->enable_contact = 1; m->enable_limits = 1;
    m->ls_iterations = 50; m->ls_tolerance = 0.01; m->ls_exact = 0;
    free(buf);
    return m;
}
void kpo_model_free(kpo_model *m) { if (m) { free(m->verts); free(m->vert_nbr_adr); free(m->vert_nbr); free(m); } }
void kpo_model_set_flags(kpo_model *m, int contact, int limits) { m->enable_contact = contact; m->enable_limits = limits; }
void kpo_model_set_gravity(kpo_model *m, double gz) { m->gravity[2] = gz; }
void kpo_model_set_gravity3(kpo_model *m, double gx, double gy, double gz) { m->gravity[0] = gx; m->gravity[1] = gy; m->gravity[2] = gz; }
void kpo_model_set_ls_exact(kpo_model *m, int exact) { m->ls_exact = exact; }
void kpo_model_set_planemesh(kpo_model *m, int maxcon, double tol) { m->planemesh_max = maxcon; m->planemesh_tol = tol; }
kpo_data *kpo_data_new(void) { return calloc(1, sizeof(kpo_data)); }
void kpo_data_free(kpo_data *d) { free(d); }
size_t kpo_data_sizeof(void) { return sizeof(kpo_data); }

/* ------------------------------------------------------------------ forward kinematics  [MJ-ext mj_kinematics + mj_comPos] */
static void kpo_kinematics(const kpo_model *m, kpo_data *d) {
    quat_normalize(d->qpos + 3); /* mj_kinematics normalises qpos quaternions */
    for (int b = 0; b < m->nb; b++) {
        if (b == 0) {
            memcpy(d->xpos[0], d->qpos, 24); memcpy(d->xquat[0], d->qpos + 3, 32);
            quat2mat(d->xmat[0], d->xquat[0]);
            for (int k = 0; k < 3; k++) { /* rotational free dofs: body-frame axes */
                d->xaxis[3 + k][0] = d->xmat[0][k]; d->xaxis[3 + k][1] = d->xmat[0][3 + k]; d->xaxis[3 + k][2] = d->xmat[0][6 + k];
                d->xaxis[k][0] = k == 0; d->xaxis[k][1] = k == 1; d->xaxis[k][2] = k == 2;
            }
        } else {
            int p = m->body_parent[b];
            double off[3]; mat_mulvec(off, d->xmat[p], m->body_pos[b]);
            for (int k = 0; k < 3; k++) d->xpos[b][k] = d->xpos[p][k] + off[k];
            double q[4]; memcpy(q, d->xquat[p], 32);
            const double *ang = d->qpos + 7 + 3 * (b - 1);
            static const double ax_local[3][3] = {{0, 0, 1}, {0, 1, 0}, {1, 0, 0}}; /* hinge order z, y, x */
            for (int j = 0; j < 3; j++) {
                double R[9]; quat2mat(R, q);
                mat_mulvec(d->xaxis[6 + 3 * (b - 1) + j], R, ax_local[j]);
                FL(10);
                double h = 0.5 * ang[j], s = sin(h);
                double qj[4] = {cos(h), ax_local[j][0] * s, ax_local[j][1] * s, ax_local[j][2] * s};
                double qn[4]; quat_mul(qn, q, qj); memcpy(q, qn, 32);
            }
            quat_normalize(q);
            memcpy(d->xquat[b], q, 32); quat2mat(d->xmat[b], q);
        }
        double io[3]; mat_mulvec(io, d->xmat[b], m->body_ipos[b]);
        FL(6);
        for (int k = 0; k < 3; k++) d->xipos[b][k] = d->xpos[b][k] + io[k];
    }
    /* subtree COM of the root = whole-humanoid COM */
    double M = 0, c[3] = {0, 0, 0};
    for (int b = 0; b < m->nb; b++) { FL(7); M += m->body_mass[b]; for (int k = 0; k < 3; k++) c[k] += m->body_mass[b] * d->xipos[b][k]; }
    for (int k = 0; k < 3; k++) d->subtree_com[k] = c[k] / M;
    /* cinert: body inertia about subtree_com, world axes */
    for (int b = 0; b < m->nb; b++) {
        const double *R = d->xmat[b], *Ib = m->body_inertia[b];
        double I3[9] = {Ib[0], Ib[3], Ib[4], Ib[3], Ib[1], Ib[5], Ib[4], Ib[5], Ib[2]};
        double T[9], W[9];
        FL(90 + 36);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += R[3 * i + k] * I3[3 * k + j]; T[3 * i + j] = s; }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += T[3 * i + k] * R[3 * j + k]; W[3 * i + j] = s; }
        double r[3] = {d->xipos[b][0] - d->subtree_com[0], d->xipos[b][1] - d->subtree_com[1], d->xipos[b][2] - d->subtree_com[2]};
        double mass = m->body_mass[b], rr = v3_dot(r, r);
        double *ci = d->cinert[b];
        ci[0] = W[0] + mass * (rr - r[0] * r[0]); ci[1] = W[4] + mass * (rr - r[1] * r[1]); ci[2] = W[8] + mass * (rr - r[2] * r[2]);
        ci[3] = W[1] - mass * r[0] * r[1]; ci[4] = W[2] - mass * r[0] * r[2]; ci[5] = W[5] - mass * r[1] * r[2];
        ci[6] = mass * r[0]; ci[7] = mass * r[1]; ci[8] = mass * r[2]; ci[9] = mass;
    }
    /* cdof */
    for (int i = 0; i < m->nv; i++) {
        double *cd = d->cdof[i];
        if (i < 3) { cd[0] = cd[1] = cd[2] = 0; cd[3] = i == 0; cd[4] = i == 1; cd[5] = i == 2; }
        else {
            int b = m->dof_body[i];
            double off[3] = {d->subtree_com[0] - d->xpos[b][0], d->subtree_com[1] - d->xpos[b][1], d->subtree_com[2] - d->xpos[b][2]};
            memcpy(cd, d->xaxis[i], 24); v3_cross(cd + 3, d->xaxis[i], off);
        }
    }
}

/* composite rigid body -> qM (sparse), mj_crb  [MJ-ext] */
static void kpo_crb(const kpo_model *m, kpo_data *d) {
    memcpy(d->crb, d->cinert, sizeof(d->crb));
    FL(10 * (m->nb - 1));
    for (int b = m->nb - 1; b > 0; b--) for (int k = 0; k < 10; k++) d->crb[m->body_parent[b]][k] += d->crb[b][k];
    memset(d->qM, 0, sizeof(d->qM));
    for (int i = 0; i < m->nv; i++) {
        double buf[6]; inert_mulvec(buf, d->crb[m->dof_body[i]], d->cdof[i]);
        int adr = m->dof_madr[i];
        d->qM[adr] = m->dof_armature[i];
        for (int j = i; j >= 0; j = m->dof_parent[j], adr++) {
            double s = 0; for (int k = 0; k < 6; k++) s += d->cdof[j][k] * buf[k];
            FL(13);
            d->qM[adr] += s;
        }
    }
}
/* sparse L'DL, mj_factorM  [MJ-ext] */
static void kpo_factor_sparse(const kpo_model *m, const double *qM, double *qLD, double *diaginv) {
    memcpy(qLD, qM, sizeof(double) * m->nM);
    for (int k = m->nv - 1; k >= 0; k--) {
        int Mkk = m->dof_madr[k];
        int i = m->dof_parent[k], Mki = Mkk + 1;
        while (i >= 0) {
            double tmp = qLD[Mki] / qLD[Mkk];
            int cnt = m->dof_madr[i + 1] - m->dof_madr[i];
            FL(1 + 2 * cnt);
            for (int c = 0; c < cnt; c++) qLD[m->dof_madr[i] + c] -= tmp * qLD[Mki + c];
            qLD[Mki] = tmp;
            i = m->dof_parent[i]; Mki++;
        }
    }
    for (int i = 0; i < m->nv; i++) diaginv[i] = 1.0 / qLD[m->dof_madr[i]];
}
static void kpo_solve_sparse(const kpo_model *m, const double *qLD, const double *diaginv, double *x) {
    for (int i = m->nv - 1; i >= 0; i--) {
        if (x[i] != 0) { int adr = m->dof_madr[i] + 1; for (int j = m->dof_parent[i]; j >= 0; j = m->dof_parent[j], adr++) { FL(2); x[j] -= qLD[adr] * x[i]; } }
    }
    FL(m->nv);
    for (int i = 0; i < m->nv; i++) x[i] *= diaginv[i];
    for (int i = 0; i < m->nv; i++) { int adr = m->dof_madr[i] + 1; for (int j = m->dof_parent[i]; j >= 0; j = m->dof_parent[j], adr++) { FL(2); x[i] -= qLD[adr] * x[j]; } }
}
/* dense nv x nv from sparse (mj_fullM) */
void kpo_fullM(const kpo_model *m, const kpo_data *d, double *M) {
    memset(M, 0, sizeof(double) * m->nv * m->nv);
    for (int i = 0; i < m->nv; i++) { int adr = m->dof_madr[i]; for (int j = i; j >= 0; j = m->dof_parent[j], adr++) M[i * m->nv + j] = M[j * m->nv + i] = d->qM[adr]; }
}
static void kpo_mulM(const kpo_model *m, const double *qM, const double *v, double *r) {
    for (int i = 0; i < m->nv; i++) r[i] = 0;
    for (int i = 0; i < m->nv; i++) { int adr = m->dof_madr[i]; r[i] += qM[adr] * v[i]; adr++;
        for (int j = m->dof_parent[i]; j >= 0; j = m->dof_parent[j], adr++) { FL(4); r[i] += qM[adr] * v[j]; r[j] += qM[adr] * v[i]; } }
}

/* mj_comVel + mj_rne(flg_acc=0) -> qfrc_bias  [MJ-ext] */
static void kpo_vel_bias(const kpo_model *m, kpo_data *d) {
    for (int b = 0; b < m->nb; b++) {
        double cv[6], ca[6];
        if (b == 0) { memset(cv, 0, 48); ca[0] = ca[1] = ca[2] = 0; ca[3] = -m->gravity[0]; ca[4] = -m->gravity[1]; ca[5] = -m->gravity[2]; }
        else { memcpy(cv, d->cvel[m->body_parent[b]], 48); memcpy(ca, d->cacc[m->body_parent[b]], 48); }
        if (b == 0) {
            for (int k = 0; k < 3; k++) { memset(d->cdof_dot[k], 0, 48); for (int c = 0; c < 6; c++) cv[c] += d->cdof[k][c] * d->qvel[k]; }
            for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot[k], cv, d->cdof[k]);
            for (int k = 3; k < 6; k++) for (int c = 0; c < 6; c++) cv[c] += d->cdof[k][c] * d->qvel[k];
            for (int k = 0; k < 6; k++) for (int c = 0; c < 6; c++) ca[c] += d->cdof_dot[k][c] * d->qvel[k];
        } else {
            for (int j = 0; j < 3; j++) {
                int i = 6 + 3 * (b - 1) + j;
                cross_motion(d->cdof_dot[i], cv, d->cdof[i]);
                for (int c = 0; c < 6; c++) cv[c] += d->cdof[i][c] * d->qvel[i];
                for (int c = 0; c < 6; c++) ca[c] += d->cdof_dot[i][c] * d->qvel[i];
            }
        }
        memcpy(d->cvel[b], cv, 48); memcpy(d->cacc[b], ca, 48);
        FL(b == 0 ? 6 * 24 : 3 * 24); FL(6);
        double Ia[6], Iv[6], cf[6];
        inert_mulvec(Ia, d->cinert[b], ca); inert_mulvec(Iv, d->cinert[b], cv); cross_force(cf, cv, Iv);
        for (int c = 0; c < 6; c++) d->cfrc[b][c] = Ia[c] + cf[c];
    }
    FL(6 * (m->nb - 1) + 11 * m->nv);
    for (int b = m->nb - 1; b > 0; b--) for (int c = 0; c < 6; c++) d->cfrc[m->body_parent[b]][c] += d->cfrc[b][c];
    for (int i = 0; i < m->nv; i++) { double s = 0; for (int c = 0; c < 6; c++) s += d->cdof[i][c] * d->cfrc[m->dof_body[i]][c]; d->qfrc_bias[i] = s; }
}

/* ------------------------------------------------------------------ dynamic free objects
 * The same mj_kinematics / mj_crb / mj_rne arithmetic as above, written out for a single free body [MJ-ext].  Dofs of
 * object k sit at nv + 6k: 3 world-frame translations of the body origin, then 3 rotations about the body axes. */
static int kpo_nvt(const kpo_model *m, const kpo_data *d) { return m->nv + 6 * d->nobj; }
static void kpo_obj_forward(const kpo_model *m, kpo_data *d) {
    d->ngeom = d->ngeom_static;
    for (int k = 0; k < d->nobj; k++) {
        kpo_obj *o = &d->obj[k];
        quat_normalize(o->qpos + 3);
        quat2mat(o->xmat, o->qpos + 3);
        const double *R = o->xmat;
        double r[3]; mat_mulvec(r, R, o->ipos);                       /* com - body origin, world axes */
        for (int a = 0; a < 3; a++) o->xipos[a] = o->qpos[a] + r[a];
        const double *Ib = o->inertia;
        double I3[9] = {Ib[0], Ib[3], Ib[4], Ib[3], Ib[1], Ib[5], Ib[4], Ib[5], Ib[2]}, T[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double v = 0; for (int c = 0; c < 3; c++) v += R[3 * i + c] * I3[3 * c + j]; T[3 * i + j] = v; }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double v = 0; for (int c = 0; c < 3; c++) v += T[3 * i + c] * R[3 * j + c]; o->Iw[3 * i + j] = v; }
        for (int gi = 0; gi < o->ngeom && d->ngeom < MAXGEOM; gi++) {
            const kpo_geom *l = &o->lgeom[gi]; kpo_geom *g = &d->geom[d->ngeom++];
            *g = *l; g->obj = k; g->invw = o->invw[0];
            double pw[3]; mat_mulvec(pw, R, l->pos);
            for (int a = 0; a < 3; a++) g->pos[a] = o->qpos[a] + pw[a];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double v = 0; for (int c = 0; c < 3; c++) v += R[3 * i + c] * l->mat[3 * c + j]; g->mat[3 * i + j] = v; }
        }
        /* velocity Jacobians of (com velocity, angular velocity) w.r.t. the 6 dofs */
        double Jv[3][6], Jw[3][6];
        memset(Jv, 0, sizeof(Jv)); memset(Jw, 0, sizeof(Jw));
        for (int j = 0; j < 3; j++) {
            Jv[j][j] = 1;
            double ax[3] = {R[j], R[3 + j], R[6 + j]}, axr[3]; v3_cross(axr, ax, r);
            for (int a = 0; a < 3; a++) { Jv[a][3 + j] = axr[a]; Jw[a][3 + j] = ax[a]; }
        }
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) {
            double v = 0;
            for (int a = 0; a < 3; a++) v += o->mass * Jv[a][i] * Jv[a][j];
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) v += Jw[a][i] * o->Iw[3 * a + b] * Jw[b][j];
            o->M[6 * i + j] = v + (i == j ? o->arm : 0.0);
        }
        double w[3]; mat_mulvec(w, R, o->qvel + 3);                    /* world angular velocity */
        double wr[3], wwr[3], Iw_w[3], gyro[3];
        v3_cross(wr, w, r); v3_cross(wwr, w, wr); mat_mulvec(Iw_w, o->Iw, w); v3_cross(gyro, w, Iw_w);
        double fl[3] = {o->mass * (wwr[0] - m->gravity[0]), o->mass * (wwr[1] - m->gravity[1]), o->mass * (wwr[2] - m->gravity[2])};
        for (int i = 0; i < 6; i++) { double v = 0; for (int a = 0; a < 3; a++) v += Jv[a][i] * fl[a] + Jw[a][i] * gyro[a]; o->bias[i] = v; }
    }
}
/* translational Jacobian (times sign) of world point p moving with entity ent (0..23 hull body, 24+k object k) */
static void kpo_point_jac(const kpo_model *m, const kpo_data *d, int ent, const double *p, double sign, double Jp[3][NVT_MAX]) {
    if (ent < 0) return;
    if (ent < NB_MAX) {
        double r[3] = {p[0] - d->subtree_com[0], p[1] - d->subtree_com[1], p[2] - d->subtree_com[2]};
        int last = ent == 0 ? 5 : 6 + 3 * (ent - 1) + 2;
        for (int i = last; i >= 0; i = m->dof_parent[i]) {
            double wxr[3]; v3_cross(wxr, d->cdof[i], r);
            FL(9);
            for (int k = 0; k < 3; k++) Jp[k][i] += sign * (d->cdof[i][3 + k] + wxr[k]);
        }
    } else {
        const kpo_obj *o = &d->obj[ent - NB_MAX];
        int base = m->nv + 6 * (ent - NB_MAX);
        double r[3] = {p[0] - o->qpos[0], p[1] - o->qpos[1], p[2] - o->qpos[2]};
        for (int j = 0; j < 3; j++) {
            Jp[j][base + j] += sign;
            double ax[3] = {o->xmat[j], o->xmat[3 + j], o->xmat[6 + j]}, axr[3]; v3_cross(axr, ax, r);
            for (int k = 0; k < 3; k++) Jp[k][base + 3 + j] += sign * axr[k];
        }
    }
}
#include "kp_collide.h"

/* mju_makeFrame: x-axis given, y from (0,1,0) or (0,0,1) made orthogonal, z = x cross y  [MJ-ext] */
static void kpo_make_frame(double *fr) {
    double *x = fr, *y = fr + 3, *z = fr + 6;
    if (fabs(x[1]) < 0.5) { y[0] = 0; y[1] = 1; y[2] = 0; } else { y[0] = 0; y[1] = 0; y[2] = 1; }
    double dp = v3_dot(x, y);
    for (int k = 0; k < 3; k++) y[k] -= dp * x[k];
    double n = sqrt(v3_dot(y, y));
    for (int k = 0; k < 3; k++) y[k] /= n;
    v3_cross(z, x, y);
}

static void kpo_shape_of_geom(const kpo_geom *g, kpc_shape *s) {
    s->type = g->type; memcpy(s->size, g->size, 24); memcpy(s->pos, g->pos, 24); memcpy(s->mat, g->mat, 72); memcpy(s->center, g->pos, 24);
    s->verts = NULL; s->nvert = 0;
}
static void kpo_shape_of_hull(const kpo_model *m, const kpo_data *d, int b, kpc_shape *s) {
    s->type = 2; s->size[0] = s->size[1] = s->size[2] = 0; memcpy(s->pos, d->xpos[b], 24); memcpy(s->mat, d->xmat[b], 72);
    memcpy(s->center, d->xipos[b], 24);              /* MuJoCo centres a mesh geom on the mesh's COM: geom_xpos == xipos for these one-geom bodies */
    s->verts = m->verts + 3 * m->vert_adr[b]; s->nvert = m->vert_adr[b + 1] - m->vert_adr[b];
}
/* append the contacts of one geom pair.  ent_a / ent_b: entities of the solver (0..23 hull body, 24+k object, -1 world); the
 * stored normal points from ent_b's geom into ent_a's (flip = the MuJoCo normal, geom 1 -> geom 2, runs the other way) */
static void kpo_emit(kpo_data *d, const kpc_contact *c, int n, int ent_a, int ent_b, double invw_b, int flip) {
    for (int i = 0; i < n && d->ncon < MAXCON; i++) {
        int k = d->ncon++;
        d->con_body[k] = ent_a; d->con_b2[k] = ent_b; d->con_dist[k] = c[i].dist; d->con_invw2[k] = invw_b;
        memcpy(d->con_pos[k], c[i].pos, 24);
        for (int a = 0; a < 3; a++) d->con_frame[k][a] = flip ? -c[i].normal[a] : c[i].normal[a];
        kpo_make_frame(d->con_frame[k]);
    }
}
/* MuJoCo orders a pair by geom type (cylinder 5 < box 6 < mesh 7) and, for equal types, by geom id  [MJ-ext] */
static int kpo_type_rank(int type) { return type == 1 ? 5 : (type == 0 ? 6 : 7); }

/* mj_collision for the scene: contacts are emitted entity by entity (24 hulls in body order: floor first, then the object geoms;
 * then the dynamic objects' geoms: floor first, then the geoms of objects in higher slots), mid-phase = bounding spheres. */
static void kpo_collide(const kpo_model *m, kpo_data *d) {
    d->ncon = 0;
    if (!m->enable_contact) return;
    kpc_contact con[KPC_MAXPAIR];
    for (int b = 0; b < m->nb; b++) {
        kpc_shape hull; kpo_shape_of_hull(m, d, b, &hull);
        for (int gi = -1; gi < d->ngeom; gi++) {
            const kpo_geom *g = gi < 0 ? NULL : &d->geom[gi];
            if (!g) {
                if (d->xpos[b][2] - m->body_rbound[b] > m->margin) continue;
                int n = kpo_plane_mesh(&hull, m->vert_nbr_adr + m->vert_adr[b], m->vert_nbr, m->margin, m->planemesh_tol * m->mesh_rbound[b], m->planemesh_max, con);
                kpo_emit(d, con, n, b, -1, 0.0, 0);
            } else {
                double dx[3] = {d->xpos[b][0] - g->pos[0], d->xpos[b][1] - g->pos[1], d->xpos[b][2] - g->pos[2]};
                if (sqrt(v3_dot(dx, dx)) - m->body_rbound[b] - g->rbound > m->margin) continue;
                kpc_shape gs; kpo_shape_of_geom(g, &gs);
                int n = kpo_convex(&gs, &hull, m->margin, con);           /* g1 = box / cylinder, g2 = mesh: normal into the hull */
                if (getenv("KPO_MPR_TRACE")) fprintf(stderr, "mpr hull %d geom %d (type %d): n %d discover %d refine %d penetr %d dist %.6f\n", b, gi, g->type, n, kpc_stat_discover, kpc_stat_refine, kpc_stat_penetr, n ? con[0].dist : 0.0);
                kpo_emit(d, con, n, b, g->obj >= 0 ? NB_MAX + g->obj : -1, g->invw, 0);
            }
        }
    }
    for (int gi = d->ngeom_static; gi < d->ngeom; gi++) {
        const kpo_geom *ga = &d->geom[gi];
        kpc_shape sa; kpo_shape_of_geom(ga, &sa);
        for (int gj = -1; gj < d->ngeom; gj++) {
            const kpo_geom *gb = gj < 0 ? NULL : &d->geom[gj];
            if (gb && (gb->obj < 0 || gb->obj <= ga->obj)) continue;
            if (!gb) {
                if (ga->pos[2] - ga->rbound > m->margin) continue;
                int n = ga->type == 0 ? kpo_plane_box(&sa, m->margin, con) : kpo_plane_cylinder(&sa, m->margin, con);
                kpo_emit(d, con, n, NB_MAX + ga->obj, -1, 0.0, 0);
            } else {
                double dx[3] = {ga->pos[0] - gb->pos[0], ga->pos[1] - gb->pos[1], ga->pos[2] - gb->pos[2]};
                if (sqrt(v3_dot(dx, dx)) - ga->rbound - gb->rbound > m->margin) continue;
                kpc_shape sb; kpo_shape_of_geom(gb, &sb);
                /* geom 1 = lower type rank, then lower geom id (ga comes first in the scene: lower object index) */
                int a_first = kpo_type_rank(ga->type) <= kpo_type_rank(gb->type);
                const kpc_shape *g1 = a_first ? &sa : &sb, *g2 = a_first ? &sb : &sa;
                int n = (ga->type == 0 && gb->type == 0) ? kpo_box_box(g1, g2, m->margin, con) : kpo_convex(g1, g2, m->margin, con);
                kpo_emit(d, con, n, NB_MAX + ga->obj, NB_MAX + gb->obj, gb->invw, a_first);   /* stored normal: from gb into ga */
            }
        }
    }
}

/* getimpedance [MJ-ext engine_core_constraint.c] */
static double kpo_impedance(const kpo_model *m, double pos) {
    double d0 = fmin(MJ_MAXIMP, fmax(MJ_MINIMP, m->solimp[0])), dw = fmin(MJ_MAXIMP, fmax(MJ_MINIMP, m->solimp[1]));
    double width = m->solimp[2], mid = fmin(MJ_MAXIMP, fmax(MJ_MINIMP, m->solimp[3])), power = fmax(1.0, m->solimp[4]);
    if (d0 == dw || width <= MJ_MINVAL) return 0.5 * (d0 + dw);
    double x = fabs(pos) / width, y;
    if (x >= 1) return dw;
    if (x <= 0) return d0;
    if (power == 1) y = x;
    else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
    else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
    return d0 + y * (dw - d0);
}

/* mj_makeConstraint + mj_makeImpedance: joint limits then pyramidal contacts  [MJ-ext] */
static void kpo_make_constraint(const kpo_model *m, kpo_data *d) {
    int ne = 0, nv = kpo_nvt(m, d);
    double qvel[NVT_MAX];
    memcpy(qvel, d->qvel, sizeof(double) * m->nv);
    for (int k = 0; k < d->nobj; k++) memcpy(qvel + m->nv + 6 * k, d->obj[k].qvel, 48);
    double tc = fmax(m->solref[0], 2 * m->timestep), dr = m->solref[1], dmax = fmin(MJ_MAXIMP, fmax(MJ_MINIMP, m->solimp[1]));
    double K = 1.0 / (dmax * dmax * tc * tc * dr * dr), B = 2.0 / (dmax * tc);
    if (m->enable_limits) {
        for (int j = 0; j < m->nu; j++) {
            if (!m->jnt_limited[j]) continue;
            double q = d->qpos[7 + j];
            for (int side = -1; side <= 1; side += 2) {
                double dist = side < 0 ? q - m->jnt_range[j][0] : m->jnt_range[j][1] - q;
                if (dist >= 0) continue; /* margin 0 */
                memset(d->efc_J[ne], 0, sizeof(double) * nv);
                d->efc_J[ne][6 + j] = -side;
                double imp = kpo_impedance(m, dist);
                double R = fmax(MJ_MINVAL, (1 - imp) * m->dof_invweight0[6 + j] / imp);
                d->efc_D[ne] = 1.0 / R;
                d->efc_aref[ne] = -B * (-side * d->qvel[6 + j]) - K * imp * dist;
                ne++;
            }
        }
    }
    for (int c = 0; c < d->ncon; c++) {
        int b = d->con_body[c];
        /* translational Jacobian of the contact point: entity carrying the vertex minus entity carrying the surface */
        double Jp[3][NVT_MAX]; memset(Jp, 0, sizeof(Jp));
        kpo_point_jac(m, d, b, d->con_pos[c], 1.0, Jp);
        kpo_point_jac(m, d, d->con_b2[c], d->con_pos[c], -1.0, Jp);
        const double *fr = d->con_frame[c];
        double mu = m->friction[0]; /* impratio 1 */
        double tran = (b < NB_MAX ? m->body_invweight0[b][0] : d->obj[b - NB_MAX].invw[0]) + d->con_invw2[c];
        double imp = kpo_impedance(m, d->con_dist[c] - m->margin);
        double dA = tran + mu * mu * tran;
        double Rn = fmax(MJ_MINVAL, (1 - imp) * dA / imp);
        double Rpy = 2 * mu * mu * Rn;
        for (int e = 0; e < 4; e++) {
            int t = 1 + e / 2; double sgn = (e & 1) ? -1.0 : 1.0;
            double vel = 0;
            FL(14 * nv + 6);
            for (int i = 0; i < nv; i++) {
                double jn = fr[0] * Jp[0][i] + fr[1] * Jp[1][i] + fr[2] * Jp[2][i];
                double jt = fr[3 * t] * Jp[0][i] + fr[3 * t + 1] * Jp[1][i] + fr[3 * t + 2] * Jp[2][i];
                d->efc_J[ne][i] = jn + sgn * mu * jt;
                vel += d->efc_J[ne][i] * qvel[i];
            }
            d->efc_D[ne] = 1.0 / Rpy;
            d->efc_aref[ne] = -B * vel - K * imp * (d->con_dist[c] - m->margin);
            ne++;
        }
    }
    d->nefc = ne;
}

/* dense Cholesky helpers */
static int chol_factor(double *A, int n) { /* lower, in place */
    for (int j = 0; j < n; j++) {
        double s = A[j * n + j];
        FL(2 * j + 2 + (double)(n - j - 1) * (2 * j + 1));
        for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
        if (s <= 0) return -1;
        s = sqrt(s); A[j * n + j] = s;
        for (int i = j + 1; i < n; i++) { double t = A[i * n + j]; for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k]; A[i * n + j] = t / s; }
    }
    return 0;
}
static void chol_solve(const double *L, int n, double *x) {
    FL(2.0 * n * n);
    for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
}

/* primal cost at qacc: returns cost; fills jar-derived force and gradient */
static void kpo_mulM_full(const kpo_model *m, const kpo_data *d, const double *v, double *r) {
    kpo_mulM(m, d->qM, v, r);
    for (int k = 0; k < d->nobj; k++) for (int i = 0; i < 6; i++) {
        double s = 0; for (int j = 0; j < 6; j++) s += d->obj[k].M[6 * i + j] * v[m->nv + 6 * k + j];
        r[m->nv + 6 * k + i] = s;
    }
}
static double kpo_cost(const kpo_model *m, kpo_data *d, const double *qacc, double *grad, double *jar_out) {
    int nv = kpo_nvt(m, d);
    double Ma[NVT_MAX]; kpo_mulM_full(m, d, qacc, Ma);
    double cost = 0;
    FL(5 * nv + 2.0 * nv * d->nefc);
    for (int i = 0; i < nv; i++) cost += 0.5 * (Ma[i] - d->qfrc_smooth[i]) * (qacc[i] - d->qacc_smooth[i]);
    for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
    for (int e = 0; e < d->nefc; e++) {
        double jar = -d->efc_aref[e];
        for (int i = 0; i < nv; i++) jar += d->efc_J[e][i] * qacc[i];
        if (jar_out) jar_out[e] = jar;
        if (jar < 0) { FL(5 + 2 * nv); cost += 0.5 * d->efc_D[e] * jar * jar; d->efc_force[e] = -d->efc_D[e] * jar; for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[e][i] * d->efc_force[e]; }
        else d->efc_force[e] = 0;
    }
    if (grad) for (int i = 0; i < nv; i++) grad[i] = Ma[i] - d->qfrc_smooth[i] - d->qfrc_constraint[i];
    return cost;
}

/* PrimalSearch of engine_solver.c  [MJ-ext]: the line search mj_solNewton runs (mjOption.ls_iterations = 50, ls_tolerance = 0.01).
 * One Newton step on phi' from alpha = 0; further Newton steps while the derivative keeps its sign; once the minimum is
 * bracketed, midpoint + the Newton points of both bracket ends are the candidates of every round.  It returns as soon as a point
 * has |phi'| < gtol = tolerance * ls_tolerance * |search| * meaninertia * nv.  (The exact minimiser, phi' = 0, satisfies the same
 * test; the two differ by at most gtol / phi'' in alpha.) */
typedef struct { int ne; double g0, h0; const double *jar, *jv, *D; int evals; } kpo_ls_ctx;
typedef struct { double alpha, cost, deriv[2]; } kpo_ls_pnt;
static void kpo_ls_eval(kpo_ls_ctx *c, kpo_ls_pnt *p) {            /* PrimalEval: cost relative to alpha = 0 and its two derivatives */
    double a = p->alpha, cost = a * c->g0 + 0.5 * a * a * c->h0, d1 = c->g0 + a * c->h0, d2 = c->h0;
    FL(8 + 10.0 * c->ne);
    for (int e = 0; e < c->ne; e++) {
        double x0 = c->jar[e], x = x0 + a * c->jv[e];
        if (x < 0) { cost += 0.5 * c->D[e] * x * x; d1 += c->D[e] * x * c->jv[e]; d2 += c->D[e] * c->jv[e] * c->jv[e]; }
        if (x0 < 0) cost -= 0.5 * c->D[e] * x0 * x0;
    }
    p->cost = cost; p->deriv[0] = d1; p->deriv[1] = d2; c->evals++;
}
static int kpo_ls_update_bracket(kpo_ls_ctx *c, kpo_ls_pnt *p, const kpo_ls_pnt cand[3], kpo_ls_pnt *pnext) {
    int flag = 0;
    for (int i = 0; i < 3; i++) {
        if (p->deriv[0] < 0 && cand[i].deriv[0] < 0 && p->deriv[0] < cand[i].deriv[0]) { *p = cand[i]; flag = 1; }
        else if (p->deriv[0] > 0 && cand[i].deriv[0] > 0 && p->deriv[0] > cand[i].deriv[0]) { *p = cand[i]; flag = 2; }
    }
    if (flag) { pnext->alpha = p->alpha - p->deriv[0] / p->deriv[1]; kpo_ls_eval(c, pnext); }
    return flag;
}
static double kpo_primal_search(kpo_ls_ctx *c, double snorm, double gtol_per_snorm, int ls_iterations) {
    kpo_ls_pnt p0, p1, p2, pmid, p1next, p2next;
    if (snorm < MJ_MINVAL) return 0;
    double gtol = gtol_per_snorm * snorm;
    p0.alpha = 0; kpo_ls_eval(c, &p0);
    if (!(p0.deriv[1] > 0)) return 0;
    p1.alpha = p0.alpha - p0.deriv[0] / p0.deriv[1]; kpo_ls_eval(c, &p1);
    if (p0.cost < p1.cost) p1 = p0;
    if (fabs(p1.deriv[0]) < gtol) return p1.alpha;
    int dir = p1.deriv[0] < 0 ? 1 : -1, p2update = 0;
    p2 = p1;
    while (p1.deriv[0] * dir <= -gtol && c->evals < ls_iterations) {
        p2 = p1; p2update = 1;
        p1.alpha -= p1.deriv[0] / p1.deriv[1]; kpo_ls_eval(c, &p1);
        if (fabs(p1.deriv[0]) < gtol) return p1.alpha;
    }
    if (c->evals >= ls_iterations || !p2update) return p1.alpha;
    p2next = p1;
    p1next.alpha = p1.alpha - p1.deriv[0] / p1.deriv[1]; kpo_ls_eval(c, &p1next);
    while (c->evals < ls_iterations) {
        pmid.alpha = 0.5 * (p1.alpha + p2.alpha); kpo_ls_eval(c, &pmid);
        kpo_ls_pnt cand[3] = {p1next, p2next, pmid};
        int best = -1; double bc = 0;
        for (int i = 0; i < 3; i++) if (fabs(cand[i].deriv[0]) < gtol && (best < 0 || cand[i].cost < bc)) { bc = cand[i].cost; best = i; }
        if (best >= 0) return cand[best].alpha;
        int b1 = kpo_ls_update_bracket(c, &p1, cand, &p1next), b2 = kpo_ls_update_bracket(c, &p2, cand, &p2next);
        if (!b1 && !b2) return pmid.cost < p0.cost ? pmid.alpha : 0;
    }
    if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
    if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
    return 0;
}

/* Newton solver on the primal problem (mj_solNewton: mj_solPrimal with flg_Newton).  [MJ-ext] */
static void kpo_solve_constraint(const kpo_model *m, kpo_data *d) {
    int nv = kpo_nvt(m, d), ne = d->nefc;
    d->solver_niter = 0;
    if (ne == 0) { memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv); memset(d->qfrc_constraint, 0, sizeof(double) * nv); return; }
    static double H[NVT_MAX * NVT_MAX], Mfull[NVT_MAX * NVT_MAX];
    double qacc[NVT_MAX], grad[NVT_MAX], jar[MAXEFC], search[NVT_MAX], jv[MAXEFC], Mv[NVT_MAX];
    /* warmstart choice */
    double cw = kpo_cost(m, d, d->qacc_warmstart, NULL, NULL), cs = kpo_cost(m, d, d->qacc_smooth, NULL, NULL);
    memcpy(qacc, cw < cs ? d->qacc_warmstart : d->qacc_smooth, sizeof(double) * nv);
    memset(Mfull, 0, sizeof(double) * nv * nv);
    for (int i = 0; i < m->nv; i++) { int adr = m->dof_madr[i]; for (int j = i; j >= 0; j = m->dof_parent[j], adr++) Mfull[i * nv + j] = Mfull[j * nv + i] = d->qM[adr]; }
    for (int k = 0; k < d->nobj; k++) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Mfull[(m->nv + 6 * k + i) * nv + m->nv + 6 * k + j] = d->obj[k].M[6 * i + j];
    double scale = 1.0 / (m->meaninertia * (m->nv_full > 1 ? m->nv_full : 1));   /* scene-wide constants of the reference model */
    double cost = kpo_cost(m, d, qacc, grad, jar);
    for (int it = 0; it < m->solver_iter; it++) {
        memcpy(H, Mfull, sizeof(double) * nv * nv);
        for (int e = 0; e < ne; e++) if (jar[e] < 0) {
            double D = d->efc_D[e]; const double *J = d->efc_J[e];
            for (int i = 0; i < nv; i++) if (J[i] != 0) { FL(1 + 2 * (i + 1)); double a = D * J[i]; for (int j = 0; j <= i; j++) H[i * nv + j] += a * J[j]; }
        }
        for (int i = 0; i < nv; i++) for (int j = 0; j < i; j++) H[j * nv + i] = H[i * nv + j];
        if (chol_factor(H, nv)) break;
        for (int i = 0; i < nv; i++) search[i] = -grad[i];
        chol_solve(H, nv, search);
        /* line search on phi(a) = cost(qacc + a * search), a convex piecewise quadratic */
        kpo_mulM_full(m, d, search, Mv);
        FL(2.0 * ne * nv + 4 * nv);
        for (int e = 0; e < ne; e++) { double s = 0; for (int i = 0; i < nv; i++) s += d->efc_J[e][i] * search[i]; jv[e] = s; }
        double g0 = 0, h0 = 0; /* Gauss part: derivative at alpha: g0 + alpha*h0 */
        for (int i = 0; i < nv; i++) { g0 += search[i] * (grad[i] + d->qfrc_constraint[i]); h0 += search[i] * Mv[i]; }
        double alpha = 0;
        if (m->ls_exact) {          /* exact minimiser (what the HIP kernel computes): Newton on phi' to machine precision */
            for (int ls = 0; ls < 50; ls++) {
                double dphi = g0 + alpha * h0, ddphi = h0;
                for (int e = 0; e < ne; e++) { double x = jar[e] + alpha * jv[e]; if (x < 0) { dphi += d->efc_D[e] * x * jv[e]; ddphi += d->efc_D[e] * jv[e] * jv[e]; } }
                if (ddphi <= 0) break;
                double step = -dphi / ddphi;
                alpha += step;
                if (fabs(step) < 1e-14 * (1 + fabs(alpha))) break;
            }
        } else {
            kpo_ls_ctx ctx = {ne, g0, h0, jar, jv, d->efc_D, 0};
            double sn = 0; for (int i = 0; i < nv; i++) sn += search[i] * search[i];
            alpha = kpo_primal_search(&ctx, sqrt(sn), m->solver_tol * m->ls_tolerance / scale, m->ls_iterations);
        }
        if (alpha == 0) break;
        for (int i = 0; i < nv; i++) qacc[i] += alpha * search[i];
        double old = cost;
        cost = kpo_cost(m, d, qacc, grad, jar);
        d->solver_niter = it + 1;
        double gn = 0; for (int i = 0; i < nv; i++) gn += grad[i] * grad[i];
        if (scale * (old - cost) < m->solver_tol || scale * sqrt(gn) < m->solver_tol) break;
    }
    kpo_cost(m, d, qacc, NULL, NULL);
    memcpy(d->qacc, qacc, sizeof(double) * nv);
}

/* mj_forward  [MJ-ext] */
void kpo_forward(const kpo_model *m, kpo_data *d) {
    int nv = m->nv;
    kpo_kinematics(m, d);
    kpo_crb(m, d);
    kpo_factor_sparse(m, d->qM, d->qLD, d->qLDiagInv);
    kpo_obj_forward(m, d);
    kpo_collide(m, d);
    kpo_vel_bias(m, d);
    for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = -d->qfrc_bias[i] + d->qfrc_applied[i] + (i >= 6 ? d->ctrl[i - 6] : 0.0);
    memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(double) * nv);
    kpo_solve_sparse(m, d->qLD, d->qLDiagInv, d->qacc_smooth);
    for (int k = 0; k < d->nobj; k++) {
        double L[36], x[6]; memcpy(L, d->obj[k].M, sizeof(L));
        for (int i = 0; i < 6; i++) { x[i] = -d->obj[k].bias[i]; d->qfrc_smooth[nv + 6 * k + i] = x[i]; }
        chol_factor(L, 6); chol_solve(L, 6, x);
        memcpy(d->qacc_smooth + nv + 6 * k, x, 48);
    }
    kpo_make_constraint(m, d);
    kpo_solve_constraint(m, d);
}

/* mj_step = mj_forward + mj_Euler (no damping => explicit in velocity)  [MJ-ext] */
void kpo_step(const kpo_model *m, kpo_data *d) {
    double h = m->timestep;
    kpo_forward(m, d);
    memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * kpo_nvt(m, d));
    for (int k = 0; k < d->nobj; k++) {
        kpo_obj *o = &d->obj[k];
        for (int i = 0; i < 6; i++) o->qvel[i] += h * d->qacc[m->nv + 6 * k + i];
        for (int i = 0; i < 3; i++) o->qpos[i] += h * o->qvel[i];
        double *ow = o->qvel + 3, on = sqrt(v3_dot(ow, ow)), oq[4] = {1, 0, 0, 0}, onew[4];
        if (on >= MJ_MINVAL) { double ang = h * on, sn = sin(0.5 * ang) / on; oq[0] = cos(0.5 * ang); oq[1] = ow[0] * sn; oq[2] = ow[1] * sn; oq[3] = ow[2] * sn; }
        quat_normalize(o->qpos + 3); quat_mul(onew, o->qpos + 3, oq); memcpy(o->qpos + 3, onew, 32);
    }
    for (int i = 0; i < m->nv; i++) d->qvel[i] += h * d->qacc[i];
    for (int k = 0; k < 3; k++) d->qpos[k] += h * d->qvel[k];
    double *w = d->qvel + 3, n = sqrt(v3_dot(w, w));
    double qr[4] = {1, 0, 0, 0};
    if (n >= MJ_MINVAL) { double ang = h * n, s = sin(0.5 * ang) / n; qr[0] = cos(0.5 * ang); qr[1] = w[0] * s; qr[2] = w[1] * s; qr[3] = w[2] * s; }
    quat_normalize(d->qpos + 3);
    double qn[4]; quat_mul(qn, d->qpos + 3, qr); memcpy(d->qpos + 3, qn, 32);
    for (int j = 0; j < m->nu; j++) d->qpos[7 + j] += h * d->qvel[6 + j];
}

/* set_state + sim.forward()  (mujoco_env.py:97-103); sim.reset() zeroes ctrl / qfrc_applied / warmstart */
void kpo_reset(const kpo_model *m, kpo_data *d, const double *qpos, const double *qvel) {
    int ng = d->ngeom_static, no = d->nobj; kpo_geom gsave[MAXGEOM]; memcpy(gsave, d->geom, sizeof(gsave));
    static kpo_obj osave[MAXOBJ]; memcpy(osave, d->obj, sizeof(osave));
    memset(d, 0, sizeof(*d));
    d->ngeom = d->ngeom_static = ng; memcpy(d->geom, gsave, sizeof(gsave));
    d->nobj = no; memcpy(d->obj, osave, sizeof(osave));
    memcpy(d->qpos, qpos, sizeof(double) * m->nq); memcpy(d->qvel, qvel, sizeof(double) * m->nv);
    kpo_forward(m, d);
}

/* ------------------------------------------------------------------ reference controller  [REF] */
/* compute_desired_accel (humanoid_im.py:418-431): dense Cholesky of M + K_d dt, as scipy cho_factor */
static void kpo_desired_accel(const kpo_model *m, const kpo_data *d, const double *qpos_err, const double *qvel_err,
                              const double *k_p, const double *k_d, double *q_accel) {
    double A[NV_MAX * NV_MAX];
    int nv = m->nv; double dt = m->timestep;
    kpo_fullM(m, d, A);
    for (int i = 0; i < nv; i++) A[i * nv + i] += k_d[i] * dt;
    FL(6 * nv);
    for (int i = 0; i < nv; i++) q_accel[i] = -d->qfrc_bias[i] - k_p[i] * qpos_err[i] - k_d[i] * qvel_err[i];
    chol_factor(A, nv); chol_solve(A, nv, q_accel);
}
/* compute_torque (humanoid_im.py:433-480), action_v=1, no meta_pd */
void kpo_compute_torque(const kpo_model *m, const kpo_data *d, const double *ctrl, const double *target_qpos, double *torque) {
    int nv = m->nv, nu = m->nu; double dt = m->timestep;
    double k_p[NV_MAX] = {0}, k_d[NV_MAX] = {0}, qpos_err[NV_MAX] = {0}, qvel_err[NV_MAX], q_accel[NV_MAX];
    for (int j = 0; j < nu; j++) {
        double base = target_qpos[7 + j], q = d->qpos[7 + j];
        /* the reference's unwrap loops (humanoid_im.py:447-452); a state that has blown up (|base - q| beyond 1e6, where subtracting
         * 2 pi no longer changes a double's neighbourhood) would spin them forever: such a state is garbage either way, leave it */
        if (fabs(base - q) < 1e6) {
            while (base - q > M_PI) base -= 2 * M_PI;
            while (base - q < -M_PI) base += 2 * M_PI;
        }
        double target = base + ctrl[j] * m->a_scale[j];
        k_p[6 + j] = m->kp[j]; k_d[6 + j] = m->kd[j];
        qpos_err[6 + j] = q + d->qvel[6 + j] * dt - target;
    }
    memcpy(qvel_err, d->qvel, sizeof(double) * nv);
    kpo_desired_accel(m, d, qpos_err, qvel_err, k_p, k_d, q_accel);
    for (int i = 0; i < nv; i++) qvel_err[i] += q_accel[i] * dt;
    for (int j = 0; j < nu; j++) torque[j] = -m->kp[j] * qpos_err[6 + j] - m->kd[j] * qvel_err[6 + j];
}
/* rfc_implicit (humanoid_im.py:497-504) */
void kpo_rfc_implicit(const kpo_model *m, kpo_data *d, const double *vf_in) {
    double vf[6]; for (int k = 0; k < 6; k++) vf[k] = vf_in[k] * m->rfc_scale;
    /* remove_base_rot: q (x) inverse(base_rot), inverse = conj / dot */
    const double *br = m->base_rot; double nn = br[0] * br[0] + br[1] * br[1] + br[2] * br[2] + br[3] * br[3];
    double binv[4] = {br[0] / nn, -br[1] / nn, -br[2] / nn, -br[3] / nn}, cq[4];
    quat_mul(cq, d->qpos + 3, binv);
    double hq[4] = {cq[0], 0, 0, cq[3]}; double hn = sqrt(hq[0] * hq[0] + hq[3] * hq[3]); hq[0] /= hn; hq[3] /= hn;
    double R[9]; /* quaternion_matrix(hq): normalises by dot */
    { double q[4] = {hq[0], hq[1], hq[2], hq[3]}; quat_normalize(q); quat2mat(R, q); }
    double f[3]; mat_mulvec(f, R, vf); vf[0] = f[0]; vf[1] = f[1]; vf[2] = f[2];
    for (int k = 0; k < 6; k++) { double v = vf[k]; if (v > m->rfc_lim) v = m->rfc_lim; if (v < -m->rfc_lim) v = -m->rfc_lim; d->qfrc_applied[k] = v; }
}
/* do_simulation (humanoid_im.py:506-533): n_frames x {compute_torque, clip, rfc_implicit, sim.step} */
void kpo_do_simulation(const kpo_model *m, kpo_data *d, const double *action, const double *target_qpos, int n_frames) {
    for (int i = 0; i < n_frames; i++) {
        double torque[69];
        kpo_compute_torque(m, d, action, target_qpos, torque);
        for (int j = 0; j < m->nu; j++) { double t = torque[j]; if (t > m->torque_lim[j]) t = m->torque_lim[j]; if (t < -m->torque_lim[j]) t = -m->torque_lim[j]; d->ctrl[j] = t; }
        kpo_rfc_implicit(m, d, action + m->nu);
        kpo_step(m, d);
    }
}

/* n x 17 doubles: type (0 box, 1 cylinder), size[3], pos[3], mat[9] (world), invweight of the owning object */
void kpo_set_geoms(kpo_data *d, int n, const double *p) {
    d->ngeom = d->ngeom_static = n > MAXGEOM ? MAXGEOM : n;
    for (int i = 0; i < d->ngeom; i++, p += 17) {
        kpo_geom *g = &d->geom[i]; g->obj = -1;
        g->type = (int)p[0]; memcpy(g->size, p + 1, 24); memcpy(g->pos, p + 4, 24); memcpy(g->mat, p + 7, 72); g->invw = p[16];
        g->rbound = g->type == 0 ? sqrt(v3_dot(g->size, g->size)) : sqrt(g->size[0] * g->size[0] + g->size[1] * g->size[1]);
    }
}

/* dynamic objects: inertial[13] = mass, com[3], inertia[6], invweight (tran, rot), armature; geoms [ng][16] body frame =
 * type, size[3], pos[3], mat[9]; state qpos[7], qvel[6].  kpo_set_objects(d, 0, ..) removes them. */
void kpo_set_object(kpo_data *d, int k, const double *inertial, int ng, const double *geoms, const double *qpos7, const double *qvel6) {
    if (k < 0 || k >= MAXOBJ) return;
    kpo_obj *o = &d->obj[k];
    memset(o, 0, sizeof(*o));
    o->mass = inertial[0]; memcpy(o->ipos, inertial + 1, 24); memcpy(o->inertia, inertial + 4, 48); o->invw[0] = inertial[10]; o->invw[1] = inertial[11]; o->arm = inertial[12];
    o->ngeom = ng > MAXGEOM ? MAXGEOM : ng;
    for (int i = 0; i < o->ngeom; i++, geoms += 16) {
        kpo_geom *g = &o->lgeom[i];
        g->type = (int)geoms[0]; memcpy(g->size, geoms + 1, 24); memcpy(g->pos, geoms + 4, 24); memcpy(g->mat, geoms + 7, 72);
        g->rbound = g->type == 0 ? sqrt(v3_dot(g->size, g->size)) : sqrt(g->size[0] * g->size[0] + g->size[1] * g->size[1]);
    }
    memcpy(o->qpos, qpos7, 56); memcpy(o->qvel, qvel6, 48);
    if (k + 1 > d->nobj) d->nobj = k + 1;
}
void kpo_clear_objects(kpo_data *d) { d->nobj = 0; d->ngeom = d->ngeom_static; }
void kpo_get_object(const kpo_data *d, int k, double *qpos7, double *qvel6) { memcpy(qpos7, d->obj[k].qpos, 56); memcpy(qvel6, d->obj[k].qvel, 48); }
void kpo_get_object_dyn(const kpo_data *d, int k, double *M36, double *bias6) { memcpy(M36, d->obj[k].M, 288); memcpy(bias6, d->obj[k].bias, 48); }
void kpo_get_qacc_full(const kpo_data *d, double *out) { memcpy(out, d->qacc, sizeof(double) * NVT_MAX); }
void kpo_get_qacc_smooth_full(const kpo_data *d, double *out) { memcpy(out, d->qacc_smooth, sizeof(double) * NVT_MAX); }
void kpo_get_contact_pairs(const kpo_data *d, int *b1, int *b2) { for (int c = 0; c < d->ncon; c++) { b1[c] = d->con_body[c]; b2[c] = d->con_b2[c]; } }

/* ------------------------------------------------------------------ accessors for ctypes */
#define GETTER(name, field, n) void kpo_get_##name(const kpo_data *d, double *out) { memcpy(out, d->field, sizeof(double) * (n)); }
GETTER(qpos, qpos, NQ_MAX) GETTER(qvel, qvel, NV_MAX) GETTER(xpos, xpos, 72) GETTER(xquat, xquat, 96) GETTER(xipos, xipos, 72)
GETTER(qM, qM, NM_MAX) GETTER(qfrc_bias, qfrc_bias, NV_MAX) GETTER(qacc, qacc, NV_MAX) GETTER(qacc_smooth, qacc_smooth, NV_MAX)
GETTER(subtree_com, subtree_com, 3) GETTER(ctrl, ctrl, 69) GETTER(qfrc_applied, qfrc_applied, NV_MAX) GETTER(qfrc_constraint, qfrc_constraint, NV_MAX)
GETTER(cvel, cvel, 144)
void kpo_get_efc(const kpo_data *d, double *force, double *D, double *aref) { memcpy(force, d->efc_force, 8 * d->nefc); memcpy(D, d->efc_D, 8 * d->nefc); memcpy(aref, d->efc_aref, 8 * d->nefc); }
void kpo_get_efc_J(const kpo_data *d, double *J) { for (int e = 0; e < d->nefc; e++) memcpy(J + (size_t)e * NV_MAX, d->efc_J[e], 8 * NV_MAX); }
void kpo_get_efc_J_full(const kpo_data *d, double *J) { for (int e = 0; e < d->nefc; e++) memcpy(J + (size_t)e * NVT_MAX, d->efc_J[e], 8 * NVT_MAX); }
int kpo_get_ncon(const kpo_data *d) { return d->ncon; }
int kpo_get_nefc(const kpo_data *d) { return d->nefc; }
int kpo_get_niter(const kpo_data *d) { return d->solver_niter; }
void kpo_get_contacts(const kpo_data *d, int *body, double *pos, double *dist) {
    for (int c = 0; c < d->ncon; c++) { body[c] = d->con_body[c]; memcpy(pos + 3 * c, d->con_pos[c], 24); dist[c] = d->con_dist[c]; }
}
void kpo_get_contact_normals(const kpo_data *d, double *n) { for (int c = 0; c < d->ncon; c++) memcpy(n + 3 * c, d->con_frame[c], 24); }
void kpo_set_qpos_qvel(kpo_data *d, const double *qpos, const double *qvel) { memcpy(d->qpos, qpos, 8 * NQ_MAX); memcpy(d->qvel, qvel, 8 * NV_MAX); }
void kpo_set_ctrl(kpo_data *d, const double *ctrl, const double *applied6) { memcpy(d->ctrl, ctrl, 8 * 69); if (applied6) memcpy(d->qfrc_applied, applied6, 48); }
void kpo_solveM(const kpo_model *m, const kpo_data *d, double *x) { kpo_solve_sparse(m, d->qLD, d->qLDiagInv, x); }

/* batch driver for the CPU baseline: n envs, each: n_steps x do_simulation(15) with fixed action/target */
void kpo_rollout_batch(const kpo_model *m, int n, double *qpos, double *qvel, const double *action, const double *target, int n_steps, int n_frames) {
    kpo_data *d = kpo_data_new();
    for (int e = 0; e < n; e++) {
        kpo_reset(m, d, qpos + (size_t)e * NQ_MAX, qvel + (size_t)e * NV_MAX);
        for (int s = 0; s < n_steps; s++) kpo_do_simulation(m, d, action + (size_t)e * 75, target + (size_t)e * NQ_MAX, n_frames);
        memcpy(qpos + (size_t)e * NQ_MAX, d->qpos, 8 * NQ_MAX); memcpy(qvel + (size_t)e * NV_MAX, d->qvel, 8 * NV_MAX);
    }
    kpo_data_free(d);
}

/* ------------------------------------------------------------------ narrow-phase test hooks (tests/test_collide_oracle.py)
 * shape record: [type, size3 (hull: the MPR centre), pos3, mat9] = 16 doubles; hull vertices / graph passed separately.
 * kind: 0 kpo_convex(a, b)  1 kpo_plane_box(a)  2 kpo_plane_cylinder(a)  3 kpo_box_box(a, b)  4 kpo_plane_mesh(b).
 * out [n][7] = dist, pos3, normal3 (geom 1 -> geom 2).  Returns the number of contacts. */
static void kpo_shape_from_record(const double *r, const double *verts, int nvert, kpc_shape *s) {
    s->type = (int)r[0]; memcpy(s->size, r + 1, 24); memcpy(s->pos, r + 4, 24); memcpy(s->mat, r + 7, 72);
    if (s->type == 2) { memcpy(s->center, r + 1, 24); s->verts = verts; s->nvert = nvert; }
    else { memcpy(s->center, s->pos, 24); s->verts = NULL; s->nvert = 0; }
}
int kpo_narrowphase(int kind, const double *a, const double *verts_a, int nvert_a, const double *b, const double *verts_b, int nvert_b,
                    const int *nbr_adr, const int *nbr, double margin, double tol_rbound, int maxcon, double *out) {
    kpc_shape sa, sb; kpc_contact con[KPC_MAXPAIR];
    int n = 0;
    if (a) kpo_shape_from_record(a, verts_a, nvert_a, &sa);
    if (b) kpo_shape_from_record(b, verts_b, nvert_b, &sb);
    if (kind == 0) n = kpo_convex(&sa, &sb, margin, con);
    else if (kind == 1) n = kpo_plane_box(&sa, margin, con);
    else if (kind == 2) n = kpo_plane_cylinder(&sa, margin, con);
    else if (kind == 3) n = kpo_box_box(&sa, &sb, margin, con);
    else if (kind == 4) n = kpo_plane_mesh(&sb, nbr_adr, nbr, margin, tol_rbound, maxcon, con);
    for (int i = 0; i < n; i++) { out[7 * i] = con[i].dist; memcpy(out + 7 * i + 1, con[i].pos, 24); memcpy(out + 7 * i + 4, con[i].normal, 24); }
    return n;
}
