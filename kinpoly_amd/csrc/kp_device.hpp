// kp_device.hpp -- device-side math + the per-env LDS layout of the fused substep kernel (gfx950).
//
// One workgroup = one environment.  All per-env dynamics state lives in LDS for the whole control
// step (15 substeps); HBM is touched only at kernel entry (qpos/qvel/action/target) and exit
// (qpos/qvel + stale kinematics for the observation kernels).  Lanes map to bodies (24), dofs (75),
// hull vertices (<=64 per hull) or contacts depending on the phase.  Three layouts (below): the full one
// (8 envs in the 160 KiB LDS of a CU), the lean one of the floor scenes' job queue (12 envs = three waves on
// every SIMD) and the one with the free objects' block (7 envs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kp {

constexpr int D_NB = 24, D_NV = 75, D_NQ = 76, D_NU = 69;
constexpr int D_MAXCON = 64;          // must equal MAXCON in oracle/kp_oracle.c
constexpr int D_NLEV = 9;             // body tree depth levels (Pelvis .. Hand)

struct DevTables {
    const float *body_pos, *body_ipos, *body_mass, *body_inertia, *body_rbound, *body_invw;
    const float *mesh_rbound;       // mjModel.geom_rbound of every hull's mesh geom (mjc_PlaneConvex's tolerance scale)
    const float *dof_armature, *jnt_lo, *jnt_hi, *lim_invw;
    const float *kp, *kd, *tlim, *ascale;
    const float *verts;
    const uint16_t *vert_adr;
    const uint16_t *vert_nbr_adr;   // hull graph: neighbours (hull-local ids) of global vertex v are vert_nbr[vert_nbr_adr[v] .. vert_nbr_adr[v + 1])
    const uint8_t *vert_nbr;
    const uint8_t *dof_body;
    const int8_t *body_parent;
    const uint8_t *body_depth, *body_subtree, *lev_start, *lev_body, *jnt_limited;
    const uint32_t *sched8;   // [64][9] static tree schedule of the 8-lanes-per-body passes (see Lane8)
    // free objects of the scene (chair, box, table, Can, step): [n_obj][13] mass, com[3], inertia[6], invweight (tran, rot),
    // armature; body-frame geoms [n_obj_geoms][18] = object, type, size[3], pos[3], mat[9], mass; geom range per object
    const float *obj_inertial, *obj_geoms;
    const int *obj_geom_adr;
    int n_obj;
};

struct Params {
    float h, gx, gy, gz;
    float K, B;                                 // solref -> stiffness / damping of the reference acceleration
    float imp_d0, imp_dw, imp_w, imp_mid, imp_pow;
    float mu, margin;
    int pm_max; float pm_tol;                   // mjc_PlaneConvex: maxplanemesh (3), tolplanemesh (0.3) [MJ-ext]; model options planemesh_max / planemesh_tol
    float scale;                                // 1 / (meaninertia * nv)
    float rfc_scale, rfc_lim;
    float br_inv[4];                            // inverse(base_rot) as the reference computes it (conj / |q|^2)
    float tol;
    int max_iter, contact, limits, stale, actuation;
};

// ------------------------------------------------------------------ LDS layouts
// EnvLds      the full layout, 18 128 B (15 allocation granules of 1 280 B; 8 envs per CU = 2 waves per SIMD): every launch form except the floor scenes' job queue
// EnvLdsLean  the floor scenes' job-queue kernel: 3 waves on every SIMD need <= 12 800 B per env (the free-fall instantiation at 12 instead of
//             8 envs per CU is 18 - 19 % faster).  Same arithmetic, same code; what differs is where a few vectors live (see the struct) and the number of
//             contacts the block holds (24: an env that needs more is handed to the full-layout kernel, kp_step_overflow_kernel)
// EnvLdsObj   EnvLds + the free objects' block
struct __attribute__((aligned(16))) EnvLds {
    static constexpr bool LEAN = false;
    static constexpr int MAXCON = D_MAXCON;   // contacts the block holds
    float qpos[76], qvel[76];             // PD targets and actions are read from their HBM rows once per substep (spd_torque_rfc)
    float xpos[72], xquat[96];            // body COMs (xipos) are recomputed where they are read: collision centres, the read-out
    float cinert[240];                    // body spatial inertia about o, world axes (10 floats / body)
    float cdof[450];                      // motion axis of every dof [ang; lin] about o
    float sv[156];                        // per-body spatial scratch (velocity / acceleration); sv[144..155]: the two object slots
    float U[450], Dinv[76], uj[76];       // articulated-body pass: U_j = IA s_j, 1/D_j, u_j
    float IAa[25 * 22], pAa[25 * 6];      // articulated inertia / bias force handed to the parent; slot 21 of a record and
                                          // record 24 are kept 0 so that padded / absent operands load a zero without exec masking
    float arm[76];                        // dof armature
    float fb[144];                        // bias wrench of every body (gyroscopic + Coriolis + gravity), about o: enters the ABA passes as pA
    float qacc_s[76], qacc[76], search[76], Mv[76], mres[76], x[76], extra[76];
    float applied_pad[2], applied[6], ctrl[72];   // applied ++ ctrl is qfrc_applied + qfrc_actuator as one 76-vector: written by spd_torque_rfc, read by
                                          // every gradient of the Newton solve (the gradient itself lives in the words of qacc_s)
    float con_pos[D_MAXCON * 3], con_D[D_MAXCON];   // con_D: contact distance from collide() until make_constraint() turns it into the row weight D
    float jar3[D_MAXCON * 3];             // contact-frame (normal, t1, t2) residuals J qacc - aref
    float lim_D[72], lim_jar[72];         // joint-limit rows: lim_D = sign x weight (sign: +1 lower / -1 upper limit violated, 0 = no row)
    // jv3 | lim_jv | sa | sw are contiguous: between the gradient and the row evaluation of a Newton iteration all four are dead, and the
    // object kernel's batched Schur-complement columns use the 552 floats as one scratch block (schur_columns)
    float jv3[D_MAXCON * 3];              // J search per contact (and aref until the first J search product)
    float lim_jv[72];                     // J search per joint-limit row; holds the reference acceleration until the first J search product overwrites it
    float sa[144], sw[144];               // per-body spatial scratch (acceleration / wrench); contact forces of the object solve live in sa ++ sw[0, 48)
    float red[8];
    unsigned char bpar[D_NB], bsub[D_NB], bdep[D_NB], dbody[76];
    unsigned char con_act[D_MAXCON];      // active pyramid rows (4 bits) of every contact at the last factorisation
    unsigned char con_body[D_MAXCON];     // entity carrying the vertex: 0..23 hull, 24 + k object slot k
    unsigned char con_start[D_NB + 4];    // contacts are grouped by that entity: 24 hulls, then the object slots (values <= D_MAXCON)
    int ncon, nlim, flag;
};

// The floor scenes' job-queue layout.  Differences from EnvLds, each with the reason it is safe (phases of a substep, in order: spd_torque_rfc, forward_kin_bias,
// collide, make_constraint, Newton solve = { gradient (wrench_project), factorisation / solve (aba_solve | aba_resolve), row evaluation, line search, update }, Euler):
//   * MAXCON = 24 contacts (con_pos, con_D, jar3, jv3, con_act, con_body): floor scenes hold 7 - 10 on average, a humanoid lying flat 12 (mjc_PlaneConvex keeps
//     at most 3 per hull and drops neighbours within 0.3 rbound); collide() reports an env that needs more and the job is re-run by kp_step_overflow_kernel on
//     the full layout (no state of the job has reached HBM at that point);
//   * ONE vector for the Newton step's gradient, right-hand side, joint-space bias u_j and search direction (search = x = qacc_s = uj): the gradient is negated in
//     place, the leaves->root pass replaces x_d by u_d (read and written by the 8 lanes that own dof d, in that order), the root->leaves pass replaces u_d by
//     the solution.  The object solver re-uses x after the solve (back-substitution) and keeps the four apart;
//   * sa | sw (the gradient's body wrenches and their subtree sums, forward_kin_bias' velocity-product accelerations) share their words with jv3 | lim_jv | search:
//     J search and the search direction are dead while a gradient is formed (its result lands in the words of sw, which is dead by then), sa is dead outside;
//   * the Newton solve's body accelerations of the iterate (sacc, 144 floats) live in the words of qpos | qvel, which nothing reads between make_constraint and the
//     Euler step: step_body restores both from the registers that hold the forward pass' state anyway (qd_save_q / qd_save_v) when the solve returns;
//   * the contact residuals jar3 live in the words of xquat, dead once collide() has run: the read-outs (xpos, xquat, xipos of the control step's last forward
//     pass) are stored right after the last substep's collision pass instead of at the job's end;
//   * the bias forces handed up the tree (pAa) live in the words of jv3 | lim_jv, dead during every factorisation / solve (the object kernel's Schur columns use
//     the same gap); their zero record is re-written at the top of a solve;
//   * no copy of the dof armature: the eliminations read it folded into `extra` (same sum, formed once), the two products with M read the model table;
//   * the stable-PD position error rides in lim_jar (dead outside the Newton solve), a_{k-2} of warm_extrap in the env's HBM row (kp_sim: warm2);
//   * sv without the two object slots.
struct __attribute__((aligned(16))) EnvLdsLean {
    static constexpr bool LEAN = true;
    static constexpr int MAXCON = 24;
    float qpos[76], qvel[76];
    float xpos[72];
    union { float xquat[96]; float jar3[MAXCON * 3]; };
    float cinert[240];
    float cdof[450];
    float sv[144];
    float U[450], Dinv[76];
    float IAa[25 * 22];
    float fb[144];
    float qacc[76], extra[76];            // extra: dof armature + the solve's extra armature (K_d h, active joint limits): the sum the eliminations read
    float applied_pad[2], applied[6], ctrl[72];
    float con_pos[MAXCON * 3], con_D[MAXCON];
    float lim_D[D_NU], lim_jar[D_NU];
    union {
        struct { union { struct { float jv3[MAXCON * 3], lim_jv[72]; }; float pAa[25 * 6]; }; union { float search[76], x[76], qacc_s[76], uj[76]; }; };
        struct { float sa[144], sw[144]; };
    };
    float red[8];
    unsigned char bpar[D_NB], bsub[D_NB], bdep[D_NB], dbody[76];
    unsigned char con_act[MAXCON], con_body[MAXCON];
    unsigned char con_start[D_NB + 4];
    int ncon, nlim, flag;
};
static_assert(offsetof(EnvLdsLean, search) >= offsetof(EnvLdsLean, sw) && offsetof(EnvLdsLean, search) + 76 * sizeof(float) <= offsetof(EnvLdsLean, sw) + 144 * sizeof(float),
              "the search direction must lie inside sw (wrench_project writes the gradient there once sw is dead) and clear of sa");
static_assert(offsetof(EnvLdsLean, U) % 8 == 0, "s.U must be 8-byte aligned");

// extension used only by the kernel instantiation that simulates object contact (kp_step_kernel<NT, true>)
constexpr int D_MAXGEOM = 8;            // must equal MAXGEOM in oracle/kp_oracle.c
constexpr int D_MAXOBJ = 2;             // must equal MAXOBJ in oracle/kp_oracle.c
struct __attribute__((aligned(16))) EnvLdsObj : EnvLds {
    float con_n[D_MAXCON * 3];          // contact normal (world), pointing from the surface (floor / geom) into the vertex' entity
    float geom[D_MAXGEOM * 17];         // type, size[3], pos[3], mat[9], invweight  (world frame)
    int ngeom, ngeom_static, nobj;
    // ---- dynamic free objects (slot k = entity 24 + k): spatial quantities about o like everything else
    signed char gobj[D_MAXGEOM];        // slot owning world geom g (-1: static)
    signed char con_b2[D_MAXCON];       // entity carrying the surface: -1 the floor, -2 - g static geom g (its invweight0 is geom[17 g + 16]), 24 + k
    unsigned char ggi[D_MAXGEOM];       // model geom (row of DevTables::obj_geoms: the body-frame geom) behind world geom g of a dynamic object
    float oq[D_MAXOBJ * 7], ov[D_MAXOBJ * 6], oc[D_MAXOBJ * 13];
    float oR[D_MAXOBJ * 9];
    float oI[D_MAXOBJ * 10], oIe[D_MAXOBJ * 10];   // spatial inertia (true / with the free-joint armature)
    float ofb[D_MAXOBJ * 6];            // bias wrench
    float oqa[D_MAXOBJ * 6];            // joint-space acceleration of the last solve (warm start, integration)
    float oqa_prev[D_MAXOBJ * 6];       // the solve before it (extrapolated start, warm_extrap)
    float oas[D_MAXOBJ * 6], oa[D_MAXOBJ * 6], omres[D_MAXOBJ * 6], osrch[D_MAXOBJ * 6], oMv[D_MAXOBJ * 6], ogr[D_MAXOBJ * 6], ot[D_MAXOBJ * 6];
    float Sm[6 * D_MAXOBJ * (6 * D_MAXOBJ + 1)];   // dense object system [n][n + 1] (last column: right-hand side)
    float cM[D_MAXCON * 6];             // per contact: D F G F^T of the active pyramid rows (world, xx yy zz xy xz yz)
    float red[4 * 21];                  // row partials of the wave-wide vector sums (lane = contact)
};

// ------------------------------------------------------------------ small math
struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
    return Q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
    float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    if (n < 1e-15f) return Q4{1.f, 0.f, 0.f, 0.f};
    float r = 1.0f / n;
    return Q4{q.w * r, q.x * r, q.y * r, q.z * r};
}
__device__ __forceinline__ void q2mat(Q4 q, float* m) {
    float w = q.w, x = q.x, y = q.y, z = q.z;
    m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
    m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
    m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {  // R(q) v for a unit quaternion
    V3 u = v3(q.x, q.y, q.z);
    V3 t = 2.0f * cross(u, v);
    return v + q.w * t + cross(u, t);
}
__device__ __forceinline__ V3 mulmat(const float* m, V3 v) {
    return V3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
// spatial vectors are [ang(3); lin(3)] about the env's reference point o = root body origin
struct S6 { V3 a, l; };
__device__ __forceinline__ S6 lds6(const float* p) { return S6{ld3(p), ld3(p + 3)}; }
__device__ __forceinline__ void sts6(float* p, S6 s) { st3(p, s.a); st3(p + 3, s.l); }
__device__ __forceinline__ S6 operator+(S6 a, S6 b) { return S6{a.a + b.a, a.l + b.l}; }
__device__ __forceinline__ S6 operator*(float s, S6 a) { return S6{s * a.a, s * a.l}; }
__device__ __forceinline__ float dot6(S6 a, S6 b) { return dot(a.a, b.a) + dot(a.l, b.l); }
__device__ __forceinline__ S6 cross_motion(S6 v, S6 s) { return S6{cross(v.a, s.a), cross(v.a, s.l) + cross(v.l, s.a)}; }
__device__ __forceinline__ S6 cross_force(S6 v, S6 f) { return S6{cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; }
// inert = [Ixx Iyy Izz Ixy Ixz Iyz | h(3) = m r | m]
__device__ __forceinline__ S6 inert_mul(const float* I, S6 v) {
    V3 h = ld3(I + 6);
    float m = I[9];
    V3 Iw = V3{I[0] * v.a.x + I[3] * v.a.y + I[4] * v.a.z, I[3] * v.a.x + I[1] * v.a.y + I[5] * v.a.z,
               I[4] * v.a.x + I[5] * v.a.y + I[2] * v.a.z};
    return S6{Iw + cross(h, v.l), m * v.l - cross(h, v.a)};
}

__device__ __forceinline__ float impedance(const Params& P, float pos) {
    float x = fabsf(pos) / P.imp_w;
    if (x >= 1.0f) return P.imp_dw;
    if (x <= 0.0f) return P.imp_d0;
    float y;
    if (P.imp_pow == 2.0f) y = x <= P.imp_mid ? x * x / P.imp_mid : 1.0f - (1.0f - x) * (1.0f - x) / (1.0f - P.imp_mid);
    else if (P.imp_pow == 1.0f) y = x;
    else y = x <= P.imp_mid ? powf(x, P.imp_pow) / powf(P.imp_mid, P.imp_pow - 1.0f)
                            : 1.0f - powf(1.0f - x, P.imp_pow) / powf(1.0f - P.imp_mid, P.imp_pow - 1.0f);
    return P.imp_d0 + y * (P.imp_dw - P.imp_d0);
}

}  // namespace kp
