// kp_step_kernel.hpp -- the fused control-step kernel: n_substeps x { stable-PD torque, residual
// force, mj_step-equivalent forward dynamics with hull-plane soft contact, semi-implicit Euler }.
//
// Replaces HOT LOOP C of the reference: HumanoidEnv.do_simulation (uhc/envs/humanoid_im.py:506-533)
//   compute_torque / compute_desired_accel   uhc/envs/humanoid_im.py:418-480
//   rfc_implicit                              uhc/envs/humanoid_im.py:497-504
//   sim.step()  (MuJoCo mj_step)              uhc/envs/humanoid_im.py:527         [MJ-ext]
// The arithmetic mirrors oracle/kp_oracle.c (fp64) in fp32; what differs is the organisation:
//   * spatial quantities are taken about the root body origin (not the subtree COM) so that
//     kinematics, velocities and bias forces come out of ONE level-synchronous tree pass;
//   * J v, J^T f and J^T D J are never materialised: contact rows act on bodies, so J v is a
//     spatial-velocity tree pass, J^T f a wrench subtree sum, and the Newton Hessian M + J^T D J is a
//     composite-inertia pass with a 6x6 "contact inertia" per body (same sparsity as M);
//   * the tree-sparse L^T D L factorisation runs rank-1 updates with lanes over ancestor pairs.
#pragma once
#include "kp_device.hpp"

namespace kp {

#define KP_SYNC() __syncthreads()

struct StepArgs {
    DevTables T;
    Params P;
    int n_envs, n_substeps;
    // per-env state (HBM, row-major [N, dim])
    float *qpos, *qvel, *qpos_d, *qvel_d, *warm;
    const float *target_qpos, *action;
    const uint8_t* env_mask;  // optional: envs with mask==0 are skipped
    // outputs: kinematics of the last forward pass (x_14 in stale mode)
    float *xpos, *xquat, *xipos;
    int* diag;  // [N,4]: ncon (last substep), newton iterations (sum), flags, max ncon
};

template <int NT>
__device__ __forceinline__ float block_sum(EnvLds& s, float v, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (NT > 64) {
        KP_SYNC();
        if ((tid & 63) == 0) s.red[tid >> 6] = v;
        KP_SYNC();
        v = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) v += s.red[w];
    }
    return v;
}

// ---------------------------------------------------------------- kinematics + velocities + bias (one tree pass)
template <int NT>
__device__ void forward_kin_bias(EnvLds& s, const DevTables& T, const Params& P, int tid) {
    int depth = tid < D_NB ? T.body_depth[tid] : -1;
    for (int lev = 0; lev < D_NLEV; lev++) {
        if (depth == lev) {
            const int b = tid;
            S6 cv, ca;
            V3 pos, o;
            Q4 q;
            if (b == 0) {
                q = qnormalize(Q4{s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]});
                s.qpos[3] = q.w; s.qpos[4] = q.x; s.qpos[5] = q.y; s.qpos[6] = q.z;
                pos = ld3(s.qpos); o = pos;
                float R[9]; q2mat(q, R);
                V3 vl = ld3(s.qvel), wb = ld3(s.qvel + 3);
                V3 ww = mulmat(R, wb);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float* c = s.cdof + 6 * k;
                    c[0] = c[1] = c[2] = 0.f; c[3] = k == 0; c[4] = k == 1; c[5] = k == 2;
                    float* r = s.cdof + 6 * (3 + k);
                    r[0] = R[k]; r[1] = R[3 + k]; r[2] = R[6 + k]; r[3] = r[4] = r[5] = 0.f;
                }
                cv = S6{ww, vl};
                ca = S6{v3(0.f, 0.f, 0.f), v3(-P.gx, -P.gy, -P.gz) + cross(vl, ww)};
            } else {
                const int p = T.body_parent[b];
                o = ld3(s.xpos);
                pos = ld3(s.xpos + 3 * p) + mulmat(s.xmat + 9 * p, ld3(T.body_pos + 3 * b));
                q = Q4{s.xquat[4 * p], s.xquat[4 * p + 1], s.xquat[4 * p + 2], s.xquat[4 * p + 3]};
                cv = lds6(s.sv + 6 * p); ca = lds6(s.sa + 6 * p);
                V3 r = o - pos;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int d = 6 + 3 * (b - 1) + j;
                    V3 e = j == 0 ? v3(0.f, 0.f, 1.f) : (j == 1 ? v3(0.f, 1.f, 0.f) : v3(1.f, 0.f, 0.f));
                    V3 axis = qrot(q, e);
                    S6 cd = S6{axis, cross(axis, r)};
                    sts6(s.cdof + 6 * d, cd);
                    S6 cdd = cross_motion(cv, cd);
                    float qd = s.qvel[d];
                    cv = cv + qd * cd; ca = ca + qd * cdd;
                    float sn, cs; sincosf(0.5f * s.qpos[d + 1], &sn, &cs);
                    q = qmul(q, Q4{cs, e.x * sn, e.y * sn, e.z * sn});
                }
                q = qnormalize(q);
            }
            float R[9]; q2mat(q, R);
            st3(s.xpos + 3 * b, pos);
            s.xquat[4 * b] = q.w; s.xquat[4 * b + 1] = q.x; s.xquat[4 * b + 2] = q.y; s.xquat[4 * b + 3] = q.z;
#pragma unroll
            for (int k = 0; k < 9; k++) s.xmat[9 * b + k] = R[k];
            V3 xi = pos + mulmat(R, ld3(T.body_ipos + 3 * b));
            st3(s.xipos + 3 * b, xi);
            // inertia about o in world axes
            const float* Ib = T.body_inertia + 6 * b;
            float I3[9] = {Ib[0], Ib[3], Ib[4], Ib[3], Ib[1], Ib[5], Ib[4], Ib[5], Ib[2]};
            float Tm[9], W[9];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) Tm[3 * i + j] = R[3 * i] * I3[j] + R[3 * i + 1] * I3[3 + j] + R[3 * i + 2] * I3[6 + j];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) W[3 * i + j] = Tm[3 * i] * R[3 * j] + Tm[3 * i + 1] * R[3 * j + 1] + Tm[3 * i + 2] * R[3 * j + 2];
            V3 rr = xi - o;
            float mass = T.body_mass[b], r2 = dot(rr, rr);
            float* ci = s.cinert + 10 * b;
            ci[0] = W[0] + mass * (r2 - rr.x * rr.x); ci[1] = W[4] + mass * (r2 - rr.y * rr.y); ci[2] = W[8] + mass * (r2 - rr.z * rr.z);
            ci[3] = W[1] - mass * rr.x * rr.y; ci[4] = W[2] - mass * rr.x * rr.z; ci[5] = W[5] - mass * rr.y * rr.z;
            ci[6] = mass * rr.x; ci[7] = mass * rr.y; ci[8] = mass * rr.z; ci[9] = mass;
            sts6(s.sv + 6 * b, cv); sts6(s.sa + 6 * b, ca);
            S6 f = inert_mul(ci, ca) + cross_force(cv, inert_mul(ci, cv));
            sts6(s.sw + 6 * b, f);
        }
        KP_SYNC();
    }
    // subtree sums of the body wrenches -> sa (cacc no longer needed), then project on the dofs
    for (int it = tid; it < D_NB * 6; it += NT) {
        int b = it / 6, c = it - 6 * b, n = T.body_subtree[b];
        float acc = 0.f;
        for (int k = b; k < b + n; k++) acc += s.sw[6 * k + c];
        s.sa[it] = acc;
    }
    KP_SYNC();
    for (int d = tid; d < D_NV; d += NT) s.bias[d] = dot6(lds6(s.cdof + 6 * d), lds6(s.sa + 6 * T.dof_body[d]));
    KP_SYNC();
}

// ---------------------------------------------------------------- composite inertia -> sparse M (mj_crb)
template <int NT>
__device__ void crb_mass_matrix(EnvLds& s, const DevTables& T, int tid) {
    for (int it = tid; it < D_NB * 10; it += NT) {
        int b = it / 10, c = it - 10 * b, n = T.body_subtree[b];
        float acc = 0.f;
        for (int k = b; k < b + n; k++) acc += s.cinert[10 * k + c];
        s.crb[it] = acc;
    }
    KP_SYNC();
    for (int d = tid; d < D_NV; d += NT) sts6(s.f6 + 6 * d, inert_mul(s.crb + 10 * T.dof_body[d], lds6(s.cdof + 6 * d)));
    KP_SYNC();
    for (int e = tid; e < D_NM; e += NT) {
        int i = T.m_row[e], j = T.m_col[e];
        float v = dot6(lds6(s.cdof + 6 * j), lds6(s.f6 + 6 * i));
        if (i == j) v += T.dof_armature[i];
        s.qM[e] = v;
    }
    KP_SYNC();
}

// ---------------------------------------------------------------- tree-sparse L^T D L of s.qLD (rows left unscaled, 1/D in diaginv)
template <int NT>
struct PairTable {
    static constexpr int NP = (D_MAXDEPTH * (D_MAXDEPTH - 1) / 2 + NT - 1) / NT;  // 435 ancestor pairs max
    int u[NP], a[NP];
    __device__ void init(int tid) {
#pragma unroll
        for (int n = 0; n < NP; n++) {
            int t = tid + n * NT;
            int uu = (int)((1.0f + sqrtf(8.0f * t + 1.0f)) * 0.5f);
            while (uu * (uu - 1) / 2 > t) uu--;
            while ((uu + 1) * uu / 2 <= t) uu++;
            u[n] = uu; a[n] = t - uu * (uu - 1) / 2 + 1;
        }
    }
};

template <int NT>
__device__ void factor_sparse(EnvLds& s, const DevTables& T, const PairTable<NT>& pt, int tid) {
    for (int k = D_NV - 1; k > 0; k--) {
        const int D = T.dof_depth[k], adr = T.dof_madr[k];
        const float dinv = 1.0f / s.qLD[adr];
#pragma unroll
        for (int n = 0; n < PairTable<NT>::NP; n++) {
            int u = pt.u[n], a = pt.a[n];
            if (u <= D) {
                int tgt = T.anc_madr[k * D_MAXDEPTH + a] + (u - a);
                s.qLD[tgt] -= s.qLD[adr + a] * dinv * s.qLD[adr + u];
            }
        }
        if (tid == 0) s.diaginv[k] = dinv;
        KP_SYNC();
    }
    if (tid == 0) s.diaginv[0] = 1.0f / s.qLD[0];
    KP_SYNC();
}

// x <- (L^T D L)^-1 x, x = s.x
template <int NT>
__device__ void solve_sparse(EnvLds& s, const DevTables& T, const uint8_t* __restrict__ anc_dof, int tid) {
    for (int i = D_NV - 1; i > 0; i--) {
        const int D = T.dof_depth[i], adr = T.dof_madr[i];
        const float xi = s.x[i] * s.diaginv[i];
        for (int c = tid + 1; c <= D; c += NT) {
            int j = anc_dof[i * D_MAXDEPTH + c];
            s.x[j] -= s.qLD[adr + c] * xi;
        }
        KP_SYNC();
    }
    for (int i = tid; i < D_NV; i += NT) s.x[i] *= s.diaginv[i];
    KP_SYNC();
    for (int j = 0; j < D_NV - 1; j++) {
        const int n = T.dof_nsub[j], dj = T.dof_depth[j];
        const float xj = s.x[j];
        for (int i = j + 1 + tid; i <= j + n; i += NT) {
            int c = T.dof_depth[i] - dj;
            s.x[i] -= s.qLD[T.dof_madr[i] + c] * s.diaginv[i] * xj;
        }
        KP_SYNC();
    }
}

// ---------------------------------------------------------------- stable-PD torque + residual force (reference controller)
template <int NT>
__device__ void spd_torque_rfc(EnvLds& s, const DevTables& T, const Params& P, const PairTable<NT>& pt,
                               const uint8_t* __restrict__ anc_dof, int tid) {
    for (int e = tid; e < D_NM; e += NT) s.qLD[e] = s.qM[e];
    KP_SYNC();
    for (int i = tid; i < D_NV; i += NT) {
        float ep = 0.f, kp = 0.f, kd = 0.f;
        if (i >= 6) {
            int j = i - 6;
            float q = s.qpos[i + 1], base = s.tq[i + 1];
            while (base - q > 3.14159265358979f) base -= 6.28318530717959f;
            while (base - q < -3.14159265358979f) base += 6.28318530717959f;
            float target = base + s.act[j] * T.ascale[j];
            kp = T.kp[j]; kd = T.kd[j];
            ep = q + s.qvel[i] * P.h - target;
            s.qLD[T.dof_madr[i]] += kd * P.h;
        }
        s.search[i] = ep;
        s.x[i] = -s.bias[i] - kp * ep - kd * s.qvel[i];
    }
    KP_SYNC();
    factor_sparse<NT>(s, T, pt, tid);
    solve_sparse<NT>(s, T, anc_dof, tid);
    for (int j = tid; j < D_NU; j += NT) {
        int i = j + 6;
        float tq = -T.kp[j] * s.search[i] - T.kd[j] * (s.qvel[i] + s.x[i] * P.h);
        float lim = T.tlim[j];
        s.ctrl[j] = fminf(fmaxf(tq, -lim), lim);
    }
    if (tid == 0) {  // rfc_implicit
        Q4 cq = qmul(Q4{s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]}, Q4{P.br_inv[0], P.br_inv[1], P.br_inv[2], P.br_inv[3]});
        float hn = sqrtf(cq.w * cq.w + cq.z * cq.z);
        Q4 hq = Q4{cq.w / hn, 0.f, 0.f, cq.z / hn};
        V3 f = qrot(hq, v3(s.act[69] * P.rfc_scale, s.act[70] * P.rfc_scale, s.act[71] * P.rfc_scale));
        float vf[6] = {f.x, f.y, f.z, s.act[72] * P.rfc_scale, s.act[73] * P.rfc_scale, s.act[74] * P.rfc_scale};
#pragma unroll
        for (int k = 0; k < 6; k++) s.applied[k] = fminf(fmaxf(vf[k], -P.rfc_lim), P.rfc_lim);
    }
    KP_SYNC();
}

// ---------------------------------------------------------------- hull-vs-plane collision (wave 0; lane = hull vertex)
template <int NT>
__device__ void collide_plane(EnvLds& s, const DevTables& T, const Params& P, int tid) {
    if (tid < 64) {
        bool near = (tid < D_NB) && P.contact && !(s.xpos[3 * (tid < D_NB ? tid : 0) + 2] - T.body_rbound[tid < D_NB ? tid : 0] > P.margin);
        unsigned long long mask = __ballot(near);
        int ncon = 0;
        for (int b = 0; b < D_NB; b++) {
            if (tid == 0) s.con_start[b] = ncon;
            if (!((mask >> b) & 1ull)) continue;
            const int vadr = T.vert_adr[b], nvb = T.vert_adr[b + 1] - vadr;
            const float* R = s.xmat + 9 * b;
            V3 v = v3(0.f, 0.f, 0.f);
            float dist = 3.0e38f;
            if (tid < nvb) { v = ld3(T.verts + 3 * (vadr + tid)); dist = s.xpos[3 * b + 2] + R[6] * v.x + R[7] * v.y + R[8] * v.z; }
            bool cand = dist < P.margin;
            for (int r = 0; r < D_CON_PER_GEOM; r++) {
                float dmin = cand ? dist : 3.0e38f;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, o, 64));
                if (!(dmin < P.margin)) break;
                int idx = (cand && dist == dmin) ? tid : 64;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) idx = min(idx, __shfl_xor(idx, o, 64));
                if (ncon < D_MAXCON) {
                    if (tid == idx) {
                        V3 w = mulmat(R, v);
                        s.con_pos[3 * ncon] = s.xpos[3 * b] + w.x; s.con_pos[3 * ncon + 1] = s.xpos[3 * b + 1] + w.y;
                        s.con_pos[3 * ncon + 2] = s.xpos[3 * b + 2] + w.z - 0.5f * dist;
                        s.con_dist[ncon] = dist; s.con_body[ncon] = b;
                    }
                    ncon++;
                }
                if (tid == idx) cand = false;
            }
        }
        if (tid == 0) { s.con_start[D_NB] = ncon; s.ncon = ncon; s.nlim = 0; }
    }
    KP_SYNC();
}

// contact row e (0..3) direction for the plane frame n=(0,0,1), t1=(0,1,0), t2=(-1,0,0): dir = n +- mu t
__device__ __forceinline__ V3 row_dir(int e, float mu) {
    float sg = (e & 1) ? -mu : mu;
    return (e < 2) ? v3(0.f, sg, 1.f) : v3(-sg, 0.f, 1.f);
}

template <int NT>
__device__ void make_constraint(EnvLds& s, const DevTables& T, const Params& P, int tid) {
    const V3 o = ld3(s.xpos);
    for (int c = tid; c < s.ncon; c += NT) {
        int b = s.con_body[c];
        float r = s.con_dist[c] - P.margin;
        float imp = impedance(P, r);
        float dA = T.body_invw[b] * (1.0f + P.mu * P.mu);
        float Rn = fmaxf(1e-15f, (1.0f - imp) * dA / imp);
        s.con_D[c] = 1.0f / (2.0f * P.mu * P.mu * Rn);
        S6 cv = lds6(s.sv + 6 * b);
        V3 vpt = cv.l + cross(cv.a, ld3(s.con_pos + 3 * c) - o);
#pragma unroll
        for (int e = 0; e < 4; e++) s.aref[4 * c + e] = -P.B * dot(row_dir(e, P.mu), vpt) - P.K * imp * r;
    }
    for (int j = tid; j < D_NU; j += NT) {
        float sgn = 0.f, aref = 0.f, Dl = 0.f;
        if (P.limits && T.jnt_limited[j]) {
            float q = s.qpos[7 + j], dlo = q - T.jnt_lo[j], dhi = T.jnt_hi[j] - q;
            float dist = 0.f;
            if (dlo < 0.f) { sgn = 1.f; dist = dlo; } else if (dhi < 0.f) { sgn = -1.f; dist = dhi; }
            if (sgn != 0.f) {
                float imp = impedance(P, dist);
                Dl = 1.0f / fmaxf(1e-15f, (1.0f - imp) * T.lim_invw[j] / imp);
                aref = -P.B * (sgn * s.qvel[6 + j]) - P.K * imp * dist;
                atomicAdd(&s.nlim, 1);
            }
        }
        s.lim_sgn[j] = sgn; s.lim_aref[j] = aref; s.lim_D[j] = Dl;
    }
    KP_SYNC();
}

// spatial "velocity" of every body induced by a generalized vector:  sv[b] = sum over ancestor dofs cdof_d * vec[d]
template <int NT>
__device__ void spatial_accumulate(EnvLds& s, const DevTables& T, const float* vec, int tid) {
    for (int lev = 0; lev < D_NLEV; lev++) {
        int l0 = T.lev_start[lev], nl = T.lev_start[lev + 1] - l0;
        for (int it = tid; it < nl * 6; it += NT) {
            int b = T.lev_body[l0 + it / 6], c = it % 6;
            float acc;
            if (b == 0) {
                acc = 0.f;
#pragma unroll
                for (int d = 0; d < 6; d++) acc += s.cdof[6 * d + c] * vec[d];
            } else {
                acc = s.sv[6 * T.body_parent[b] + c];
                int d0 = 6 + 3 * (b - 1);
#pragma unroll
                for (int d = d0; d < d0 + 3; d++) acc += s.cdof[6 * d + c] * vec[d];
            }
            s.sv[6 * b + c] = acc;
        }
        KP_SYNC();
    }
}

// rows: out[r] = w_r . sv[body_r] (- aref if sub_aref); limits likewise
template <int NT>
__device__ void eval_rows(EnvLds& s, const float* vec, float* rows, float* lim_rows, bool sub_aref, const Params& P, int tid) {
    const V3 o = ld3(s.xpos);
    for (int r = tid; r < 4 * s.ncon; r += NT) {
        int c = r >> 2;
        S6 S = lds6(s.sv + 6 * s.con_body[c]);
        V3 vpt = S.l + cross(S.a, ld3(s.con_pos + 3 * c) - o);
        rows[r] = dot(row_dir(r & 3, P.mu), vpt) - (sub_aref ? s.aref[r] : 0.f);
    }
    for (int j = tid; j < D_NU; j += NT) lim_rows[j] = s.lim_sgn[j] * vec[6 + j] - (sub_aref ? s.lim_aref[j] : 0.f);
    KP_SYNC();
}

// out = M vec (with_inertia) - J^T f(jar) (with_forces);  sv must hold spatial_accumulate(vec) when with_inertia
template <int NT>
__device__ void wrench_project(EnvLds& s, const DevTables& T, const Params& P, const float* vec, float* out,
                               bool with_inertia, bool with_forces, int tid) {
    if (tid < D_NB) {
        const int b = tid;
        S6 W = with_inertia ? inert_mul(s.cinert + 10 * b, lds6(s.sv + 6 * b)) : S6{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};
        if (with_forces) {
            const V3 o = ld3(s.xpos);
            for (int c = s.con_start[b]; c < s.con_start[b + 1]; c++) {
                V3 F = v3(0.f, 0.f, 0.f);
                float Dc = s.con_D[c];
#pragma unroll
                for (int e = 0; e < 4; e++) { float jr = s.jar[4 * c + e]; if (jr < 0.f) F = F + (-Dc * jr) * row_dir(e, P.mu); }
                V3 p = ld3(s.con_pos + 3 * c) - o;
                W.a = W.a - cross(p, F); W.l = W.l - F;
            }
        }
        sts6(s.sw + 6 * b, W);
    }
    KP_SYNC();
    for (int it = tid; it < D_NB * 6; it += NT) {
        int b = it / 6, c = it - 6 * b, n = T.body_subtree[b];
        float acc = 0.f;
        for (int k = b; k < b + n; k++) acc += s.sw[6 * k + c];
        s.sa[it] = acc;
    }
    KP_SYNC();
    for (int d = tid; d < D_NV; d += NT) {
        float v = dot6(lds6(s.cdof + 6 * d), lds6(s.sa + 6 * T.dof_body[d]));
        if (with_inertia) v += T.dof_armature[d] * vec[d];
        if (with_forces && d >= 6) { float jr = s.lim_jar[d - 6]; if (jr < 0.f) v -= s.lim_sgn[d - 6] * (-s.lim_D[d - 6] * jr); }
        out[d] = v;
    }
    KP_SYNC();
}

// Newton Hessian H = M + J^T D_active J  -> s.qLD   (contact "inertia" composite pass)
template <int NT>
__device__ void assemble_hessian(EnvLds& s, const DevTables& T, const Params& P, int tid) {
    if (tid < D_NB) {
        const int b = tid;
        float Kb[21];
#pragma unroll
        for (int k = 0; k < 21; k++) Kb[k] = 0.f;
        const V3 o = ld3(s.xpos);
        for (int c = s.con_start[b]; c < s.con_start[b + 1]; c++) {
            V3 p = ld3(s.con_pos + 3 * c) - o;
            float Dc = s.con_D[c];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (s.jar[4 * c + e] < 0.f) {
                    V3 dir = row_dir(e, P.mu);
                    V3 m = cross(p, dir);
                    float w[6] = {m.x, m.y, m.z, dir.x, dir.y, dir.z};
                    int k = 0;
#pragma unroll
                    for (int r = 0; r < 6; r++)
#pragma unroll
                        for (int cc = r; cc < 6; cc++) Kb[k++] += Dc * w[r] * w[cc];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 21; k++) s.K[21 * b + k] = Kb[k];
    }
    KP_SYNC();
    float* Ksub = s.qLD;  // scratch: qLD is rewritten below
    for (int it = tid; it < D_NB * 21; it += NT) {
        int b = it / 21, c = it - 21 * b, n = T.body_subtree[b];
        float acc = 0.f;
        for (int k = b; k < b + n; k++) acc += s.K[21 * k + c];
        Ksub[it] = acc;
    }
    KP_SYNC();
    for (int d = tid; d < D_NV; d += NT) {
        const float* Kd = Ksub + 21 * T.dof_body[d];
        float cd[6], f[6];
#pragma unroll
        for (int k = 0; k < 6; k++) cd[k] = s.cdof[6 * d + k];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 6; c++) acc += Kd[sym6_idx(r, c)] * cd[c];
            f[r] = acc;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) s.f6[6 * d + k] = f[k];
    }
    KP_SYNC();
    for (int e = tid; e < D_NM; e += NT) {
        int i = T.m_row[e], j = T.m_col[e];
        float v = s.qM[e] + dot6(lds6(s.cdof + 6 * j), lds6(s.f6 + 6 * i));
        if (i == j && i >= 6 && s.lim_jar[i - 6] < 0.f) v += s.lim_D[i - 6];
        s.qLD[e] = v;
    }
    KP_SYNC();
}

// primal cost at the current (qacc, mres, jar):  0.5 mres.(qacc - qacc_s) + sum 0.5 D jar_-^2
template <int NT>
__device__ float primal_cost(EnvLds& s, int tid) {
    float c = 0.f;
    for (int i = tid; i < D_NV; i += NT) c += 0.5f * s.mres[i] * (s.qacc[i] - s.qacc_s[i]);
    for (int r = tid; r < 4 * s.ncon; r += NT) { float x = s.jar[r]; if (x < 0.f) c += 0.5f * s.con_D[r >> 2] * x * x; }
    for (int j = tid; j < D_NU; j += NT) { float x = s.lim_jar[j]; if (x < 0.f) c += 0.5f * s.lim_D[j] * x * x; }
    return block_sum<NT>(s, c, tid);
}

// constraint solve: Newton on the primal problem, exact line search.  Returns iterations used.
template <int NT>
__device__ int solve_constraints(EnvLds& s, const DevTables& T, const Params& P, const PairTable<NT>& pt,
                                 const uint8_t* __restrict__ anc_dof, int tid) {
    // start from qacc_smooth: M qacc_s = qfrc_smooth  =>  mres = 0
    for (int i = tid; i < D_NV; i += NT) { s.qacc[i] = s.qacc_s[i]; s.mres[i] = 0.f; }
    for (int j = tid; j < D_NU; j += NT) s.lim_jar[j] = 0.f;
    KP_SYNC();
    if (s.ncon == 0 && s.nlim == 0) return 0;
    spatial_accumulate<NT>(s, T, s.qacc, tid);
    eval_rows<NT>(s, s.qacc, s.jar, s.lim_jar, true, P, tid);
    // inactive limit rows must never look active
    for (int j = tid; j < D_NU; j += NT) if (s.lim_sgn[j] == 0.f) s.lim_jar[j] = 0.f;
    KP_SYNC();
    float cost = primal_cost<NT>(s, tid);
    int it = 0;
    for (; it < P.max_iter; it++) {
        // gradient = mres - J^T f
        wrench_project<NT>(s, T, P, nullptr, s.grad, false, true, tid);
        float g2 = 0.f;
        for (int i = tid; i < D_NV; i += NT) { float g = s.mres[i] + s.grad[i]; s.grad[i] = g; g2 += g * g; }
        g2 = block_sum<NT>(s, g2, tid);
        KP_SYNC();
        if (P.scale * sqrtf(g2) < P.tol) break;
        assemble_hessian<NT>(s, T, P, tid);
        factor_sparse<NT>(s, T, pt, tid);
        for (int i = tid; i < D_NV; i += NT) s.x[i] = -s.grad[i];
        KP_SYNC();
        solve_sparse<NT>(s, T, anc_dof, tid);
        for (int i = tid; i < D_NV; i += NT) s.search[i] = s.x[i];
        KP_SYNC();
        spatial_accumulate<NT>(s, T, s.search, tid);
        eval_rows<NT>(s, s.search, s.jv, s.lim_jv, false, P, tid);
        wrench_project<NT>(s, T, P, s.search, s.Mv, true, false, tid);
        // exact line search on phi(alpha)
        float g0 = 0.f, h0 = 0.f;
        for (int i = tid; i < D_NV; i += NT) { g0 += s.search[i] * s.mres[i]; h0 += s.search[i] * s.Mv[i]; }
        g0 = block_sum<NT>(s, g0, tid); h0 = block_sum<NT>(s, h0, tid);
        float alpha = 0.f, lo = 0.f, hi = 3.0e38f;
        for (int ls = 0; ls < 20; ls++) {
            float d1 = 0.f, d2 = 0.f;
            for (int r = tid; r < 4 * s.ncon; r += NT) {
                float jv = s.jv[r], x = s.jar[r] + alpha * jv;
                if (x < 0.f) { float Dc = s.con_D[r >> 2]; d1 += Dc * x * jv; d2 += Dc * jv * jv; }
            }
            for (int j = tid; j < D_NU; j += NT) {
                if (s.lim_sgn[j] != 0.f) { float jv = s.lim_jv[j], x = s.lim_jar[j] + alpha * jv; if (x < 0.f) { d1 += s.lim_D[j] * x * jv; d2 += s.lim_D[j] * jv * jv; } }
            }
            d1 = block_sum<NT>(s, d1, tid); d2 = block_sum<NT>(s, d2, tid);
            float dphi = g0 + alpha * h0 + d1, ddphi = h0 + d2;
            if (!(ddphi > 0.f)) break;
            if (dphi < 0.f) lo = alpha; else hi = alpha;
            float an = alpha - dphi / ddphi;
            if (!(an > lo && an < hi)) an = hi < 1.0e38f ? 0.5f * (lo + hi) : 2.0f * alpha + 1.0f;
            float step = an - alpha;
            alpha = an;
            if (fabsf(step) <= 1e-6f * fabsf(alpha)) break;
        }
        if (!(alpha > 0.f)) break;
        for (int i = tid; i < D_NV; i += NT) { s.qacc[i] += alpha * s.search[i]; s.mres[i] += alpha * s.Mv[i]; }
        for (int r = tid; r < 4 * s.ncon; r += NT) s.jar[r] += alpha * s.jv[r];
        for (int j = tid; j < D_NU; j += NT) if (s.lim_sgn[j] != 0.f) s.lim_jar[j] += alpha * s.lim_jv[j];
        KP_SYNC();
        float newcost = primal_cost<NT>(s, tid);
        float improvement = P.scale * (cost - newcost);
        cost = newcost;
        if (improvement < P.tol) { it++; break; }
    }
    return it;
}

// ---------------------------------------------------------------- the kernel
template <int NT>
__global__ __launch_bounds__(NT) void kp_step_kernel(StepArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    EnvLds& s = *reinterpret_cast<EnvLds*>(smem_raw);
    const int env = blockIdx.x, tid = threadIdx.x;
    if (env >= A.n_envs) return;
    if (A.env_mask && !A.env_mask[env]) return;
    const DevTables& T = A.T;
    const Params& P = A.P;
    const uint8_t* anc_dof = T.anc_dof;
    PairTable<NT> pt; pt.init(tid);

    // ---- load: derived state first (the state the last forward pass ran on), then the real state
    for (int i = tid; i < D_NQ; i += NT) { s.qpos[i] = A.qpos_d[(size_t)env * D_NQ + i]; s.tq[i] = A.target_qpos ? A.target_qpos[(size_t)env * D_NQ + i] : 0.f; }
    for (int i = tid; i < D_NV; i += NT) { s.qvel[i] = A.qvel_d[(size_t)env * D_NV + i]; s.act[i] = A.action ? A.action[(size_t)env * D_NV + i] : 0.f; s.warm[i] = A.warm[(size_t)env * D_NV + i]; }
    if (tid < 8) s.applied[tid] = 0.f;
    if (tid == 0) { s.ncon = 0; s.nlim = 0; s.flag = 0; }
    KP_SYNC();
    forward_kin_bias<NT>(s, T, P, tid);
    if (A.n_substeps > 0 && P.stale) crb_mass_matrix<NT>(s, T, tid);
    float qd_save_q[(D_NQ + NT - 1) / NT], qd_save_v[(D_NV + NT - 1) / NT];
#pragma unroll
    for (int n = 0; n < (D_NQ + NT - 1) / NT; n++) { int i = tid + n * NT; qd_save_q[n] = i < D_NQ ? s.qpos[i] : 0.f; }
#pragma unroll
    for (int n = 0; n < (D_NV + NT - 1) / NT; n++) { int i = tid + n * NT; qd_save_v[n] = i < D_NV ? s.qvel[i] : 0.f; }
    KP_SYNC();
    if (A.n_substeps > 0) {
        for (int i = tid; i < D_NQ; i += NT) s.qpos[i] = A.qpos[(size_t)env * D_NQ + i];
        for (int i = tid; i < D_NV; i += NT) s.qvel[i] = A.qvel[(size_t)env * D_NV + i];
        KP_SYNC();
    }
    int niter_total = 0, maxcon = 0;
    for (int sub = 0; sub < A.n_substeps; sub++) {
        if (P.stale) spd_torque_rfc<NT>(s, T, P, pt, anc_dof, tid);
        // ---- mj_forward at the current state
#pragma unroll
        for (int n = 0; n < (D_NQ + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NQ) qd_save_q[n] = s.qpos[i]; }
#pragma unroll
        for (int n = 0; n < (D_NV + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NV) qd_save_v[n] = s.qvel[i]; }
        forward_kin_bias<NT>(s, T, P, tid);
        crb_mass_matrix<NT>(s, T, tid);
        if (!P.stale) spd_torque_rfc<NT>(s, T, P, pt, anc_dof, tid);
        collide_plane<NT>(s, T, P, tid);
        make_constraint<NT>(s, T, P, tid);
        for (int i = tid; i < D_NV; i += NT) {
            float f = -s.bias[i] + (i < 6 ? s.applied[i] : s.ctrl[i - 6]);
            s.smooth[i] = f; s.x[i] = f;
        }
        for (int e = tid; e < D_NM; e += NT) s.qLD[e] = s.qM[e];
        KP_SYNC();
        factor_sparse<NT>(s, T, pt, tid);
        solve_sparse<NT>(s, T, anc_dof, tid);
        for (int i = tid; i < D_NV; i += NT) s.qacc_s[i] = s.x[i];
        KP_SYNC();
        niter_total += solve_constraints<NT>(s, T, P, pt, anc_dof, tid);
        maxcon = max(maxcon, s.ncon);
        // ---- semi-implicit Euler (mj_Euler, no damping)
        for (int i = tid; i < D_NV; i += NT) { float v = s.qvel[i] + P.h * s.qacc[i]; s.qvel[i] = v; s.warm[i] = s.qacc[i]; }
        KP_SYNC();
        for (int j = tid; j < D_NU; j += NT) s.qpos[7 + j] += P.h * s.qvel[6 + j];
        if (tid == 0) {
            s.qpos[0] += P.h * s.qvel[0]; s.qpos[1] += P.h * s.qvel[1]; s.qpos[2] += P.h * s.qvel[2];
            V3 w = ld3(s.qvel + 3);
            float n = sqrtf(dot(w, w));
            Q4 qr = Q4{1.f, 0.f, 0.f, 0.f};
            if (n >= 1e-15f) { float sn, cs; sincosf(0.5f * P.h * n, &sn, &cs); float k = sn / n; qr = Q4{cs, w.x * k, w.y * k, w.z * k}; }
            Q4 q = qmul(qnormalize(Q4{s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]}), qr);
            s.qpos[3] = q.w; s.qpos[4] = q.x; s.qpos[5] = q.y; s.qpos[6] = q.z;
        }
        KP_SYNC();
    }
    if (A.n_substeps > 0 && !P.stale) {  // fresh mode: outputs are the kinematics of the final state
#pragma unroll
        for (int n = 0; n < (D_NQ + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NQ) qd_save_q[n] = s.qpos[i]; }
#pragma unroll
        for (int n = 0; n < (D_NV + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NV) qd_save_v[n] = s.qvel[i]; }
        forward_kin_bias<NT>(s, T, P, tid);
    }
    // ---- store
    bool bad = false;
    for (int i = tid; i < D_NQ; i += NT) { float v = s.qpos[i]; bad |= !(fabsf(v) < 1e10f); if (A.n_substeps > 0) A.qpos[(size_t)env * D_NQ + i] = v; }
    for (int i = tid; i < D_NV; i += NT) { float v = s.qvel[i]; bad |= !(fabsf(v) < 1e10f); if (A.n_substeps > 0) { A.qvel[(size_t)env * D_NV + i] = v; A.warm[(size_t)env * D_NV + i] = s.warm[i]; } }
#pragma unroll
    for (int n = 0; n < (D_NQ + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NQ) A.qpos_d[(size_t)env * D_NQ + i] = qd_save_q[n]; }
#pragma unroll
    for (int n = 0; n < (D_NV + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NV) A.qvel_d[(size_t)env * D_NV + i] = qd_save_v[n]; }
    for (int i = tid; i < 72; i += NT) { A.xpos[(size_t)env * 72 + i] = s.xpos[i]; A.xipos[(size_t)env * 72 + i] = s.xipos[i]; }
    for (int i = tid; i < 96; i += NT) A.xquat[(size_t)env * 96 + i] = s.xquat[i];
    if (bad) atomicOr(&s.flag, 1);
    KP_SYNC();
    if (tid == 0 && A.diag && A.n_substeps > 0) { int* dg = A.diag + 4 * (size_t)env; dg[0] = s.ncon; dg[1] = niter_total; dg[2] = s.flag; dg[3] = maxcon; }
}

}  // namespace kp
