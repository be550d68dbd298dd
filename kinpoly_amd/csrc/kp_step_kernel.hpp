// kp_step_kernel.hpp -- the fused control-step kernel: n_substeps x { stable-PD torque, residual
// force, mj_step-equivalent forward dynamics with hull-plane soft contact, semi-implicit Euler }.
//
// Replaces HOT LOOP C of the reference: HumanoidEnv.do_simulation (uhc/envs/humanoid_im.py:506-533)
//   compute_torque / compute_desired_accel   uhc/envs/humanoid_im.py:418-480
//   rfc_implicit                              uhc/envs/humanoid_im.py:497-504
//   sim.step()  (MuJoCo mj_step)              uhc/envs/humanoid_im.py:527         [MJ-ext]
// The arithmetic follows oracle/kp_oracle.c (fp64) in fp32, organised MATRIX-FREE for a wavefront:
//   * every spatial quantity is expressed in ONE world-aligned frame at the root body origin, so tree
//     recursions need no coordinate transforms: parents and children simply add;
//   * kinematics, velocities and bias forces (RNE) come out of one level-synchronous pass (9 levels);
//   * every linear solve -- (M + K_d h) for the stable-PD controller, M for the smooth acceleration,
//     M + J^T D J for the Newton step of the contact solver -- is an articulated-body (ABA) pass:
//     leaves->root articulated inertia + bias, root->leaves accelerations.  K_d h and joint-limit terms
//     enter as extra joint armature, active contact rows as a per-body 6x6 "contact inertia" D w w^T.
//     The joint-space mass matrix is never formed or factorised (the reference materialises a dense
//     105x105 M and a Cholesky factor per substep);
//   * J v is read off the spatial accelerations the ABA forward pass leaves behind, J^T f and M v are a
//     body-wrench subtree sum projected on the dofs.
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

__device__ __forceinline__ constexpr int s6i(int r, int c) { return r <= c ? (r * (13 - r)) / 2 + (c - r) : (c * (13 - c)) / 2 + (r - c); }

// contact row e (0..3) of the pyramid in the plane frame n=(0,0,1), t1=(0,1,0), t2=(-1,0,0): dir = n +- mu t
__device__ __forceinline__ V3 row_dir(int e, float mu) {
    float sg = (e & 1) ? -mu : mu;
    return (e < 2) ? v3(0.f, sg, 1.f) : v3(-sg, 0.f, 1.f);
}
// row value from contact-frame components (n, t1, t2)
__device__ __forceinline__ float row_val(int e, float mu, float jn, float jt1, float jt2) {
    float sg = (e & 1) ? -mu : mu;
    return jn + sg * (e < 2 ? jt1 : jt2);
}
__device__ __forceinline__ V3 to_frame(V3 v) { return v3(v.z, v.y, -v.x); }  // (n, t1, t2) components of a world vector

// ---------------------------------------------------------------- kinematics + velocities + bias (one tree pass)
template <int NT>
__device__ void forward_kin_bias(EnvLds& s, const DevTables& T, const Params& P, int depth, int tid) {
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
                const int p = s.bpar[b];
                o = ld3(s.xpos);
                q = Q4{s.xquat[4 * p], s.xquat[4 * p + 1], s.xquat[4 * p + 2], s.xquat[4 * p + 3]};
                pos = ld3(s.xpos + 3 * p) + qrot(q, ld3(T.body_pos + 3 * b));
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
        int b = it / 6, c = it - 6 * b, n = s.bsub[b];
        float acc = 0.f;
        for (int k = b; k < b + n; k++) acc += s.sw[6 * k + c];
        s.sa[it] = acc;
    }
    KP_SYNC();
    for (int d = tid; d < D_NV; d += NT) s.bias[d] = dot6(lds6(s.cdof + 6 * d), lds6(s.sa + 6 * s.dbody[d]));
    KP_SYNC();
}

// ---------------------------------------------------------------- articulated-body solve
// out = (M + diag(s.extra) [+ J^T D_active J])^-1 rhs.  Leaves s.sv[b] = sum over ancestor dofs cdof_d out_d
// (the spatial "acceleration" of every body induced by out).  rhs/out are LDS vectors (may alias).
template <int NT>
__device__ void aba_solve(EnvLds& s, const Params& P, const float* rhs, float* out, bool contact_inertia, int depth, int tid) {
    for (int lev = D_NLEV - 1; lev >= 0; lev--) {
        if (depth == lev) {
            const int b = tid;
            float IA[21], pA[6];
            {
                const float* ci = s.cinert + 10 * b;
                const float hx = ci[6], hy = ci[7], hz = ci[8], m = ci[9];
                IA[s6i(0, 0)] = ci[0]; IA[s6i(0, 1)] = ci[3]; IA[s6i(0, 2)] = ci[4]; IA[s6i(0, 3)] = 0.f; IA[s6i(0, 4)] = -hz; IA[s6i(0, 5)] = hy;
                IA[s6i(1, 1)] = ci[1]; IA[s6i(1, 2)] = ci[5]; IA[s6i(1, 3)] = hz; IA[s6i(1, 4)] = 0.f; IA[s6i(1, 5)] = -hx;
                IA[s6i(2, 2)] = ci[2]; IA[s6i(2, 3)] = -hy; IA[s6i(2, 4)] = hx; IA[s6i(2, 5)] = 0.f;
                IA[s6i(3, 3)] = m; IA[s6i(3, 4)] = 0.f; IA[s6i(3, 5)] = 0.f; IA[s6i(4, 4)] = m; IA[s6i(4, 5)] = 0.f; IA[s6i(5, 5)] = m;
            }
#pragma unroll
            for (int k = 0; k < 6; k++) pA[k] = 0.f;
            if (contact_inertia) {
                const V3 o = ld3(s.xpos);
                for (int c = s.con_start[b]; c < s.con_start[b + 1]; c++) {
                    V3 p = ld3(s.con_pos + 3 * c) - o;
                    const float Dc = s.con_D[c], jn = s.jar3[3 * c], jt1 = s.jar3[3 * c + 1], jt2 = s.jar3[3 * c + 2];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (row_val(e, P.mu, jn, jt1, jt2) < 0.f) {
                            V3 dir = row_dir(e, P.mu);
                            V3 mm = cross(p, dir);
                            float w[6] = {mm.x, mm.y, mm.z, dir.x, dir.y, dir.z};
#pragma unroll
                            for (int r = 0; r < 6; r++)
#pragma unroll
                                for (int cc = r; cc < 6; cc++) IA[s6i(r, cc)] += Dc * w[r] * w[cc];
                        }
                    }
                }
            }
            const int nsub = s.bsub[b];
            for (int k = b + 1; k < b + nsub; k += s.bsub[k]) {
#pragma unroll
                for (int i = 0; i < 21; i++) IA[i] += s.IAa[21 * k + i];
#pragma unroll
                for (int i = 0; i < 6; i++) pA[i] += s.pAa[6 * k + i];
            }
            const int nd = b == 0 ? 6 : 3, d0 = b == 0 ? 0 : 6 + 3 * (b - 1);
            for (int j = nd - 1; j >= 0; j--) {
                const int d = d0 + j;
                float sj[6], Uv[6];
#pragma unroll
                for (int k = 0; k < 6; k++) sj[k] = s.cdof[6 * d + k];
                float D = s.arm[d] + s.extra[d], u = rhs[d];
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    float acc = 0.f;
#pragma unroll
                    for (int c = 0; c < 6; c++) acc += IA[s6i(r, c)] * sj[c];
                    Uv[r] = acc;
                    D += sj[r] * acc;
                    u -= sj[r] * pA[r];
                }
                const float Dinv = 1.0f / D, ud = u * Dinv;
#pragma unroll
                for (int k = 0; k < 6; k++) s.U[6 * d + k] = Uv[k];
                s.Dinv[d] = Dinv; s.uj[d] = u;
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const float ur = Uv[r] * Dinv;
#pragma unroll
                    for (int c = r; c < 6; c++) IA[s6i(r, c)] -= ur * Uv[c];
                    pA[r] += Uv[r] * ud;
                }
            }
#pragma unroll
            for (int i = 0; i < 21; i++) s.IAa[21 * b + i] = IA[i];
#pragma unroll
            for (int i = 0; i < 6; i++) s.pAa[6 * b + i] = pA[i];
        }
        KP_SYNC();
    }
    for (int lev = 0; lev < D_NLEV; lev++) {
        if (depth == lev) {
            const int b = tid;
            S6 a = b == 0 ? S6{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)} : lds6(s.sv + 6 * s.bpar[b]);
            const int nd = b == 0 ? 6 : 3, d0 = b == 0 ? 0 : 6 + 3 * (b - 1);
            for (int j = 0; j < nd; j++) {
                const int d = d0 + j;
                float qdd = (s.uj[d] - dot6(lds6(s.U + 6 * d), a)) * s.Dinv[d];
                out[d] = qdd;
                a = a + qdd * lds6(s.cdof + 6 * d);
            }
            sts6(s.sv + 6 * b, a);
        }
        KP_SYNC();
    }
}

// ---------------------------------------------------------------- stable-PD torque + residual force (reference controller)
template <int NT>
__device__ void spd_torque_rfc(EnvLds& s, const DevTables& T, const Params& P, int depth, int tid) {
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
        }
        s.extra[i] = kd * P.h;                      // (M + K_d dt): K_d dt is extra joint armature
        s.search[i] = ep;
        s.x[i] = -s.bias[i] - kp * ep - kd * s.qvel[i];
    }
    KP_SYNC();
    aba_solve<NT>(s, P, s.x, s.x, false, depth, tid);
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
        const int bb = tid < D_NB ? tid : 0;
        bool near = (tid < D_NB) && P.contact && !(s.xpos[3 * bb + 2] - T.body_rbound[bb] > P.margin);
        unsigned long long mask = __ballot(near);
        int ncon = 0;
        for (int b = 0; b < D_NB; b++) {
            if (tid == 0) s.con_start[b] = ncon;
            if (!((mask >> b) & 1ull)) continue;
            const int vadr = T.vert_adr[b], nvb = T.vert_adr[b + 1] - vadr;
            float R[9];
            q2mat(Q4{s.xquat[4 * b], s.xquat[4 * b + 1], s.xquat[4 * b + 2], s.xquat[4 * b + 3]}, R);
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

// efc_D and the reference acceleration of every constraint row.  aref (contact-frame 3-vector) goes to jv3.
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
        S6 cv = lds6(s.sv + 6 * b);                         // sv still holds cvel from forward_kin_bias
        V3 vf = to_frame(cv.l + cross(cv.a, ld3(s.con_pos + 3 * c) - o));
        s.jv3[3 * c] = -P.B * vf.x - P.K * imp * r; s.jv3[3 * c + 1] = -P.B * vf.y; s.jv3[3 * c + 2] = -P.B * vf.z;
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

// contact-frame residuals of all rows for the spatial accelerations in sv:  out3 = frame^T (point accel) [- aref]
template <int NT>
__device__ void eval_rows(EnvLds& s, const float* vec, float* out3, float* lim_rows, bool sub_aref, int tid) {
    const V3 o = ld3(s.xpos);
    for (int c = tid; c < s.ncon; c += NT) {
        S6 S = lds6(s.sv + 6 * s.con_body[c]);
        V3 a = to_frame(S.l + cross(S.a, ld3(s.con_pos + 3 * c) - o));
        if (sub_aref) { a.x -= s.jv3[3 * c]; a.y -= s.jv3[3 * c + 1]; a.z -= s.jv3[3 * c + 2]; }
        out3[3 * c] = a.x; out3[3 * c + 1] = a.y; out3[3 * c + 2] = a.z;
    }
    for (int j = tid; j < D_NU; j += NT) {
        float sg = s.lim_sgn[j];
        lim_rows[j] = sg != 0.f ? sg * vec[6 + j] - (sub_aref ? s.lim_aref[j] : 0.f) : 0.f;
    }
    KP_SYNC();
}

// out = M vec (with_inertia; sv must hold the spatial accelerations of vec) - J^T f(jar) (with_forces)
template <int NT>
__device__ void wrench_project(EnvLds& s, const Params& P, const float* vec, float* out, bool with_inertia, bool with_forces, int tid) {
    if (tid < D_NB) {
        const int b = tid;
        S6 W = with_inertia ? inert_mul(s.cinert + 10 * b, lds6(s.sv + 6 * b)) : S6{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};
        if (with_forces) {
            const V3 o = ld3(s.xpos);
            for (int c = s.con_start[b]; c < s.con_start[b + 1]; c++) {
                V3 F = v3(0.f, 0.f, 0.f);
                const float Dc = s.con_D[c], jn = s.jar3[3 * c], jt1 = s.jar3[3 * c + 1], jt2 = s.jar3[3 * c + 2];
#pragma unroll
                for (int e = 0; e < 4; e++) { float x = row_val(e, P.mu, jn, jt1, jt2); if (x < 0.f) F = F + (-Dc * x) * row_dir(e, P.mu); }
                V3 p = ld3(s.con_pos + 3 * c) - o;
                W.a = W.a - cross(p, F); W.l = W.l - F;
            }
        }
        sts6(s.sw + 6 * b, W);
    }
    KP_SYNC();
    for (int it = tid; it < D_NB * 6; it += NT) {
        int b = it / 6, c = it - 6 * b, n = s.bsub[b];
        float acc = 0.f;
        for (int k = b; k < b + n; k++) acc += s.sw[6 * k + c];
        s.sa[it] = acc;
    }
    KP_SYNC();
    for (int d = tid; d < D_NV; d += NT) {
        float v = dot6(lds6(s.cdof + 6 * d), lds6(s.sa + 6 * s.dbody[d]));
        if (with_inertia) v += s.arm[d] * vec[d];
        if (with_forces && d >= 6) { float jr = s.lim_jar[d - 6]; if (jr < 0.f) v -= s.lim_sgn[d - 6] * (-s.lim_D[d - 6] * jr); }
        out[d] = v;
    }
    KP_SYNC();
}

// primal cost at the current (qacc, mres, jar):  0.5 mres.(qacc - qacc_s) + sum 0.5 D jar_-^2
template <int NT>
__device__ float primal_cost(EnvLds& s, const Params& P, const float* qacc, int tid) {
    float c = 0.f;
    for (int i = tid; i < D_NV; i += NT) c += 0.5f * s.mres[i] * (qacc[i] - s.qacc_s[i]);
    for (int k = tid; k < s.ncon; k += NT) {
        const float Dc = s.con_D[k], jn = s.jar3[3 * k], jt1 = s.jar3[3 * k + 1], jt2 = s.jar3[3 * k + 2];
#pragma unroll
        for (int e = 0; e < 4; e++) { float x = row_val(e, P.mu, jn, jt1, jt2); if (x < 0.f) c += 0.5f * Dc * x * x; }
    }
    for (int j = tid; j < D_NU; j += NT) { float x = s.lim_jar[j]; if (x < 0.f) c += 0.5f * s.lim_D[j] * x * x; }
    return block_sum<NT>(s, c, tid);
}

// constraint solve: Newton on the primal problem (mj_solNewton) with an exact line search.  Returns iterations.
// On entry sv holds the spatial accelerations of qacc_s (left by the smooth aba_solve), jv3 holds aref.
template <int NT>
__device__ int solve_constraints(EnvLds& s, const Params& P, int depth, int tid) {
    for (int i = tid; i < D_NV; i += NT) { s.qacc[i] = s.qacc_s[i]; s.mres[i] = 0.f; }
    KP_SYNC();
    if (s.ncon == 0 && s.nlim == 0) return 0;
    // start from qacc_smooth (M qacc_s = qfrc_smooth => mres = 0)
    eval_rows<NT>(s, s.qacc_s, s.jar3, s.lim_jar, true, tid);
    float cost = primal_cost<NT>(s, P, s.qacc, tid);
    int it = 0;
    for (; it < P.max_iter; it++) {
        // gradient = mres - J^T f
        wrench_project<NT>(s, P, nullptr, s.grad, false, true, tid);
        float g2 = 0.f;
        for (int i = tid; i < D_NV; i += NT) {
            float g = s.mres[i] + s.grad[i];
            s.grad[i] = g; g2 += g * g;
            s.x[i] = -g;
            s.extra[i] = (i >= 6 && s.lim_jar[i - 6] < 0.f) ? s.lim_D[i - 6] : 0.f;
        }
        g2 = block_sum<NT>(s, g2, tid);
        KP_SYNC();
        if (P.scale * sqrtf(g2) < P.tol) break;
        // search = -H^-1 grad,  H = M + J^T D_active J : articulated-body pass with contact inertia
        aba_solve<NT>(s, P, s.x, s.search, true, depth, tid);
        eval_rows<NT>(s, s.search, s.jv3, s.lim_jv, false, tid);           // aref (in jv3) is folded into jar3 by now
        wrench_project<NT>(s, P, s.search, s.Mv, true, false, tid);
        // exact line search on phi(alpha)
        float g0 = 0.f, h0 = 0.f;
        for (int i = tid; i < D_NV; i += NT) { g0 += s.search[i] * s.mres[i]; h0 += s.search[i] * s.Mv[i]; }
        g0 = block_sum<NT>(s, g0, tid); h0 = block_sum<NT>(s, h0, tid);
        float alpha = 0.f, lo = 0.f, hi = 3.0e38f;
        for (int ls = 0; ls < 20; ls++) {
            float d1 = 0.f, d2 = 0.f;
            for (int k = tid; k < s.ncon; k += NT) {
                const float Dc = s.con_D[k];
                const float vn = s.jv3[3 * k], vt1 = s.jv3[3 * k + 1], vt2 = s.jv3[3 * k + 2];
                const float jn = s.jar3[3 * k] + alpha * vn, jt1 = s.jar3[3 * k + 1] + alpha * vt1, jt2 = s.jar3[3 * k + 2] + alpha * vt2;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float x = row_val(e, P.mu, jn, jt1, jt2);
                    if (x < 0.f) { float jv = row_val(e, P.mu, vn, vt1, vt2); d1 += Dc * x * jv; d2 += Dc * jv * jv; }
                }
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
        for (int k = tid; k < 3 * s.ncon; k += NT) s.jar3[k] += alpha * s.jv3[k];
        for (int j = tid; j < D_NU; j += NT) if (s.lim_sgn[j] != 0.f) s.lim_jar[j] += alpha * s.lim_jv[j];
        KP_SYNC();
        float newcost = primal_cost<NT>(s, P, s.qacc, tid);
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
    const int depth = tid < D_NB ? T.body_depth[tid] : -1;

    // ---- load: derived state first (the state the last forward pass ran on), then the real state
    for (int i = tid; i < D_NQ; i += NT) { s.qpos[i] = A.qpos_d[(size_t)env * D_NQ + i]; s.tq[i] = A.target_qpos ? A.target_qpos[(size_t)env * D_NQ + i] : 0.f; }
    for (int i = tid; i < D_NV; i += NT) {
        s.qvel[i] = A.qvel_d[(size_t)env * D_NV + i]; s.act[i] = A.action ? A.action[(size_t)env * D_NV + i] : 0.f;
        s.arm[i] = T.dof_armature[i]; s.dbody[i] = T.dof_body[i]; s.extra[i] = 0.f; s.qacc[i] = 0.f;
    }
    if (tid < D_NB) { s.bpar[tid] = (unsigned char)(T.body_parent[tid] < 0 ? 0 : T.body_parent[tid]); s.bsub[tid] = T.body_subtree[tid]; s.bdep[tid] = T.body_depth[tid]; }
    if (tid < 8) s.applied[tid] = 0.f;
    if (tid == 0) { s.ncon = 0; s.nlim = 0; s.flag = 0; }
    KP_SYNC();
    forward_kin_bias<NT>(s, T, P, depth, tid);
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
        // stale mode: the controller sees M / bias of the previous forward pass (cinert, cdof, bias still in LDS)
        if (P.stale) spd_torque_rfc<NT>(s, T, P, depth, tid);
        // ---- mj_forward at the current state
#pragma unroll
        for (int n = 0; n < (D_NQ + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NQ) qd_save_q[n] = s.qpos[i]; }
#pragma unroll
        for (int n = 0; n < (D_NV + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NV) qd_save_v[n] = s.qvel[i]; }
        forward_kin_bias<NT>(s, T, P, depth, tid);
        collide_plane<NT>(s, T, P, tid);
        make_constraint<NT>(s, T, P, tid);                 // needs sv = cvel: before any aba_solve
        if (!P.stale) spd_torque_rfc<NT>(s, T, P, depth, tid);
        for (int i = tid; i < D_NV; i += NT) {
            float f = -s.bias[i] + (i < 6 ? s.applied[i] : s.ctrl[i - 6]);
            s.smooth[i] = f; s.extra[i] = 0.f;
        }
        KP_SYNC();
        aba_solve<NT>(s, P, s.smooth, s.qacc_s, false, depth, tid);   // qacc_smooth = M^-1 qfrc_smooth; sv = its spatial accel
        niter_total += solve_constraints<NT>(s, P, depth, tid);
        maxcon = max(maxcon, s.ncon);
        // ---- semi-implicit Euler (mj_Euler, no damping)
        for (int i = tid; i < D_NV; i += NT) s.qvel[i] += P.h * s.qacc[i];
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
        forward_kin_bias<NT>(s, T, P, depth, tid);
    }
    // ---- store
    bool bad = false;
    for (int i = tid; i < D_NQ; i += NT) { float v = s.qpos[i]; bad |= !(fabsf(v) < 1e10f); if (A.n_substeps > 0) A.qpos[(size_t)env * D_NQ + i] = v; }
    for (int i = tid; i < D_NV; i += NT) { float v = s.qvel[i]; bad |= !(fabsf(v) < 1e10f); if (A.n_substeps > 0) { A.qvel[(size_t)env * D_NV + i] = v; A.warm[(size_t)env * D_NV + i] = s.qacc[i]; } }
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
