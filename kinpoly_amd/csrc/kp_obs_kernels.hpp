// kp_obs_kernels.hpp -- the Python-side arithmetic of HumanoidAREnv.step() around the physics:
//   k_step_kin     HumanoidAREnv.step_ar                kin_poly/envs/humanoid_ar_v1.py:216-241
//   k_target_fk    Humanoid.qpos_fk / forward_kinematics kin_poly/utils/numpy_smpl_humanoid.py:180-249
//   k_obs_cc       HumanoidEnv.get_full_obs_v1 (+ZFilter) uhc/envs/humanoid_im.py:144-233, zfilter.py:58-67
//   k_bquat        HumanoidEnv.get_body_quat            uhc/envs/humanoid_im.py:342-354
// Quaternion helpers restate uhc/khrylib/utils/math.py:102-198 and transformation.py (Gohlke).
#pragma once
#include "kp_device.hpp"

namespace kp {

// ---- reference quaternion helpers (w,x,y,z); these do NOT assume unit quaternions where the reference does not
__device__ __forceinline__ Q4 q_inverse(Q4 q) {  // quaternion_inverse: conj / dot
    float n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return Q4{q.w / n, -q.x / n, -q.y / n, -q.z / n};
}
__device__ __forceinline__ void q_matrix(Q4 q, float* m) {  // quaternion_matrix: normalises by dot, identity if ~0
    float n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    if (n < 8.881784197001252e-16f) { m[0] = m[4] = m[8] = 1.f; m[1] = m[2] = m[3] = m[5] = m[6] = m[7] = 0.f; return; }
    float s = sqrtf(2.0f / n);
    float w = q.w * s, x = q.x * s, y = q.y * s, z = q.z * s;
    m[0] = 1.f - y * y - z * z; m[1] = x * y - z * w; m[2] = x * z + y * w;
    m[3] = x * y + z * w; m[4] = 1.f - x * x - z * z; m[5] = y * z - x * w;
    m[6] = x * z - y * w; m[7] = y * z + x * w; m[8] = 1.f - x * x - y * y;
}
__device__ __forceinline__ V3 q_mul_vec(Q4 q, V3 v) { float m[9]; q_matrix(q, m); return mulmat(m, v); }            // quat_mul_vec
__device__ __forceinline__ V3 q_tmul_vec(Q4 q, V3 v) {                                                              // transform_vec(.., 'root')
    float m[9]; q_matrix(q, m);
    return V3{m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z};
}
__device__ __forceinline__ Q4 q_heading(Q4 q) {  // get_heading_q
    float n = sqrtf(q.w * q.w + q.z * q.z);
    return Q4{q.w / n, 0.f, 0.f, q.z / n};
}
__device__ __forceinline__ float heading_angle(Q4 q) {  // get_heading
    float w = q.w, z = q.z;
    if (z < 0.f) { w = -w; z = -z; }
    // 2 acos(w / |(w, z)|) with z >= 0, as the angle of the point (w, z): acos next to 1 would turn the 6e-8 of an fp32 cosine into 1e-4 rad
    // for a heading of 1e-3 rad (the reference works in fp64)
    return 2.0f * atan2f(fabsf(z), w);
}
__device__ __forceinline__ Q4 q_from_expmap(V3 e) {  // quat_from_expmap + quaternion_about_axis
    float angle = sqrtf(dot(e, e));
    V3 axis = angle < 1e-12f ? v3(1.f, 0.f, 0.f) : (1.0f / angle) * e;
    float sn, cs; sincosf(0.5f * angle, &sn, &cs);
    float ql = sqrtf(dot(axis, axis));
    float k = ql > 8.881784197001252e-16f ? sn / ql : 1.0f;
    return Q4{cs, axis.x * k, axis.y * k, axis.z * k};
}
// local hinge rotation: Gohlke quaternion_from_euler(tz, ty, tx, 'rzyx') = qz (x) qy (x) qx
__device__ __forceinline__ Q4 q_euler_rzyx(float tz, float ty, float tx) {
    float sz, cz, sy, cy, sx, cx;
    sincosf(0.5f * tz, &sz, &cz); sincosf(0.5f * ty, &sy, &cy); sincosf(0.5f * tx, &sx, &cx);
    return qmul(qmul(Q4{cz, 0.f, 0.f, sz}, Q4{cy, 0.f, sy, 0.f}), Q4{cx, sx, 0.f, 0.f});
}
// the reference's "bquat" quirk: quaternion_from_euler(a0, a1, a2) with default 'sxyz' on (tz, ty, tx)
__device__ __forceinline__ Q4 q_euler_sxyz(float ai, float aj, float ak) {
    float si, ci, sj, cj, sk, ck;
    sincosf(0.5f * ai, &si, &ci); sincosf(0.5f * aj, &sj, &cj); sincosf(0.5f * ak, &sk, &ck);
    float cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
    return Q4{cj * cc + sj * ss, cj * sc - sj * cs, cj * ss + sj * cc, cj * cs - sj * sc};
}

// ---------------------------------------------------------------- step_ar: one thread per env
__global__ void k_step_kin(int n, const float* __restrict__ qpos, const float* __restrict__ act, float* __restrict__ next_qpos, float dt) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* q = qpos + (size_t)e * D_NQ;
    const float* a = act + (size_t)e * 80;
    float* o = next_qpos + (size_t)e * D_NQ;
    Q4 rot = Q4{q[3], q[4], q[5], q[6]};
    Q4 hq = q_heading(rot);
    V3 linv = q_mul_vec(hq, v3(a[74], a[75], a[76]));
    o[0] = q[0] + linv.x * dt; o[1] = q[1] + linv.y * dt;
    o[2] = a[0];
    V3 angv = q_mul_vec(rot, v3(a[77], a[78], a[79]));
    Q4 nr = qmul(q_from_expmap(dt * angv), rot);
    o[3] = nr.w; o[4] = nr.x; o[5] = nr.y; o[6] = nr.z;
    for (int j = 0; j < D_NU; j++) o[7 + j] = a[5 + j];
}

// One frame of TrajARNet's kinematic roll-out (kin_poly/models/traj_ar_smpl_net.py:292-330): next_qpos = step(action) with the root quaternion
// normalised (:323-327) and qvel = get_qvel_fd_batch(qpos, next_qpos, dt) (kin_poly/utils/torch_utils.py:315-331: linear part, rotation vector of
// next (x) cur^-1 through rotation_from_quaternion's sqrt(1 - w^2) form, wrapped to (-pi, pi], expressed in the current root frame, joint part).
// One thread per env; replaces ~30 elementwise launches per frame of the batched init_context.
__global__ void k_kin_advance(int n, const float* __restrict__ qpos, const float* __restrict__ act, float dt, float* __restrict__ next_qpos, float* __restrict__ qvel) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* q = qpos + (size_t)e * D_NQ;
    const float* a = act + (size_t)e * 80;
    float* o = next_qpos + (size_t)e * D_NQ;
    float* v = qvel + (size_t)e * D_NV;
    const Q4 rot = Q4{q[3], q[4], q[5], q[6]};
    const Q4 hq = q_heading(rot);
    const V3 linv = q_mul_vec(hq, v3(a[74], a[75], a[76]));
    const float nx = q[0] + linv.x * dt, ny = q[1] + linv.y * dt, nz = a[0];
    const V3 angv = q_mul_vec(rot, v3(a[77], a[78], a[79]));
    Q4 nr = qmul(q_from_expmap(dt * angv), rot);
    const float nn = sqrtf(nr.w * nr.w + nr.x * nr.x + nr.y * nr.y + nr.z * nr.z);
    nr = Q4{nr.w / nn, nr.x / nn, nr.y / nn, nr.z / nn};
    o[0] = nx; o[1] = ny; o[2] = nz; o[3] = nr.w; o[4] = nr.x; o[5] = nr.y; o[6] = nr.z;
    const float idt = 1.0f / dt;
    v[0] = (nx - q[0]) * idt; v[1] = (ny - q[1]) * idt; v[2] = (nz - q[2]) * idt;
    // qrel = next (x) inverse(cur), inverse = conjugate / |cur|^2
    const float c2 = rot.w * rot.w + rot.x * rot.x + rot.y * rot.y + rot.z * rot.z;
    const Q4 qrel = qmul(nr, Q4{rot.w / c2, -rot.x / c2, -rot.y / c2, -rot.z / c2});
    // sin(acos(w)) and 2 acos(w) of the reference (fp64 there) without the fp32 cancellation of 1 - w^2: for the unit quaternion qrel the sine is |xyz|.
    // The reference's `sin < 1e-5 -> no rotation` branch (rotation_from_quaternion_batch, torch_utils.py:126-128) is dead code there: its safe_acos
    // clamps w to +-(1 - 1e-7) (:32-36), so the sine it tests is never below 4.5e-4 and a rotation too small for 1 - w to show comes out as
    // xyz / sin(acos(1 - 1e-7)) * 2 acos(1 - 1e-7) = 2 xyz -- which is what |xyz| and atan2 give; only the exact zero needs a guard.
    float sn = sqrtf(qrel.x * qrel.x + qrel.y * qrel.y + qrel.z * qrel.z);
    const bool small = !(sn > 0.0f);
    sn = fmaxf(sn, 1e-30f);
    const V3 axis = small ? v3(1.f, 0.f, 0.f) : v3(qrel.x / sn, qrel.y / sn, qrel.z / sn);
    float angle = small ? 0.0f : 2.0f * atan2f(sn, qrel.w);
    if (angle > 3.14159265358979f) angle -= 6.28318530717959f;
    if (angle < -3.14159265358979f) angle += 6.28318530717959f;
    const V3 rv = (angle * idt) * axis;
    // transform_vec_batch(rv, cur, 'root'): R(cur / |cur|)^T rv
    const float cn = sqrtf(c2);
    const float qw = rot.w / cn;
    const V3 u = v3(-rot.x / cn, -rot.y / cn, -rot.z / cn);
    const V3 t = 2.0f * cross(u, rv);
    const V3 wv = rv + qw * t + cross(u, t);
    v[3] = wv.x; v[4] = wv.y; v[5] = wv.z;
    for (int j = 0; j < D_NU; j++) { const float nj = a[5 + j]; o[7 + j] = nj; v[6 + j] = (nj - q[7 + j]) * idt; }
}

// ---------------------------------------------------------------- target FK: one wavefront per row, lane = body, level-synchronous chain in LDS
// (256-thread blocks = 4 rows).  Row loads and stores are coalesced; the 9-level parent chain goes through LDS.
struct TargetBufs { float *qpos, *wbpos, *wbquat, *bquat, *com; };
// KinStep (optional, K.act != null): the head of HumanoidAREnv.step in the same launch (humanoid_ar_v1.py:246-256) -- the prev_bquat /
// prev_hpos records of the CURRENT state (k_snapshot), next_qpos = step_ar(action) (k_step_kin) computed straight into the row the FK
// runs on; `tq` is then the current qpos.  Three launches of the env-step become one.
struct KinStep { const float *act, *xpos, *xquat; float *prev_bquat, *prev_hpos; float dt; };
__global__ __launch_bounds__(256) void k_target_fk(int n, const float* __restrict__ tq, const uint8_t* __restrict__ mask, TargetBufs B,
                                                    const float* __restrict__ body_pos, const float* __restrict__ body_ipos, const int8_t* __restrict__ parent,
                                                    const uint8_t* __restrict__ depth, KinStep K = KinStep{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f}) {
    __shared__ float sq[4][80], swq[4][D_NB * 4], swp[4][D_NB * 3];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + w;
    const bool live = e < n && !(mask && !mask[e]);
    if (!__syncthreads_or(live)) return;             // masked resets touch a few rows: a block whose four rows are all masked out leaves at once
    if (live) {
        const float* q = tq + (size_t)e * D_NQ;
        if (K.act) {
            const float* a = K.act + (size_t)e * 80;
            if (lane < D_NB) {                      // prev_bquat = get_body_quat(), prev_hpos = get_head() of the state before the step
                const Q4 r = lane == 0 ? Q4{q[3], q[4], q[5], q[6]} : q_euler_sxyz(q[7 + 3 * (lane - 1)], q[8 + 3 * (lane - 1)], q[9 + 3 * (lane - 1)]);
                float* o = K.prev_bquat + ((size_t)e * D_NB + lane) * 4;
                o[0] = r.w; o[1] = r.x; o[2] = r.y; o[3] = r.z;
                if (lane == 13) {
                    for (int k = 0; k < 3; k++) K.prev_hpos[(size_t)e * 7 + k] = K.xpos[(size_t)e * 72 + 39 + k];
                    for (int k = 0; k < 4; k++) K.prev_hpos[(size_t)e * 7 + 3 + k] = K.xquat[(size_t)e * 96 + 52 + k];
                }
            }
            if (lane == 32) {                       // step_ar's root part (:216-241), same arithmetic as k_step_kin
                const Q4 rot = Q4{q[3], q[4], q[5], q[6]};
                const V3 linv = q_mul_vec(q_heading(rot), v3(a[74], a[75], a[76]));
                const V3 angv = q_mul_vec(rot, v3(a[77], a[78], a[79]));
                const Q4 nr = qmul(q_from_expmap(K.dt * angv), rot);
                sq[w][0] = q[0] + linv.x * K.dt; sq[w][1] = q[1] + linv.y * K.dt; sq[w][2] = a[0];
                sq[w][3] = nr.w; sq[w][4] = nr.x; sq[w][5] = nr.y; sq[w][6] = nr.z;
            }
            for (int j = lane; j < D_NU; j += 64) sq[w][7 + j] = a[5 + j];
        } else {
            for (int i = lane; i < D_NQ; i += 64) sq[w][i] = q[i];
        }
    }
    __syncthreads();
    const int b = lane;
    Q4 wqb = Q4{1.f, 0.f, 0.f, 0.f}, lq = wqb, bq = wqb;
    V3 pos = v3(0.f, 0.f, 0.f);
    int dep = 99, p = 0;
    if (live && b < D_NB) {
        dep = depth[b]; p = parent[b] < 0 ? 0 : parent[b];
        const float* q = sq[w];
        if (b == 0) {
            Q4 rq = Q4{q[3], q[4], q[5], q[6]};
            const float rn = sqrtf(rq.w * rq.w + rq.x * rq.x + rq.y * rq.y + rq.z * rq.z);
            rq = Q4{rq.w / rn, rq.x / rn, rq.y / rn, rq.z / rn};  // fix_quat: root_rotations /= norm (no zero guard in the reference)
            wqb = rq; bq = rq; pos = v3(q[0], q[1], q[2]);
        } else {
            const float* ang = q + 7 + 3 * (b - 1);
            lq = q_euler_rzyx(ang[0], ang[1], ang[2]);
            bq = q_euler_sxyz(ang[0], ang[1], ang[2]);
        }
    }
    for (int lev = 0; lev < D_NLEV; lev++) {
        if (dep == lev) {
            if (b > 0) {
                const Q4 pq = Q4{swq[w][4 * p], swq[w][4 * p + 1], swq[w][4 * p + 2], swq[w][4 * p + 3]};
                pos = q_mul_vec(pq, ld3(body_pos + 3 * b)) + ld3(swp[w] + 3 * p);
                wqb = qmul(pq, lq);
            }
            swq[w][4 * b] = wqb.w; swq[w][4 * b + 1] = wqb.x; swq[w][4 * b + 2] = wqb.y; swq[w][4 * b + 3] = wqb.z;
            st3(swp[w] + 3 * b, pos);
        }
        __syncthreads();
    }
    if (live && b < D_NB) {
        if (B.bquat) { float* o = B.bquat + (size_t)e * 96 + 4 * b; o[0] = bq.w; o[1] = bq.x; o[2] = bq.y; o[3] = bq.z; }
        if (B.wbpos) st3(B.wbpos + (size_t)e * 72 + 3 * b, pos);
        if (B.wbquat) { float* o = B.wbquat + (size_t)e * 96 + 4 * b; o[0] = wqb.w; o[1] = wqb.x; o[2] = wqb.y; o[3] = wqb.z; }
        if (B.com) st3(B.com + (size_t)e * 72 + 3 * b, q_mul_vec(wqb, ld3(body_ipos + 3 * b)) + pos);
    }
    if (live && B.qpos) {
        float* oq = B.qpos + (size_t)e * D_NQ;
        for (int i = lane; i < D_NQ; i += 64) oq[i] = (i >= 3 && i < 7) ? swq[w][i - 3] : sq[w][i];     // root quaternion normalised
    }
}

// ---------------------------------------------------------------- backward of qpos -> wbpos (the end-effector term of compute_loss_lite)
// grad_qpos = (d wbpos / d qpos)^T grad_wbpos for the rows of a k_target_fk call (its wbpos / wbquat outputs are the saved forward state).
// A hinge of body b turns the strict descendants of b about pos_b, so with F_b = sum g_j and M_b = sum (pos_j - pos_b) x g_j over
// them: d/d theta_k = a_k . M_b (a_k = world axis of the hinge: R_p e_z, R_p Rz e_y, R_p Rz Ry e_x for the 'rzyx' order);
// root translation = sum of all g_j; root rotation, parametrised by the raw quaternion q (normalised inside the forward pass):
// torque tau = M_0 about pos_0 -> gradient (0, 2 tau) (x) u / |q| with u = q / |q|.  One wave per row, lane = body.
__global__ __launch_bounds__(256) void k_fk_wbpos_grad(int n, const float* __restrict__ qpos, const float* __restrict__ wbpos, const float* __restrict__ wbquat,
                                                        const float* __restrict__ gw, float* __restrict__ gq, const int8_t* __restrict__ parent,
                                                        const uint8_t* __restrict__ subtree) {
    __shared__ float sp[4][D_NB * 3], sg[4][D_NB * 3];
    const int w = threadIdx.x >> 6, b = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + w;
    const bool live = e < n;
    if (live && b < D_NB) {
        st3(sp[w] + 3 * b, ld3(wbpos + (size_t)e * 72 + 3 * b));
        st3(sg[w] + 3 * b, ld3(gw + (size_t)e * 72 + 3 * b));
    }
    __syncthreads();
    if (!live || b >= D_NB) return;
    const V3 pb = ld3(sp[w] + 3 * b);
    V3 F = v3(0.f, 0.f, 0.f), M = v3(0.f, 0.f, 0.f);
    const int nb = subtree[b];
    for (int j = b + 1; j < b + nb; j++) {
        const V3 g = ld3(sg[w] + 3 * j);
        F = F + g; M = M + cross(ld3(sp[w] + 3 * j) - pb, g);
    }
    const float* q = qpos + (size_t)e * D_NQ;
    float* o = gq + (size_t)e * D_NQ;
    if (b == 0) {
        st3(o, F + ld3(sg[w]));
        const float qn = sqrtf(q[3] * q[3] + q[4] * q[4] + q[5] * q[5] + q[6] * q[6]);
        const float* uq = wbquat + (size_t)e * 96;
        const Q4 G = qmul(Q4{0.f, 2.f * M.x, 2.f * M.y, 2.f * M.z}, Q4{uq[0], uq[1], uq[2], uq[3]});
        o[3] = G.w / qn; o[4] = G.x / qn; o[5] = G.y / qn; o[6] = G.z / qn;
    } else {
        const int p = parent[b];
        const float* pq = wbquat + (size_t)e * 96 + 4 * p;
        float R[9];
        q_matrix(Q4{pq[0], pq[1], pq[2], pq[3]}, R);
        const float tz = q[7 + 3 * (b - 1)], ty = q[8 + 3 * (b - 1)];
        float sz, cz, sy, cy;
        sincosf(tz, &sz, &cz); sincosf(ty, &sy, &cy);
        const V3 az = mulmat(R, v3(0.f, 0.f, 1.f));
        const V3 ay = mulmat(R, v3(-sz, cz, 0.f));                  // Rz e_y
        const V3 ax = mulmat(R, v3(cz * cy, sz * cy, -sy));         // Rz Ry e_x
        o[7 + 3 * (b - 1)] = dot(az, M); o[8 + 3 * (b - 1)] = dot(ay, M); o[9 + 3 * (b - 1)] = dot(ax, M);
    }
}

// ---------------------------------------------------------------- get_body_quat: thread per (env, body)
__global__ void k_bquat(int n, const float* __restrict__ qpos, float* __restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * D_NB) return;
    int e = t / D_NB, b = t - e * D_NB;
    const float* q = qpos + (size_t)e * D_NQ;
    Q4 r;
    if (b == 0) r = Q4{q[3], q[4], q[5], q[6]};
    else r = q_euler_sxyz(q[7 + 3 * (b - 1)], q[8 + 3 * (b - 1)], q[9 + 3 * (b - 1)]);
    float* o = out + (size_t)t * 4;
    o[0] = r.w; o[1] = r.x; o[2] = r.y; o[3] = r.z;
}

// ---------------------------------------------------------------- get_full_obs_v1 (+ ZFilter): one wave per env, staged in LDS
struct ObsCcArgs {
    int n;
    const float *qpos, *qvel, *xpos, *xquat, *xipos;   // sim state (qpos/qvel fresh, x* stale)
    const float *t_qpos, *t_wbpos, *t_wbquat, *t_com;  // target dict
    float br_inv[4];
    const float *zf_mean, *zf_std; float clip;
    float* out;
};
__global__ __launch_bounds__(64) void k_obs_cc(ObsCcArgs A) {
    __shared__ float ob[784];
    const int e = blockIdx.x, tid = threadIdx.x;
    if (e >= A.n) return;
    const float* qpos = A.qpos + (size_t)e * D_NQ;
    const float* qvel = A.qvel + (size_t)e * D_NV;
    const float* tq = A.t_qpos + (size_t)e * D_NQ;
    const Q4 rq = Q4{qpos[3], qpos[4], qpos[5], qpos[6]};
    const Q4 binv = Q4{A.br_inv[0], A.br_inv[1], A.br_inv[2], A.br_inv[3]};
    const Q4 crq = qmul(rq, binv);                      // remove_base_rot
    const Q4 hq = q_heading(crq);
    const Q4 trq = qmul(Q4{tq[3], tq[4], tq[5], tq[6]}, binv);
    const V3 root = v3(qpos[0], qpos[1], qpos[2]);
    if (tid == 0) {
        ob[0] = hq.w; ob[1] = hq.x; ob[2] = hq.y; ob[3] = hq.z;
        Q4 dh = qmul(q_inverse(hq), crq);               // de_heading(curr_root_quat)
        ob[78] = qpos[2]; ob[79] = dh.w; ob[80] = dh.x; ob[81] = dh.y; ob[82] = dh.z;
        Q4 dq = qmul(trq, q_inverse(crq));
        ob[152] = tq[2] - qpos[2]; ob[153] = dq.w; ob[154] = dq.x; ob[155] = dq.y; ob[156] = dq.z;
        V3 v = q_tmul_vec(crq, q_tmul_vec(rq, v3(qvel[0], qvel[1], qvel[2])));  // transformed twice (:150, :173)
        ob[226] = v.x; ob[227] = v.y; ob[228] = v.z;
        float rel_h = heading_angle(trq) - heading_angle(crq);
        if (rel_h > 3.14159265358979f) rel_h -= 6.28318530717959f;
        if (rel_h < -3.14159265358979f) rel_h += 6.28318530717959f;
        ob[301] = rel_h;
        V3 rp = q_tmul_vec(crq, v3(trq.w, trq.x, trq.y) - root);  // sic: quaternion components used as a position (:187)
        ob[302] = rp.x; ob[303] = rp.y;
    }
    for (int i = tid; i < 74; i += 64) ob[4 + i] = tq[2 + i];
    for (int i = tid; i < 69; i += 64) { ob[83 + i] = qpos[7 + i]; ob[157 + i] = tq[7 + i] - qpos[7 + i]; }
    for (int i = tid; i < 72; i += 64) ob[229 + i] = qvel[3 + i];
    if (tid < D_NB) {
        const int b = tid;
        const float* xp = A.xpos + (size_t)e * 72 + 3 * b;
        const float* xi = A.xipos + (size_t)e * 72 + 3 * b;
        const float* xq = A.xquat + (size_t)e * 96;
        V3 cj = ld3(xp), ci = ld3(xi);
        V3 tj = ld3(A.t_wbpos + (size_t)e * 72 + 3 * b), tc = ld3(A.t_com + (size_t)e * 72 + 3 * b);
        // transform_vec_batch returns a (3, 24) array that the reference ravel()s: component-major blocks
        V3 p1 = q_tmul_vec(crq, cj - root), p2 = q_tmul_vec(crq, tj - cj), p3 = q_tmul_vec(crq, ci - root), p4 = q_tmul_vec(crq, tc - ci);
        ob[304 + b] = p1.x; ob[328 + b] = p1.y; ob[352 + b] = p1.z;
        ob[376 + b] = p2.x; ob[400 + b] = p2.y; ob[424 + b] = p2.z;
        ob[448 + b] = p3.x; ob[472 + b] = p3.y; ob[496 + b] = p3.z;
        ob[520 + b] = p4.x; ob[544 + b] = p4.y; ob[568 + b] = p4.z;
        const float* twq = A.t_wbquat + (size_t)e * 96 + 4 * b;
        Q4 tqt = Q4{twq[0], twq[1], twq[2], twq[3]};
        Q4 cq = (xq[0] == 0.f) ? tqt : Q4{xq[4 * b], xq[4 * b + 1], xq[4 * b + 2], xq[4 * b + 3]};
        Q4 r1 = qmul(q_inverse(hq), cq), r2 = qmul(q_inverse(cq), tqt);
        float* o1 = ob + 592 + 4 * b; o1[0] = r1.w; o1[1] = r1.x; o1[2] = r1.y; o1[3] = r1.z;
        float* o2 = ob + 688 + 4 * b; o2[0] = r2.w; o2[1] = r2.x; o2[2] = r2.y; o2[3] = r2.z;
    }
    __syncthreads();
    for (int i = tid; i < 784; i += 64) {
        float v = ob[i];
        if (A.zf_mean) {
            v = (v - A.zf_mean[i]) / (A.zf_std[i] + 1e-8f);
            if (A.clip > 0.f) v = fminf(fmaxf(v, -A.clip), A.clip);
        }
        A.out[(size_t)e * 784 + i] = v;
    }
}

}  // namespace kp
