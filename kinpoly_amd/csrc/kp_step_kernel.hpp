// kp_step_kernel.hpp -- the fused control-step kernel: n_substeps x { stable-PD torque, residual
// force, mj_step-equivalent forward dynamics with soft contact (hulls, floor, free objects), semi-implicit Euler }.
//
// Replaces HOT LOOP C of the reference: HumanoidEnv.do_simulation (uhc/envs/humanoid_im.py:506-533)
//   compute_torque / compute_desired_accel   uhc/envs/humanoid_im.py:418-480
//   rfc_implicit                              uhc/envs/humanoid_im.py:497-504
//   sim.step()  (MuJoCo mj_step)              uhc/envs/humanoid_im.py:527         [MJ-ext]
// The arithmetic follows oracle/kp_oracle.c (fp64) in fp32, organised MATRIX-FREE for a wavefront:
//   * every spatial quantity is expressed in ONE world-aligned frame at the root body origin, so tree
//     recursions need no coordinate transforms: parents and children simply add;
//   * kinematics, velocities and bias forces (RNE) come out of one level-synchronous pass (9 levels);
//   * every linear solve -- (M + K_d h) for the stable-PD controller, M + J^T D J for the Newton steps of the
//     contact solver (M alone when no constraint row exists; qacc_smooth is never formed otherwise, see
//     solve_constraints_direct) -- is an articulated-body (ABA) pass:
//     leaves->root articulated inertia + bias, root->leaves accelerations.  K_d h and joint-limit terms
//     enter as extra joint armature, active contact rows as a per-body 6x6 "contact inertia" D w w^T.
//     The joint-space mass matrix is never formed or factorised (the reference materialises a dense
//     105x105 M and a Cholesky factor per substep);
//   * J v is read off the spatial accelerations the ABA forward pass leaves behind, J^T f and M v are a
//     body-wrench subtree sum projected on the dofs; the solver's Gauss term lives in body form (spatial
//     accelerations), so one projection per Newton iteration suffices;
//   * the first Newton factorisation of a substep walks every tree level; the later ones reuse its factors on the levels no active constraint reaches;
//   * the RNE bias forces are never projected on the dofs: they stay body wrenches and enter the passes as articulated bias force.
// Launch forms: kp_step_kernel (one workgroup = one wavefront per env and control step), kp_forward_kernel (sim.forward() only) and
// kp_step_queue_kernel (the same step_body run by resident wavefronts that pull (env, few substeps) jobs from a FIFO in HBM, used
// when there are more envs than wavefront slots; bit-identical results; the finishing job hands its successor the next stable-PD torque, and a wave
// keeps an env it finds heavy or that started late).  The functions are templates over the LDS layout (kp_device.hpp): floor scenes' queue launches run on
// EnvLdsLean (12 envs per CU) and hand a job with more contacts than it holds to kp_step_overflow_kernel (full layout); same arithmetic, same bits.
#pragma once
#include <type_traits>

#include "kp_device.hpp"
#include "kp_collide.hpp"

namespace kp {

#define KP_SYNC() __syncthreads()

// An opaque copy of a lane index.  Everything the compiler derives from threadIdx.x alone (per-lane table addresses, the Lane8 schedule)
// is invariant over the substep loop and over the job loop of the queue kernel, so LLVM hoists it all to the kernel entry and then has
// ~150 such values live across the whole kernel: with the narrow phases' own demand that is far beyond the 256 VGPRs two waves per SIMD
// leave each wave, and the allocator parks them in scratch and reloads them at every use (839 scratch instructions in the object kernel,
// tools/micro/spill_report.py).  Re-deriving them from a laundered index per substep / per solve costs a few address adds and keeps
// their live ranges inside the phase that uses them.
__device__ __forceinline__ int kp_launder(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int kp_launder_uniform(int x) { asm volatile("" : "+s"(x)); return x; }      // the same for a wave-uniform value (stays in an SGPR)

// the object extension of an env's LDS block.  Only reached under OBJ (a compile-time constant), where SL is EnvLdsObj itself; the other layouts never execute it
template <class SL> __device__ __forceinline__ EnvLdsObj& as_obj(SL& s) { return reinterpret_cast<EnvLdsObj&>(s); }
template <class SL> __device__ __forceinline__ const EnvLdsObj& as_obj(const SL& s) { return reinterpret_cast<const EnvLdsObj&>(s); }

struct StepArgs {
    DevTables T;
    Params P;
    int n_envs, n_substeps;
    // per-env state (HBM, row-major [N, dim])
    float *qpos, *qvel, *qpos_d, *qvel_d, *warm;
    float *warm3;              // [N][75] lean layout: the same vector between the solves of one job (step_body)
    float *warm2;              // [N][75] hand-over of the previous-but-one solution between the jobs of a control step (warm_extrap)
    float warm_extrap;         // beta of the extrapolated Newton start a_{k-1} + beta (a_{k-1} - a_{k-2}); 0 = MuJoCo's plain warm start
    const float *target_qpos, *action;
    const uint8_t* env_mask;  // optional: envs with mask==0 are skipped
    // outputs: kinematics of the last forward pass (x_14 in stale mode)
    float *xpos, *xquat, *xipos;
    int* diag;  // [N,4]: ncon (last substep), newton iterations (sum), flags | substeps whose Newton solve ended at the iteration cap << 8, max ncon | factorisations << 8
    unsigned long long* prof;  // optional [N,8] shader-clock cycles per phase (kp_sim_phase_cycles)
    float* dbg_contacts;       // optional [N, 1 + 64 * 9]: contacts of the last substep's collision pass (kp_sim_contacts; tests)
    const float* geoms;        // OBJ kernels: [N, D_MAXGEOM, 17] world-frame static geoms
    const int* ngeom;          // OBJ kernels: [N]
    // OBJ kernels, dynamic free objects: slot -> object index of the scene (-1 = empty), free-joint state of all objects
    const signed char* obj_slot;   // [N, D_MAXOBJ]
    float *obj_qpos, *obj_qvel, *obj_warm;   // [N, 35], [N, 30], [N, 6 * D_MAXOBJ]
    float *obj_warm2;                         // [N, 6 * D_MAXOBJ] the objects' a_{k-2} between the jobs of a control step (warm_extrap)
    // launch order (longest-processing-time first): workgroup i simulates env order[i]; cost[env] = shader-clock cycles >> 10 this
    // launch spent on env.  Both optional.  The result of an env does not depend on the workgroup that computes it.
    const int* order;
    unsigned* cost;
    // job-queue launch (kp_step_queue_kernel): a control step of an env = n_parts jobs of part_sub[] substeps handed from wave to
    // wave through HBM.  jobq [n_envs * n_parts] entries (env | part << 24, 0xFFFFFFFF = not yet published), jobctr = {head, tail, stalled}
    unsigned* jobq;
    unsigned* jobctr;
    int lean_cap;              // contacts a lean job accepts before it hands the env over (<= EnvLdsLean::MAXCON; model option lean_max_contacts)
    unsigned* ovfq;            // [n_envs] lean queue kernel: (env | part << 24) of the jobs whose contacts did not fit its layout; jobctr[64] = count, [65] = claimed (kp_step_overflow_kernel)
    float* spd_next;           // [N, 80]: qfrc_applied ++ qfrc_actuator (78 floats) of an env's NEXT substep, computed by the job that ran the substep before it
    int n_parts;
    int queue_heavy;           // > 0: a wave that finds its env heavy (this job's cycles per substep > queue_heavy % of the launch's running mean) runs the env's next job itself
    int queue_fence;           // 1: the hand-over is a release (publish) / acquire (consume) pair at agent scope instead of relaxed sc1 accesses + s_waitcnt
    int queue_late;            // 1: a wave whose env's FIRST job came from beyond the resident slots (it started late) runs that env's later jobs itself, at once
    int queue_prio;            // issue priority (s_setprio) of a wave: 0 none; 1 envs known to be heavy; 2 by the env's remaining jobs; 3 by its remaining substeps, re-set at every substep
    int order_valid;           // the first jobs were queued longest-env-first
    unsigned long long part_sub_lo, part_sub_hi;   // substeps of job 0 .. 15, one byte each (sum = n_substeps); packed so that no lookup indexes the kernel argument
};

// wave-wide sum without LDS traffic: xor butterflies inside each 16-lane row with DPP (quad_perm, row_half_mirror,
// row_mirror), then the row sums travel up with row_bcast:15 (rows 1, 3 += lane 15 of the row below) and row_bcast:31 (rows 2, 3 +=
// lane 31), so lane 63 holds (r2 + r3) + (r0 + r1) and one v_readlane hands it to every lane.
__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // xor 1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // xor 2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // half mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)); // row_bcast:15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false)); // row_bcast:31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// wave-wide minimum, same exchange pattern (every lane gets the result)
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
    const int vi = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}

__device__ __forceinline__ float wave_max_f(float v) { return -wave_min(-v); }

template <int NT, class SL>
__device__ __forceinline__ float block_sum(SL& s, float v, int tid) {
    v = wave_sum(v);
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

// contact frame (n, t1, t2).  Floor-only kernels: the constant plane frame; OBJ kernels: mju_makeFrame of the stored normal.
struct Frame { V3 n, t1, t2; };
template <bool OBJ, class SL>
__device__ __forceinline__ Frame contact_frame(const SL& s, int c) {
    if (!OBJ) return Frame{v3(0.f, 0.f, 1.f), v3(0.f, 1.f, 0.f), v3(-1.f, 0.f, 0.f)};
    const float* cn = as_obj(s).con_n + 3 * c;
    const V3 n = ld3(cn);
    V3 y = fabsf(n.y) < 0.5f ? v3(0.f, 1.f, 0.f) : v3(0.f, 0.f, 1.f);
    y = y - dot(n, y) * n;
    y = (1.0f / sqrtf(dot(y, y))) * y;
    return Frame{n, y, cross(n, y)};
}
__device__ __forceinline__ V3 frame_comp(const Frame& f, V3 v) { return v3(dot(f.n, v), dot(f.t1, v), dot(f.t2, v)); }
__device__ __forceinline__ V3 frame_world(const Frame& f, V3 c) { return c.x * f.n + c.y * f.t1 + c.z * f.t2; }

// ---------------------------------------------------------------- kinematics + velocities + bias
// Phases, so that the level-serial chain stays short:
//   0. half-angle sin/cos of all 69 hinge angles, one lane per hinge (scratch: s.U, free outside the ABA passes), then per
//      body (parallel) the local rotation qz qy qx and the second / third hinge axis in the parent frame;
//   K. two short level-synchronous chains (lane = body) with a body-parallel block between them: world pose; then motion axes cdof
//      and each body's own velocity terms; then spatial velocity cvel and the velocity-product acceleration cacc
//      (mj_kinematics + mj_comVel + the forward half of mj_rne);
//   B. body-parallel (24 lanes at once): COM, world inertia about o, body wrench fb = I a + v x* I v.  qfrc_bias itself
//      (backward half of mj_rne: subtree sums + projection on the dofs) is never formed: the solves take fb as bias force.
template <int NT, class SL>
__device__ __forceinline__ void forward_kin_bias(SL& s, const DevTables& T, const Params& P, int depth, V3 bpos, int tid) {
    float* sc = s.U;              // scratch (free outside the ABA passes): [0, 138) half-angle sin/cos, [144, 384) per-body joint frames
    float* jf = s.U + 144;        // per body: local rotation qz (x) qy (x) qx (4), second axis R(qz) e_y (3), third axis R(qz qy) e_x (3)
    for (int i = tid; i < D_NU; i += NT) { float sn, cs; sincosf(0.5f * s.qpos[7 + i], &sn, &cs); sc[2 * i] = sn; sc[2 * i + 1] = cs; }
    KP_SYNC();
    if (tid >= 1 && tid < D_NB) {         // body-parallel: everything of the three hinges that does not depend on the parent
        const int j0 = 3 * (tid - 1);
        const Q4 qz = Q4{sc[2 * j0 + 1], 0.f, 0.f, sc[2 * j0]}, qy = Q4{sc[2 * j0 + 3], 0.f, sc[2 * j0 + 2], 0.f}, qx = Q4{sc[2 * j0 + 5], sc[2 * j0 + 4], 0.f, 0.f};
        const Q4 qzy = qmul(qz, qy);
        const Q4 ql = qmul(qzy, qx);
        const V3 a1 = qrot(qz, v3(0.f, 1.f, 0.f)), a2 = qrot(qzy, v3(1.f, 0.f, 0.f));
        float* f = jf + 10 * tid;
        f[0] = ql.w; f[1] = ql.x; f[2] = ql.y; f[3] = ql.z; st3(f + 4, a1); st3(f + 7, a2);
    }
    KP_SYNC();
    if (tid == 0) {                       // root (level 0): free joint
        const Q4 q = qnormalize(Q4{s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]});
        s.qpos[3] = q.w; s.qpos[4] = q.x; s.qpos[5] = q.y; s.qpos[6] = q.z;
        const V3 pos = ld3(s.qpos);
        float R[9]; q2mat(q, R);
        const V3 vl = ld3(s.qvel), wb = ld3(s.qvel + 3);
        const V3 ww = mulmat(R, wb);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float* c = s.cdof + 6 * k;
            c[0] = c[1] = c[2] = 0.f; c[3] = k == 0; c[4] = k == 1; c[5] = k == 2;
            float* r = s.cdof + 6 * (3 + k);
            r[0] = R[k]; r[1] = R[3 + k]; r[2] = R[6 + k]; r[3] = r[4] = r[5] = 0.f;
        }
        st3(s.xpos, pos);
        s.xquat[0] = q.w; s.xquat[1] = q.x; s.xquat[2] = q.y; s.xquat[3] = q.z;
        sts6(s.sv, S6{ww, vl}); sts6(s.sa, S6{v3(0.f, 0.f, 0.f), v3(-P.gx, -P.gy, -P.gz) + cross(vl, ww)});
    }
    KP_SYNC();
    // K1. pose chain (level-synchronous, lane = body): world position and orientation only
#pragma nounroll
    for (int lev = 1; lev < D_NLEV; lev++) {
        if (depth == lev) {
            const int b = tid;
            const int p = s.bpar[b];
            const Q4 q = Q4{s.xquat[4 * p], s.xquat[4 * p + 1], s.xquat[4 * p + 2], s.xquat[4 * p + 3]};
            const V3 ppos = ld3(s.xpos + 3 * p);
            const float* f = jf + 10 * b;
            const Q4 ql = Q4{f[0], f[1], f[2], f[3]};
            const V3 pos = ppos + qrot(q, bpos);
            const Q4 qn = qnormalize(qmul(q, ql));
            st3(s.xpos + 3 * b, pos);
            s.xquat[4 * b] = qn.w; s.xquat[4 * b + 1] = qn.x; s.xquat[4 * b + 2] = qn.y; s.xquat[4 * b + 3] = qn.z;
        }
        KP_SYNC();
    }
    // K2. body-parallel: motion axes of the three hinges (about o) and the body's own share of the velocity chain.  With
    // run_j = sum_{i<j} qd_i cdof_i the level-serial recursion  cv += qd_j cdof_j,  ca += qd_j (cv x cdof_j)  splits into
    //   cv_b = cv_p + dv,   ca_b = ca_p + cv_p x dv + loc,   dv = run_3,   loc = sum_j qd_j (run_j x cdof_j)
    // (x = cross_motion, bilinear), so that only two 6-vector updates per body remain on the chain.
    S6 dv = S6{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)}, loc = dv;
    if (tid >= 1 && tid < D_NB) {
        const int b = tid;
        const int p = s.bpar[b];
        const int d0 = 6 + 3 * (b - 1);
        const V3 o = ld3(s.xpos);
        const Q4 q = Q4{s.xquat[4 * p], s.xquat[4 * p + 1], s.xquat[4 * p + 2], s.xquat[4 * p + 3]};
        const V3 r = o - ld3(s.xpos + 3 * b);
        const float qds[3] = {s.qvel[d0], s.qvel[d0 + 1], s.qvel[d0 + 2]};
        const float* f = jf + 10 * b;
        const V3 a1 = ld3(f + 4), a2 = ld3(f + 7);
        float R[9];
        q2mat(q, R);                                           // parent rotation: the three hinge axes are R e_z, R a1, R a2
        const V3 axes[3] = {v3(R[2], R[5], R[8]), mulmat(R, a1), mulmat(R, a2)};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const S6 cd = S6{axes[j], cross(axes[j], r)};
            sts6(s.cdof + 6 * (d0 + j), cd);
            if (j > 0) loc = loc + qds[j] * cross_motion(dv, cd);
            dv = dv + qds[j] * cd;
        }
    }
    KP_SYNC();
    // K3. velocity chain (level-synchronous): spatial velocity cvel and velocity-product acceleration cacc
#pragma nounroll
    for (int lev = 1; lev < D_NLEV; lev++) {
        if (depth == lev) {
            const int b = tid;
            const int p = s.bpar[b];
            const S6 cvp = lds6(s.sv + 6 * p), cap = lds6(s.sa + 6 * p);
            sts6(s.sv + 6 * b, cvp + dv);
            sts6(s.sa + 6 * b, cap + cross_motion(cvp, dv) + loc);
        }
        KP_SYNC();
    }
    if (tid < D_NB) {
        const int b = tid;
        const V3 o = ld3(s.xpos), pos = ld3(s.xpos + 3 * b);
        float R[9];
        q2mat(Q4{s.xquat[4 * b], s.xquat[4 * b + 1], s.xquat[4 * b + 2], s.xquat[4 * b + 3]}, R);
        const S6 cv = lds6(s.sv + 6 * b), ca = lds6(s.sa + 6 * b);
        const V3 xi = pos + mulmat(R, ld3(T.body_ipos + 3 * b));
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
        const V3 rr = xi - o;
        const float mass = T.body_mass[b], r2 = dot(rr, rr);
        float* ci = s.cinert + 10 * b;
        ci[0] = W[0] + mass * (r2 - rr.x * rr.x); ci[1] = W[4] + mass * (r2 - rr.y * rr.y); ci[2] = W[8] + mass * (r2 - rr.z * rr.z);
        ci[3] = W[1] - mass * rr.x * rr.y; ci[4] = W[2] - mass * rr.x * rr.z; ci[5] = W[5] - mass * rr.y * rr.z;
        ci[6] = mass * rr.x; ci[7] = mass * rr.y; ci[8] = mass * rr.z; ci[9] = mass;
        const S6 f = inert_mul(ci, ca) + cross_force(cv, inert_mul(ci, cv));
        sts6(s.fb + 6 * b, f);         // stays a body wrench: the articulated-body passes fold it into their bias force,
    }                                  // which is the backward half of mj_rne (subtree sums + projection) done for free
    KP_SYNC();
}

// ---------------------------------------------------------------- articulated-body solve, 8 lanes per body
// Lane layout inside a tree level: tid = 8 * slot + r; slot = index of the body within the level (<= 5 bodies),
// r = row of the 6x6 articulated inertia (rows 6, 7 are zero padding).  A lane keeps its row in "XOR order":
// register k holds column r ^ KX[k], KX = {0,1,2,3,7,6,5,4}.  With that order the three DPP exchanges
// quad_perm[1,0,3,2] (xor 1), quad_perm[2,3,0,1] (xor 2) and row_half_mirror (xor 7) implement both the
// 8-lane sum and the 8-lane all-gather (7 moves) with results already aligned to the row registers, so the
// rank-1 updates  IA -= U U^T / D  need no LDS round trip.
__device__ __forceinline__ float dpp_x1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_x2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_x7(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float sum8(float v) { v += dpp_x1(v); v += dpp_x2(v); v += dpp_x7(v); return v; }
__device__ __forceinline__ void gather8(float v, float* g) {
    g[0] = v; g[1] = dpp_x1(v); g[2] = dpp_x2(g[0]); g[3] = dpp_x2(g[1]);
    g[4] = dpp_x7(g[0]); g[5] = dpp_x7(g[1]); g[6] = dpp_x7(g[2]); g[7] = dpp_x7(g[3]);
}
__device__ __forceinline__ float rcp_nr(float d) { float r = __builtin_amdgcn_rcpf(d); return r * (2.0f - d * r); }

struct Lane8 {
    int slot, r;
    int col[8];      // column held by register k
    int ciidx[8];    // index into the 10-float body inertia giving IA[r][col[k]] (times cisgn)
    float cisgn[8];
    int idx21[8];    // index into the 21-float symmetric storage
    unsigned long long sb, sp, sc0, sc1, sc2;  // 5 bits per level: body, parent, children (31 = none) of this lane's slot
    unsigned multi;  // bit lev set: some body of that level has more than one child (wave-uniform; only levels 0 and 3 of the SMPL tree)
    // The per-lane schedule comes ready-made from the host (kp_sim.hip: build_lane8_table, 28 dwords per lane): seven 16-byte loads
    // and a few bit-field extracts, cheap enough to redo at the top of every solve instead of keeping 45 registers alive between them.
    //   words 0..9   sb, sp, sc0, sc1, sc2 (64 bit each: 5 bits per tree level = body / parent / children of this lane's slot;
    //                31 = no body at that level, absent child -> the zero record 24)
    //   word  10     multi (bit lev: some body of that level has more than one child; only levels 0 and 3 of the SMPL tree)
    //   words 12..19 per register k: col | ciidx << 4 | idx21 << 8     (column r ^ KX[k], index into the 10-float body inertia, index into the 21-float symmetric storage)
    //   words 20..27 per register k: cisgn as float
    __device__ __forceinline__ void init(int tid, const uint32_t* __restrict__ tab) {
        slot = tid >> 3; r = tid & 7;
        const uint4* q = reinterpret_cast<const uint4*>(tab + 28 * (tid & 63));
        const uint4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4], w5 = q[5], w6 = q[6];
        sb = (unsigned long long)w0.x | ((unsigned long long)w0.y << 32); sp = (unsigned long long)w0.z | ((unsigned long long)w0.w << 32);
        sc0 = (unsigned long long)w1.x | ((unsigned long long)w1.y << 32); sc1 = (unsigned long long)w1.z | ((unsigned long long)w1.w << 32);
        sc2 = (unsigned long long)w2.x | ((unsigned long long)w2.y << 32); multi = w2.z;
        if (tid >= 64) {            // threads_per_env = 128 / 256: only the first wavefront serves the tree passes (31 = no body, 24 = zero record)
            sb = sp = 0x1FFFFFFFFFFFull;
            sc0 = sc1 = sc2 = 0x18C6318C6318ull;      // 24 in every 5-bit field of the 9 levels
        }
        const unsigned pk[8] = {w3.x, w3.y, w3.z, w3.w, w4.x, w4.y, w4.z, w4.w};
        const unsigned sg[8] = {w5.x, w5.y, w5.z, w5.w, w6.x, w6.y, w6.z, w6.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            col[k] = (int)(pk[k] & 7u); ciidx[k] = (int)((pk[k] >> 4) & 15u); idx21[k] = (int)((pk[k] >> 8) & 31u);
            cisgn[k] = __builtin_bit_cast(float, sg[k]);
        }
    }
};

// out = (M + diag(s.extra) [+ J^T D_active J])^-1 rhs.  Leaves s.sv[b] = sum over ancestor dofs cdof_d out_d
// (the spatial "acceleration" of every body induced by out).  rhs/out are LDS vectors (may alias).
// eliminate the three dofs d0+2, d0+1, d0 of one body from its articulated inertia (row r in XOR order)
#ifndef KP_PK_ELIM
#define KP_PK_ELIM 1
#endif
template <class SL>
__device__ __forceinline__ void aba_elim3(SL& s, const Lane8& L, const float* rhs, int d0, bool store_ok, float* IAx, float& pA) {
    const int r = L.r;
    const bool rowok = r < 6;
    float sxa[3][8], dsc[3], rh[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int d = d0 + j;
        gather8(rowok ? s.cdof[6 * d + r] : 0.f, sxa[j]);    // lane r fetches its own component, the 8-lane all-gather lands in XOR order
        if constexpr (SL::LEAN) dsc[j] = s.extra[d]; else dsc[j] = s.arm[d] + s.extra[d];      // lean layout: extra holds armature + extra (see step_body)
        rh[j] = rhs[d];
    }
    float Uo[3], Do[3], uo[3];
#if KP_PK_ELIM
    // Round 5 (VERDICT r4 #5a): the two 8-term blocks of a joint elimination -- the row product U_r = sum_k IA[r][k] s[k] and the rank-1 update
    // IA[r][k] -= (U_r / D) U[k] -- as hand-placed packed fp32 operations (v_pk_mul_f32 / v_pk_fma_f32 on register pairs), everything else unchanged.
    // The blanket SLP vectoriser lost to register pairing on this kernel (build.py OPT_FLAGS); here the operands of a pair are adjacent by construction:
    // 576 scalar fp32 instructions become 324 packed ones, no extra v_mov, 208 -> 212 VGPRs (floor kernel), launch 2.665 -> 2.611 ms on the metric's
    // workload in three A/B pairs (profiles/r05/pk_elim_ab.log), objects 4.79 -> 4.78 ms.  -DKP_PK_ELIM=0 builds the scalar form (the pairwise row
    // product sums in another order: the two builds are not bit-identical; every parity test and sweep was re-run on this one).
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 IA2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) IA2[k] = f2{IAx[2 * k], IAx[2 * k + 1]};
#pragma unroll
    for (int j = 2; j >= 0; j--) {
        f2 acc = IA2[0] * f2{sxa[j][0], sxa[j][1]};
#pragma unroll
        for (int k = 1; k < 4; k++) acc = __builtin_elementwise_fma(IA2[k], f2{sxa[j][2 * k], sxa[j][2 * k + 1]}, acc);
        const float Ur = acc.x + acc.y;
        const float sr = sxa[j][0];
        const float dsum = sum8(sr * Ur), usum = sum8(sr * pA);
        const float D = dsc[j] + dsum, u = rh[j] - usum;
        const float Dinv = rcp_nr(D);
        float Ux[8];
        gather8(Ur, Ux);
        const float ur = -(Ur * Dinv);
        const f2 ur2 = f2{ur, ur};
#pragma unroll
        for (int k = 0; k < 4; k++) IA2[k] = __builtin_elementwise_fma(ur2, f2{Ux[2 * k], Ux[2 * k + 1]}, IA2[k]);
        pA += Ur * (u * Dinv);
        Uo[j] = Ur; Do[j] = Dinv; uo[j] = u;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { IAx[2 * k] = IA2[k].x; IAx[2 * k + 1] = IA2[k].y; }
#else
#pragma unroll
    for (int j = 2; j >= 0; j--) {
        float Ur = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) Ur += IAx[k] * sxa[j][k];
        const float sr = sxa[j][0];
        const float dsum = sum8(sr * Ur), usum = sum8(sr * pA);
        const float D = dsc[j] + dsum, u = rh[j] - usum;
        const float Dinv = rcp_nr(D);
        float Ux[8];
        gather8(Ur, Ux);
        const float ur = Ur * Dinv;
#pragma unroll
        for (int k = 0; k < 8; k++) IAx[k] -= ur * Ux[k];
        pA += Ur * (u * Dinv);
        Uo[j] = Ur; Do[j] = Dinv; uo[j] = u;
    }
#endif
    if (store_ok && rowok) {
#pragma unroll
        for (int j = 0; j < 3; j++) s.U[6 * (d0 + j) + r] = Uo[j];
    }
    if (store_ok && r == 0) {
#pragma unroll
        for (int j = 0; j < 3; j++) { s.Dinv[d0 + j] = Do[j]; s.uj[d0 + j] = uo[j]; }
    }
}

template <class SL>
__device__ __forceinline__ float aba_fwd3(SL& s, const Lane8& L, float* out, int d0, bool store_ok, float a) {
    const int r = L.r;
    const int rc = r < 6 ? r : 5;
    const float rmask = r < 6 ? 1.f : 0.f;
    float Ud[3], sd[3], ujd[3], Did[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int d = d0 + j;
        Ud[j] = rmask * s.U[6 * d + rc]; sd[j] = s.cdof[6 * d + rc];
        ujd[j] = s.uj[d]; Did[j] = s.Dinv[d];
    }
    float qo[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float qdd = (ujd[j] - sum8(Ud[j] * a)) * Did[j];
        qo[j] = qdd;
        a += qdd * sd[j];
    }
    if (store_ok && r == 0) {
#pragma unroll
        for (int j = 0; j < 3; j++) out[d0 + j] = qo[j];
    }
    return a;
}

// root->leaves pass: joint accelerations from (U, 1/D, u) and the parent's spatial acceleration; leaves them in sv
template <class SL>
__device__ __forceinline__ void aba_forward(SL& s, const Lane8& L, float* out, int lev_max = D_NLEV - 1) {
    const int r = L.r;
    const bool rowok = r < 6;
#pragma nounroll
    for (int lev = 0; lev <= lev_max; lev++) {
        const int sh = 5 * lev;
        const int bq = (int)((L.sb >> sh) & 31ull), par = (int)((L.sp >> sh) & 31ull);
        const bool active = bq != 31;
        const int b = active ? bq : 0;
        float a = (rowok && par != 31) ? s.sv[6 * (par == 31 ? 0 : par) + r] : 0.f;
        if (lev == 0) {
            a = aba_fwd3(s, L, out, 0, active, a);
            a = aba_fwd3(s, L, out, 3, active, a);
        } else {
            a = aba_fwd3(s, L, out, b == 0 ? 6 : 6 + 3 * (b - 1), active, a);
        }
        if (active && rowok) s.sv[6 * b + r] = a;
        KP_SYNC();
    }
}

// bit lev set: the body this lane serves at tree level lev has contacts (con_start is fixed for the substep, so the Newton
// factorisations test a register instead of reading two LDS words per level)
template <class SL>
__device__ __forceinline__ unsigned contact_levels(const SL& s, const Lane8& L) {
    unsigned m = 0;
#pragma unroll
    for (int lev = 0; lev < D_NLEV; lev++) {
        const int bq = (int)((L.sb >> (5 * lev)) & 31ull), b = bq != 31 ? bq : 0;
        if (bq != 31 && s.con_start[b + 1] > s.con_start[b]) m |= 1u << lev;
    }
    return m;
}

template <int NT, bool OBJ, class SL>
// lev_clean: tree levels >= lev_clean carry no active contact row and no active joint limit, so their articulated inertias, U and
// 1/D are the ones the substep's first Newton factorisation (which walks every level; same M, no extra armature there) left in LDS:
// only the bias-force half runs there.
__device__ __forceinline__ void aba_solve(SL& s, const Params& P, const Lane8& L, const float* rhs, float* out, bool contact_inertia, int tid, int lev_clean = D_NLEV, const float* bwrench = nullptr,
                                          unsigned conlev = 0xFFFFFFFFu) {
    const int r = L.r;
    const bool rowok = r < 6;
    if constexpr (SL::LEAN) {              // pAa shares its words with J search there: the zero record absent children read is written anew for this solve
        if (tid < 6) s.pAa[6 * 24 + tid] = 0.f;
        KP_SYNC();
    }
#pragma nounroll
    for (int lev = D_NLEV - 1; lev >= 0; lev--) {
        const int sh = 5 * lev;
        const int bq = (int)((L.sb >> sh) & 31ull), c0 = (int)((L.sc0 >> sh) & 31ull), c1 = (int)((L.sc1 >> sh) & 31ull), c2 = (int)((L.sc2 >> sh) & 31ull);
        const bool active = bq != 31;
        const int b = active ? bq : 0;
        if (lev >= lev_clean) {                                    // lev_clean >= 1: the root level is never clean
            const float rmask = rowok ? 1.f : 0.f;
            const int rc = rowok ? r : 5, pr = 6 * 24;
            float pA = s.pAa[rowok ? 6 * c0 + r : pr];
            if ((L.multi >> lev) & 1u) pA += s.pAa[rowok ? 6 * c1 + r : pr] + s.pAa[rowok ? 6 * c2 + r : pr];
            const int d0 = 6 + 3 * (b == 0 ? 0 : b - 1);
            float uo[3];
#pragma unroll
            for (int j = 2; j >= 0; j--) {
                const int d = d0 + j;
                const float u = rhs[d] - sum8(rmask * s.cdof[6 * d + rc] * pA);
                pA += rmask * s.U[6 * d + rc] * (u * s.Dinv[d]);
                uo[j] = u;
            }
            if (active && r == 0) {
#pragma unroll
                for (int j = 0; j < 3; j++) s.uj[d0 + j] = uo[j];
            }
            if (active && rowok) s.pAa[6 * b + r] = pA;
            KP_SYNC();
            continue;
        }
        // ---- one LDS round: body inertia row + children rows (the per-dof operands are fetched by aba_elim3 in the same round)
        float IAx[8];
        float pA;
        {
            const float* ci = s.cinert + 10 * b;
            const float* r0 = s.IAa + 22 * c0;
            const int pr = rowok ? r : 6 * 24;                 // padding rows read the zero record
#pragma unroll
            for (int k = 0; k < 8; k++) IAx[k] = L.cisgn[k] * ci[L.ciidx[k]] + r0[L.idx21[k]];
            pA = s.pAa[rowok ? 6 * c0 + r : pr];
            if ((L.multi >> lev) & 1u) {                       // second / third child: only the levels where some body branches
                const float* r1 = s.IAa + 22 * c1; const float* r2 = s.IAa + 22 * c2;
#pragma unroll
                for (int k = 0; k < 8; k++) IAx[k] += r1[L.idx21[k]] + r2[L.idx21[k]];
                pA += s.pAa[rowok ? 6 * c1 + r : pr] + s.pAa[rowok ? 6 * c2 + r : pr];
            }
            if (bwrench) pA += (rowok ? 1.f : 0.f) * bwrench[6 * b + (rowok ? r : 5)];
        }
        if (contact_inertia && active && ((conlev >> lev) & 1u)) {     // conlev: this lane's body at this level carries contacts (contact_levels)
            // active pyramid rows come from con_act (active_set_changed ran on this iterate); the next contact's operands are
            // requested before this one's arithmetic, so the loop pays one LDS round trip, not one per contact
            const int c0 = s.con_start[b], c1 = s.con_start[b + 1];
            if (c1 > c0) {
                const V3 o = ld3(s.xpos);
                float Krow[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (OBJ) {
                    // object kernels: M_c = D F G F^T of every contact's active rows is already in LDS for this iterate (con_prepare, lane = contact, which
                    // also feeds the object rows of the Hessian), so a row of K = P M_c P^T is one symmetric 3 x 3 product and a cross product -- no
                    // mju_makeFrame, no frame round trip per contact, body row and factorisation
                    const float* cM = as_obj(s).cM;
                    V3 pn = ld3(s.con_pos + 3 * c0);
                    float mn[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) mn[k] = cM[6 * c0 + k];
                    for (int c = c0; c < c1; c++) {
                        const V3 p = pn - o;
                        float m[6];
#pragma unroll
                        for (int k = 0; k < 6; k++) m[k] = mn[k];
                        const int cn = min(c + 1, c1 - 1);
                        pn = ld3(s.con_pos + 3 * cn);
#pragma unroll
                        for (int k = 0; k < 6; k++) mn[k] = cM[6 * cn + k];
                        V3 Pr;  // row r of P = [[p]x ; 1]:  row r of K = [p x q ; q] with q = M_c P_r
                        if (r == 0) Pr = v3(0.f, -p.z, p.y); else if (r == 1) Pr = v3(p.z, 0.f, -p.x); else if (r == 2) Pr = v3(-p.y, p.x, 0.f);
                        else Pr = v3(r == 3 ? 1.f : 0.f, r == 4 ? 1.f : 0.f, r == 5 ? 1.f : 0.f);
                        const V3 q = v3(m[0] * Pr.x + m[3] * Pr.y + m[4] * Pr.z, m[3] * Pr.x + m[1] * Pr.y + m[5] * Pr.z, m[4] * Pr.x + m[5] * Pr.y + m[2] * Pr.z);
                        const V3 pq = cross(p, q);
                        Krow[0] += pq.x; Krow[1] += pq.y; Krow[2] += pq.z; Krow[3] += q.x; Krow[4] += q.y; Krow[5] += q.z;
                    }
                } else {
                const float mu = P.mu, mu2 = P.mu * P.mu;
                V3 pn = ld3(s.con_pos + 3 * c0);
                float Dn = s.con_D[c0];
                unsigned an = s.con_act[c0];
                for (int c = c0; c < c1; c++) {
                    const V3 p = pn - o;
                    const float Dc = Dn;
                    const unsigned am = an;
                    const int cn = min(c + 1, c1 - 1);
                    pn = ld3(s.con_pos + 3 * cn); Dn = s.con_D[cn]; an = s.con_act[cn];
                    const float a0 = (am & 1u) ? 1.f : 0.f, a1 = (am & 2u) ? 1.f : 0.f, a2 = (am & 4u) ? 1.f : 0.f, a3 = (am & 8u) ? 1.f : 0.f;
                    // G = sum_e a_e dir_e dir_e^T with dir_e = n +- mu t1, n +- mu t2; in frame coordinates (n, t1, t2):
                    const float gnn = a0 + a1 + a2 + a3, g11 = mu2 * (a0 + a1), g22 = mu2 * (a2 + a3), gn1 = mu * (a0 - a1), gn2 = mu * (a2 - a3);
                    const Frame fr = contact_frame<OBJ>(s, c);
                    V3 Pr;  // row r of P = [[p]x ; 1]:  K = D P G P^T, row r = [p x q ; q] with q = D (P_r G)
                    if (r == 0) Pr = v3(0.f, -p.z, p.y); else if (r == 1) Pr = v3(p.z, 0.f, -p.x); else if (r == 2) Pr = v3(-p.y, p.x, 0.f);
                    else Pr = v3(r == 3 ? 1.f : 0.f, r == 4 ? 1.f : 0.f, r == 5 ? 1.f : 0.f);
                    const V3 pf = frame_comp(fr, Pr);
                    const V3 q = frame_world(fr, v3(Dc * (gnn * pf.x + gn1 * pf.y + gn2 * pf.z), Dc * (gn1 * pf.x + g11 * pf.y), Dc * (gn2 * pf.x + g22 * pf.z)));
                    const V3 pq = cross(p, q);
                    Krow[0] += pq.x; Krow[1] += pq.y; Krow[2] += pq.z; Krow[3] += q.x; Krow[4] += q.y; Krow[5] += q.z;
                }
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int c = L.col[k];
                    float v = Krow[0];
                    v = c == 1 ? Krow[1] : v; v = c == 2 ? Krow[2] : v; v = c == 3 ? Krow[3] : v; v = c == 4 ? Krow[4] : v; v = c == 5 ? Krow[5] : v;
                    if (rowok && c < 6) IAx[k] += v;
                }
            }
        }
        if (lev == 0) {
            aba_elim3(s, L, rhs, 3, active, IAx, pA);
            aba_elim3(s, L, rhs, 0, active, IAx, pA);
        } else {
            aba_elim3(s, L, rhs, 6 + 3 * (b - 1) < 6 ? 6 : 6 + 3 * (b - 1), active, IAx, pA);
        }
        if (active && rowok) {
#pragma unroll
            for (int k = 0; k < 8; k++) {      // upper triangle only; the other lanes' copies go to a scrap word instead of eight exec-masked stores
                float* dst = (L.col[k] < 6 && r <= L.col[k]) ? s.IAa + 22 * b + L.idx21[k] : s.red + 7;
                *dst = IAx[k];
            }
            s.pAa[6 * b + r] = pA;
        }
        KP_SYNC();
    }
    aba_forward(s, L, out, D_NLEV - 1);
}

// out = H^-1 (rhs + J_body^T wrench) with the factorisation (U, 1/D) the last aba_solve left in LDS: the bias-force half of the
// leaves->root pass only (no inertia updates), then the usual root->leaves pass.  rhs / wrench ([24][6], about o) may be null.
// lev_max < D_NLEV - 1: the caller knows that rhs and wrench vanish on every body below tree level lev_max and only needs the
// accelerations (sv) / out entries down to that level: the deeper levels would carry exact zeros up and are skipped in both halves.
template <class SL>
__device__ __forceinline__ void aba_resolve(SL& s, const Lane8& L, const float* rhs, const float* wrench, float* out, int lev_max = D_NLEV - 1) {
    const int r = L.r;
    const bool rowok = r < 6;
    const int rc = rowok ? r : 5;
    const float rmask = rowok ? 1.f : 0.f;
    if constexpr (SL::LEAN) {              // see aba_solve
        if (threadIdx.x < 6) s.pAa[6 * 24 + threadIdx.x] = 0.f;
        KP_SYNC();
    }
    if (lev_max < D_NLEV - 1) {            // the skipped children hand up zero bias forces
        for (int i = threadIdx.x; i < 24 * 6; i += 64) s.pAa[i] = 0.f;
        KP_SYNC();
    }
#pragma nounroll
    for (int lev = lev_max; lev >= 0; lev--) {
        const int sh = 5 * lev;
        const int bq = (int)((L.sb >> sh) & 31ull), c0 = (int)((L.sc0 >> sh) & 31ull), c1 = (int)((L.sc1 >> sh) & 31ull), c2 = (int)((L.sc2 >> sh) & 31ull);
        const bool active = bq != 31;
        const int b = active ? bq : 0;
        const int pr = 6 * 24;
        float pA = s.pAa[rowok ? 6 * c0 + r : pr];
        if ((L.multi >> lev) & 1u) pA += s.pAa[rowok ? 6 * c1 + r : pr] + s.pAa[rowok ? 6 * c2 + r : pr];
        if (wrench) pA -= rmask * wrench[6 * b + rc];
        const int nrounds = lev == 0 ? 2 : 1;
        for (int rd = 0; rd < nrounds; rd++) {
            const int d0 = lev == 0 ? (rd == 0 ? 3 : 0) : 6 + 3 * (b == 0 ? 0 : b - 1);
            float uo[3];
#pragma unroll
            for (int j = 2; j >= 0; j--) {
                const int d = d0 + j;
                const float u = (rhs ? rhs[d] : 0.f) - sum8(rmask * s.cdof[6 * d + rc] * pA);
                pA += rmask * s.U[6 * d + rc] * (u * s.Dinv[d]);
                uo[j] = u;
            }
            if (active && r == 0) {
#pragma unroll
                for (int j = 0; j < 3; j++) s.uj[d0 + j] = uo[j];
            }
        }
        if (active && rowok) s.pAa[6 * b + r] = pA;
        KP_SYNC();
    }
    aba_forward(s, L, out, lev_max);
}

// ---------------------------------------------------------------- stable-PD torque + residual force (reference controller)
template <int NT, bool OBJ, class SL>
// tq / act: this env's rows of the PD target and the action in HBM (null: zeros); read here once per substep instead of living in LDS
__device__ __forceinline__ void spd_torque_rfc(SL& s, const DevTables& T, const Params& P, const Lane8& L8, int tid, const float* __restrict__ tq, const float* __restrict__ act) {
    float* const epv = SL::LEAN ? s.lim_jar - 6 : s.search;      // the position error of dof i >= 6 (lean layout: search is the solve's own vector, lim_jar is dead here)
    for (int i = tid; i < D_NV; i += NT) {
        float ep = 0.f, kp = 0.f, kd = 0.f;
        if (i >= 6) {
            int j = i - 6;
            float q = s.qpos[i + 1], base = tq ? tq[i + 1] : 0.f;
            // the reference's 2 pi unwrap loops (humanoid_im.py:447-452) in closed form: no trip count that depends on the data, so a
            // non-finite or absurd target cannot spin the wavefront (it yields a non-finite state, which diag flags)
            const float dq = base - q;
            if (dq > 3.14159265358979f) base -= 6.28318530717959f * ceilf((dq - 3.14159265358979f) * 0.159154943091895f);
            else if (dq < -3.14159265358979f) base += 6.28318530717959f * ceilf((-dq - 3.14159265358979f) * 0.159154943091895f);
            float target = base + (act ? act[j] : 0.f) * T.ascale[j];
            kp = T.kp[j]; kd = T.kd[j];
            ep = q + s.qvel[i] * P.h - target;
        }
        float kdh = kd * P.h;                         // (M + K_d dt): K_d dt is extra joint armature
        if constexpr (SL::LEAN) { asm volatile("" : "+v"(kdh)); kdh = T.dof_armature[i] + kdh; }      // the rounded product, then the sum aba_elim3 forms on the full layout (no fused multiply-add across the two)
        s.extra[i] = kdh;
        if (!SL::LEAN || i >= 6) epv[i] = ep;
        s.x[i] = -kp * ep - kd * s.qvel[i];
    }
    KP_SYNC();
    aba_solve<NT, OBJ>(s, P, L8, s.x, s.x, false, tid, D_NLEV, s.fb);
    for (int j = tid; j < D_NU; j += NT) {
        int i = j + 6;
        float tau = -T.kp[j] * epv[i] - T.kd[j] * (s.qvel[i] + s.x[i] * P.h);
        float lim = T.tlim[j];
        s.ctrl[j] = fminf(fmaxf(tau, -lim), lim);
    }
    if (tid == 0) {  // rfc_implicit
        Q4 cq = qmul(Q4{s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]}, Q4{P.br_inv[0], P.br_inv[1], P.br_inv[2], P.br_inv[3]});
        float hn = sqrtf(cq.w * cq.w + cq.z * cq.z);
        Q4 hq = Q4{cq.w / hn, 0.f, 0.f, cq.z / hn};
        float ar[6];
#pragma unroll
        for (int k = 0; k < 6; k++) ar[k] = act ? act[69 + k] : 0.f;
        V3 f = qrot(hq, v3(ar[0] * P.rfc_scale, ar[1] * P.rfc_scale, ar[2] * P.rfc_scale));
        float vf[6] = {f.x, f.y, f.z, ar[3] * P.rfc_scale, ar[4] * P.rfc_scale, ar[5] * P.rfc_scale};
#pragma unroll
        for (int k = 0; k < 6; k++) s.applied[k] = fminf(fmaxf(vf[k], -P.rfc_lim), P.rfc_lim);
    }
    KP_SYNC();
}

// ---------------------------------------------------------------- collision (first wavefront)
// signed distance of world point x to a box (type 0) / z-axis cylinder (type 1) geom record
__device__ __forceinline__ float geom_sdf(const float* g, V3 x) {
    const float* R = g + 7;
    const V3 r = x - ld3(g + 4);
    const V3 l = v3(R[0] * r.x + R[3] * r.y + R[6] * r.z, R[1] * r.x + R[4] * r.y + R[7] * r.z, R[2] * r.x + R[5] * r.y + R[8] * r.z);
    if (g[0] == 0.f) {
        const V3 q = v3(fabsf(l.x) - g[1], fabsf(l.y) - g[2], fabsf(l.z) - g[3]);
        const V3 o = v3(fmaxf(q.x, 0.f), fmaxf(q.y, 0.f), fmaxf(q.z, 0.f));
        return sqrtf(dot(o, o)) + fminf(fmaxf(q.x, fmaxf(q.y, q.z)), 0.f);
    }
    const float qr = sqrtf(l.x * l.x + l.y * l.y) - g[1], qz = fabsf(l.z) - g[2];
    const float orr = fmaxf(qr, 0.f), oz = fmaxf(qz, 0.f);
    return sqrtf(orr * orr + oz * oz) + fminf(fmaxf(qr, qz), 0.f);
}
__device__ __forceinline__ float geom_rbound(const float* g) { return g[0] == 0.f ? sqrtf(g[1] * g[1] + g[2] * g[2] + g[3] * g[3]) : sqrtf(g[1] * g[1] + g[2] * g[2]); }

// one lane appends a contact: vertex-side entity A (0..23 hull, 24 + k object slot), surface-side entity B (-1 world / static geom),
// normal pointing from B's geom into A's
template <bool OBJ, class SL>
// B: entity carrying the surface (-1 floor, -2 - g static geom g, 24 + k object slot k); its invweight0 is looked up by make_constraint
__device__ __forceinline__ void put_contact(SL& s, int c, V3 pos, float dist, V3 nrm, int A, int B) {
    st3(s.con_pos + 3 * c, pos);
    s.con_D[c] = dist; s.con_body[c] = (unsigned char)A;          // con_D holds the distance until make_constraint
    if constexpr (OBJ) {
        EnvLdsObj& so = as_obj(s);
        st3(so.con_n + 3 * c, nrm); so.con_b2[c] = (signed char)B;
    }
}

// LDS scratch of the MPR query (witnesses of the portal vertices, 30 doubles) inside s.U, which is free outside the ABA passes:
// U[0, 96) box-box polygon, U[96, 152) contact records of a pair, U[160, 220) this, U[232, 247) hull record of the support functor
template <class SL>
__device__ __forceinline__ double* mpr_scratch(SL& s) {
    static_assert(offsetof(SL, U) % 8 == 0, "s.U must be 8-byte aligned for the fp64 MPR scratch");
    return reinterpret_cast<double*>(s.U + 160);
}

// mj_collision of the scene.  Mid phase in parallel: lane = hull body (then lane = object geom) tests all its targets (bit 0 = floor,
// bit j = geom j - 1) with bounding spheres; the serial part visits only the pairs that passed, in the oracle's order (entity-major,
// floor first), so contact indices and the con_start[] grouping are the oracle's.  Narrow phases: kp_collide.hpp.
template <int NT, bool OBJ, class SL>
__device__ __forceinline__ void collide(SL& s, const DevTables& T, const Params& P, int tid) {
    if (tid < 64) {
        int ncon = 0, next_b = 0;
        bool over = false;           // wave-uniform: a pair found more contacts than the layout has room for (the lean layout reports it: ncon = MAXCON + 1)
        const int ngeom = OBJ ? as_obj(s).ngeom : 0;
        unsigned mybits = 0;
        if (tid < D_NB && P.contact) {
            const V3 xb = ld3(s.xpos + 3 * tid);
            const float rb = T.body_rbound[tid];
            if (!(xb.z - rb > P.margin)) mybits = 1u;
            if constexpr (OBJ) {
                for (int gi = 0; gi < ngeom; gi++) {
                    const float* g = as_obj(s).geom + 17 * gi;
                    const V3 dx = xb - ld3(g + 4);
                    if (sqrtf(dot(dx, dx)) - rb - geom_rbound(g) > P.margin) continue;
                    // second, exact-safe cull: the hull lies inside the sphere (body origin, rbound) and a signed distance is 1-Lipschitz, so
                    // no point of it is closer to the primitive than sdf(origin) - rbound; only pairs that cannot touch are dropped (the
                    // 0.1 mm slack keeps fp32 rounding of this test away from the margin).  Saves the MPR query for most near misses.
                    if (geom_sdf(g, xb) - rb > P.margin + 1e-4f) continue;
                    mybits |= 2u << gi;
                }
            }
        }
        unsigned long long bodies = __ballot(mybits != 0u);
        while (bodies) {
            const int b = __ffsll((long long)bodies) - 1;
            bodies &= bodies - 1ull;
            unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)mybits, b);
            if (tid == 0) for (int bb = next_b; bb <= b; bb++) s.con_start[bb] = ncon;
            next_b = b + 1;
            const int vadr = T.vert_adr[b], nvb = T.vert_adr[b + 1] - vadr;
            const V3 xb = ld3(s.xpos + 3 * b);
            float R[9];
            q2mat(Q4{s.xquat[4 * b], s.xquat[4 * b + 1], s.xquat[4 * b + 2], s.xquat[4 * b + 3]}, R);
            V3 v = v3(0.f, 0.f, 0.f), xw = v3(0.f, 0.f, 3.0e38f);
            if (tid < nvb) { v = ld3(T.verts + 3 * (vadr + tid)); xw = xb + mulmat(R, v); }
            while (bits) {
                const int gi = __ffs((int)bits) - 2;           // -1 = floor
                bits &= bits - 1u;
                if (gi < 0) {
                    // mjc_PlaneConvex: the support vertex of -normal (deepest, first on ties) makes the first contact; then its hull-graph
                    // neighbours in graph order, while fewer than maxplanemesh (3) contacts exist, each within the margin and not closer
                    // than tolplanemesh * geom_rbound (0.3 rbound) to the first contact's position (addplanemesh)
                    const float dmin = wave_min(xw.z);
                    if (dmin > P.margin) continue;
                    const int idx = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(xw.z == dmin)) - 1);
                    const int n0 = T.vert_nbr_adr[vadr + idx], n1 = T.vert_nbr_adr[vadr + idx + 1];
                    // lane k < deg takes neighbour k of the support vertex (one coalesced load of the list), fetches that vertex from the
                    // lane that holds it, and ranks itself among the qualifying neighbours in list order with a ballot
                    const int deg = n1 - n0;
                    const int j = tid < deg ? (int)T.vert_nbr[n0 + tid] : 0;
                    const V3 xj = v3(__shfl(xw.x, j, 64), __shfl(xw.y, j, 64), __shfl(xw.z, j, 64));
                    const V3 x0 = v3(bcast_lane(xw.x, idx), bcast_lane(xw.y, idx), bcast_lane(xw.z, idx));
                    const V3 dj = xj - v3(x0.x, x0.y, x0.z - 0.5f * x0.z);           // vertex against the first contact's position
                    const float tolr = P.pm_tol * T.mesh_rbound[b];
                    const bool ok = tid < deg && !(xj.z > P.margin) && !(dot(dj, dj) < tolr * tolr);
                    const unsigned long long m = __ballot(ok);
                    const int rank = __popcll(m & ((1ull << tid) - 1ull));
                    const int extra = P.pm_max - 1;
                    const int cnt = 1 + min(__popcll(m), extra);
                    const int room = SL::MAXCON - ncon;
                    over |= cnt > room;
                    if (tid == idx && room > 0) put_contact<OBJ>(s, ncon, v3(xw.x, xw.y, xw.z - 0.5f * xw.z), xw.z, v3(0.f, 0.f, 1.f), b, -1);
                    if (ok && rank < extra && 1 + rank < room) put_contact<OBJ>(s, ncon + 1 + rank, v3(xj.x, xj.y, xj.z - 0.5f * xj.z), xj.z, v3(0.f, 0.f, 1.f), b, -1);
                    ncon += min(cnt, max(room, 0));
                } else if constexpr (OBJ) {
                    // mjc_Convex (libccd MPR): geom 1 = the box / cylinder, geom 2 = the hull; one contact, normal into the hull
                    EnvLdsObj& so = as_obj(s);
                    const float* g = so.geom + 17 * gi;
                    // Third, exact-safe cull before the MPR query (round 4: on the envs that end the `objects` launch 13 of 15 queries per substep
                    // found nothing and cost 5 - 6 k cycles each): a separating PLANE.  lane = hull vertex (xw, already in registers); if every
                    // vertex lies beyond one face plane of the box -- or beyond a cap plane / a tangent plane of the cylinder -- by more than the
                    // margin, the hull is farther than the margin from the primitive and libccd would return "no intersection" for the shapes
                    // inflated by margin / 2 each.  The 0.1 mm slack keeps the fp32 evaluation of the test away from the margin (the query itself
                    // runs in fp64 on the same fp32 poses); a pair that is not separated by one of these planes goes to the query as before.
                    {
                        const float* Rg = g + 7;
                        const V3 dv = xw - ld3(g + 4);
                        const float px = Rg[0] * dv.x + Rg[3] * dv.y + Rg[6] * dv.z, py = Rg[1] * dv.x + Rg[4] * dv.y + Rg[7] * dv.z, pz = Rg[2] * dv.x + Rg[5] * dv.y + Rg[8] * dv.z;
                        const bool has = tid < nvb;
                        const float lim = P.margin + 1e-4f;
                        bool sep;
                        if (g[0] == 0.f) {
                            const float hx = g[1] + lim, hy = g[2] + lim, hz = g[3] + lim;
                            sep = __ballot(has && !(px > hx)) == 0ull || __ballot(has && !(px < -hx)) == 0ull || __ballot(has && !(py > hy)) == 0ull ||
                                  __ballot(has && !(py < -hy)) == 0ull || __ballot(has && !(pz > hz)) == 0ull || __ballot(has && !(pz < -hz)) == 0ull;
                        } else {
                            const float hz = g[2] + lim;
                            sep = __ballot(has && !(pz > hz)) == 0ull || __ballot(has && !(pz < -hz)) == 0ull;
                            if (!sep) {          // tangent plane facing the hull: the unit radial direction u from the cylinder's axis towards the body origin
                                const V3 cb = xb - ld3(g + 4);
                                const float ux = Rg[0] * cb.x + Rg[3] * cb.y + Rg[6] * cb.z, uy = Rg[1] * cb.x + Rg[4] * cb.y + Rg[7] * cb.z;
                                const float un = sqrtf(ux * ux + uy * uy);
                                if (un > 1e-6f) sep = __ballot(has && !((px * ux + py * uy) > (g[1] + lim) * un)) == 0ull;
                            }
                        }
                        if (sep) continue;
                    }
                    const GeomSupport ga(g);
                    float* hrec = s.U + 232;                              // hull record of the support functor (LDS scratch: s.U is free outside the ABA passes)
                    hull_support_store(hrec, xb, xb + mulmat(R, ld3(T.body_ipos + 3 * b)), R);       // centre = the body's COM (xipos)
                    const HullSupport hb(hrec, v, tid < nvb);
                    Contact c;
                    if (convex_pair(ga, hb, P.margin, c, mpr_scratch(s)) && ncon < SL::MAXCON) {
                        if (tid == 0) put_contact<OBJ>(s, ncon, c.pos, c.dist, c.n, b, so.gobj[gi] < 0 ? -2 - gi : D_NB + so.gobj[gi]);
                        ncon++;
                    }
                }
            }
        }
        if (tid == 0) for (int bb = next_b; bb <= D_NB; bb++) s.con_start[bb] = ncon;
        if constexpr (OBJ) {
            // dynamic objects in slot order: every geom against the floor, then against the geoms of the objects in higher slots
            EnvLdsObj& so = as_obj(s);
            float* rec = s.U + 96;                             // contact records of one pair; s.U is free outside the ABA passes
            int slot_done = 0;
            unsigned gbits = 0;
            if (tid >= so.ngeom_static && tid < ngeom && P.contact) {
                const float* g = so.geom + 17 * tid;
                const V3 gp = ld3(g + 4);
                const float gra = geom_rbound(g);
                const int ka = so.gobj[tid];
                if (!(gp.z - gra > P.margin)) gbits = 1u;
                for (int gb = 0; gb < ngeom; gb++) {
                    const float* h = so.geom + 17 * gb;
                    if (so.gobj[gb] < 0 || so.gobj[gb] <= ka) continue;
                    const V3 dx = gp - ld3(h + 4);
                    if (sqrtf(dot(dx, dx)) - gra - geom_rbound(h) > P.margin) continue;
                    gbits |= 2u << gb;
                }
            }
            unsigned long long geoms = __ballot(gbits != 0u);
            while (geoms) {
                const int ga = __ffsll((long long)geoms) - 1;
                geoms &= geoms - 1ull;
                unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)gbits, ga);
                const float* g = so.geom + 17 * ga;
                const int ka = so.gobj[ga];
                while (slot_done < ka) { slot_done++; if (tid == 0) s.con_start[D_NB + slot_done] = ncon; }
                while (bits) {
                    const int gb = __ffs((int)bits) - 2;
                    bits &= bits - 1u;
                    const float* h = gb < 0 ? nullptr : so.geom + 17 * gb;
                    int n = 0, entB = -1; float sgn = 1.f;
                    if (gb < 0) {
                        if (g[0] == 0.f) {
                            // mjc_PlaneBox: lane = corner (bit 0 / 1 / 2 = +x / +y / +z); corners that point up or lie beyond the margin
                            // are skipped, the first four of the rest (in index order) make contacts
                            const int i = tid & 7;
                            const V3 corner = mulmat(g + 7, v3((i & 1) ? g[1] : -g[1], (i & 2) ? g[2] : -g[2], (i & 4) ? g[3] : -g[3]));
                            const float dist = g[6], ldist = corner.z;
                            const bool ok = tid < 8 && !(dist + ldist > P.margin || ldist > 0.f);
                            const unsigned long long m = __ballot(ok);
                            const int rank = __popcll(m & ((1ull << tid) - 1ull));
                            n = min(__popcll(m), 4);
                            if (ok && rank < 4) put_rec(rec, rank, dist + ldist, corner + ld3(g + 4) - (0.5f * (dist + ldist)) * v3(0.f, 0.f, 1.f), v3(0.f, 0.f, 1.f));
                        } else n = plane_cylinder(g, P.margin, rec, tid == 0);
                    } else {
                        // geom 1 = the lower geom type (cylinder < box), then the lower geom id (ga); the stored normal runs from gb into ga
                        const bool a_first = !(g[0] == 0.f && h[0] != 0.f);
                        entB = D_NB + so.gobj[gb]; sgn = a_first ? -1.f : 1.f;
                        if (g[0] == 0.f && h[0] == 0.f) {
                            if (tid == 0) n = box_box(g, h, P.margin, rec, s.U);
                            n = __builtin_amdgcn_readfirstlane(n);
                        } else {
                            const GeomSupport s1(a_first ? g : h), s2(a_first ? h : g);
                            Contact c;
                            n = convex_pair(s1, s2, P.margin, c, mpr_scratch(s));
                            if (n && tid == 0) put_rec(rec, 0, c.dist, c.pos, c.n);
                        }
                    }
                    n = min(n, SL::MAXCON - ncon);
                    if (tid < n) put_contact<OBJ>(s, ncon + tid, ld3(rec + 7 * tid + 1), rec[7 * tid], sgn * ld3(rec + 7 * tid + 4), D_NB + ka, entB);
                    ncon += n;
                }
            }
            while (slot_done < D_MAXOBJ) { slot_done++; if (tid == 0) s.con_start[D_NB + slot_done] = ncon; }
        }
        if (tid == 0) { s.ncon = (SL::LEAN && over) ? SL::MAXCON + 1 : ncon; s.nlim = 0; }
    }
    KP_SYNC();
}

// efc_D and the reference acceleration of every constraint row.  aref (contact-frame 3-vector) goes to jv3.
template <int NT, bool OBJ, class SL>
__device__ __forceinline__ void make_constraint(SL& s, const DevTables& T, const Params& P, int tid) {
    const V3 o = ld3(s.xpos);
    for (int c = tid; c < s.ncon; c += NT) {
        int b = s.con_body[c];
        float r = s.con_D[c] - P.margin;                    // collide() left the distance here
        float imp = impedance(P, r);
        float iwA = T.body_invw[b < D_NB ? b : 0];
        if (OBJ && b >= D_NB) iwA = as_obj(s).oc[13 * (b - D_NB) + 10];
        float iwB = 0.f;
        if (OBJ) {                                         // invweight0 of the surface's entity: static geom / object slot / floor (0)
            const EnvLdsObj& so = as_obj(s);
            const int b2 = so.con_b2[c];
            if (b2 >= D_NB) iwB = so.oc[13 * (b2 - D_NB) + 10]; else if (b2 < -1) iwB = so.geom[17 * (-2 - b2) + 16];
        }
        float dA = (iwA + iwB) * (1.0f + P.mu * P.mu);
        float Rn = fmaxf(1e-15f, (1.0f - imp) * dA / imp);
        s.con_D[c] = 1.0f / (2.0f * P.mu * P.mu * Rn);
        S6 cv = lds6(s.sv + 6 * b);                         // sv still holds cvel from forward_kin_bias (objects: obj_forward)
        V3 vp = cv.l + cross(cv.a, ld3(s.con_pos + 3 * c) - o);
        if (OBJ) {
            const int b2 = as_obj(s).con_b2[c];
            if (b2 >= 0) { const S6 c2 = lds6(s.sv + 6 * b2); vp = vp - (c2.l + cross(c2.a, ld3(s.con_pos + 3 * c) - o)); }
        }
        V3 vf = frame_comp(contact_frame<OBJ>(s, c), vp);
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
        s.lim_jv[j] = aref; s.lim_D[j] = sgn * Dl;          // signed weight; aref sits in lim_jv until the first J search product
    }
    KP_SYNC();
}

// contact-frame residuals of all rows for the body spatial accelerations in acc (default: sv) and the generalized vector vec [- vec_b]:
// out3 = frame^T (point accel) [- aref] [+ jar3];  sub_aref: subtract the reference acceleration (jv3 / lim_jv);  add_base: add the
// residuals already in jar3 / lim_jar (rows are linear: residual(q + dq) = residual(q) + J dq)
template <int NT, bool OBJ, class SL>
__device__ __forceinline__ void eval_rows(SL& s, const float* vec, float* out3, float* lim_rows, bool sub_aref, int tid, const float* acc = nullptr, const float* vec_b = nullptr, bool add_base = false) {
    const V3 o = ld3(s.xpos);
    if (!acc) acc = s.sv;
    for (int c = tid; c < s.ncon; c += NT) {
        S6 S = lds6(acc + 6 * s.con_body[c]);
        V3 ap = S.l + cross(S.a, ld3(s.con_pos + 3 * c) - o);
        if (OBJ) {
            const int b2 = as_obj(s).con_b2[c];
            if (b2 >= 0) { const S6 S2 = lds6(acc + 6 * b2); ap = ap - (S2.l + cross(S2.a, ld3(s.con_pos + 3 * c) - o)); }
        }
        V3 a = frame_comp(contact_frame<OBJ>(s, c), ap);
        if (sub_aref) { a.x -= s.jv3[3 * c]; a.y -= s.jv3[3 * c + 1]; a.z -= s.jv3[3 * c + 2]; }
        if (add_base) { a.x += s.jar3[3 * c]; a.y += s.jar3[3 * c + 1]; a.z += s.jar3[3 * c + 2]; }
        out3[3 * c] = a.x; out3[3 * c + 1] = a.y; out3[3 * c + 2] = a.z;
    }
    for (int j = tid; j < D_NU; j += NT) {
        const float Ds = s.lim_D[j], sg = Ds > 0.f ? 1.f : (Ds < 0.f ? -1.f : 0.f);
        lim_rows[j] = sg != 0.f ? sg * (vec[6 + j] - (vec_b ? vec_b[6 + j] : 0.f)) - (sub_aref ? s.lim_jv[j] : 0.f) + (add_base ? s.lim_jar[j] : 0.f) : 0.f;
    }
    KP_SYNC();
}

// sa[b] = sum of the body wrenches sw over the subtree of b (bodies are in depth-first order: the subtree is [b, b + bsub[b])), summed in
// ascending body order.  The trip count is wave-uniform (the largest subtree among the wave's items) and the LDS reads go out in
// batches of 8 before the first add: a per-lane loop over its own subtree pays one LDS round trip per element (24 for the root).
// Reads past the subtree (at most 7 records, still inside EnvLds) are discarded by the select.
template <int NT, class SL>
__device__ __forceinline__ void subtree_sums(SL& s, int tid) {
    for (int base = 0; base < D_NB * 6; base += NT) {
        const int it = base + tid;
        const bool ok = it < D_NB * 6;
        const int b = ok ? it / 6 : 0, c = ok ? it - 6 * b : 0, n = ok ? (int)s.bsub[b] : 0;
        const float* src = s.sw + 6 * b + c;
        float acc = 0.f;
        for (int k0 = 0; __builtin_amdgcn_ballot_w64(k0 < n) != 0ull; k0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = src[6 * (k0 + j)];
#pragma unroll
            for (int j = 0; j < 8; j++) acc += (k0 + j < n) ? v[j] : 0.f;
        }
        if (ok) s.sa[it] = acc;
    }
    KP_SYNC();
}

// out = M (va - vb) (with_inertia; acc6 [24][6] must hold the body spatial accelerations of va - vb) - J^T f(jar) (with_forces)
template <int NT, bool OBJ, class SL>
// forces (optional): the contacts' world forces at the current residuals, [ncon][3], already evaluated (object kernels: con_prepare, lane = contact)
// bias / rhs (optional): body wrenches [24][6] added to, and a generalized force [75] subtracted from, the result: with the bias wrenches fb and
// rhs = qfrc_applied + qfrc_actuator, out = M va - qfrc_smooth - J^T f, the gradient of the primal problem without a detour through qacc_smooth
__device__ __forceinline__ void wrench_project(SL& s, const Params& P, const float* acc6, const float* va, const float* vb, float* out, bool with_inertia, bool with_forces, int tid,
                                               const float* arm, const float* forces = nullptr, const float* bias = nullptr, const float* rhs = nullptr) {
    if (tid < D_NB) {
        const int b = tid;
        S6 W = with_inertia ? inert_mul(s.cinert + 10 * b, lds6(acc6 + 6 * b)) : S6{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};
        if (bias) W = W + lds6(bias + 6 * b);
        if (with_forces) {
            const int c0 = s.con_start[b], c1 = s.con_start[b + 1];
            if (c1 > c0) {
                const V3 o = ld3(s.xpos);
                if (forces) {
                    V3 Fn = ld3(forces + 3 * c0), pn = ld3(s.con_pos + 3 * c0);
                    for (int c = c0; c < c1; c++) {
                        const V3 F = Fn, p = pn - o;
                        const int cn = min(c + 1, c1 - 1);
                        Fn = ld3(forces + 3 * cn); pn = ld3(s.con_pos + 3 * cn);
                        W.a = W.a - cross(p, F); W.l = W.l - F;
                    }
                } else {
                // the next contact's operands are requested before this one's arithmetic (one LDS round trip for the loop, not one per contact)
                float Dn = s.con_D[c0];
                V3 jn3 = ld3(s.jar3 + 3 * c0), pn = ld3(s.con_pos + 3 * c0);
                for (int c = c0; c < c1; c++) {
                    const float Dc = Dn, jn = jn3.x, jt1 = jn3.y, jt2 = jn3.z;
                    const V3 p = pn - o;
                    const int cn = min(c + 1, c1 - 1);
                    Dn = s.con_D[cn]; jn3 = ld3(s.jar3 + 3 * cn); pn = ld3(s.con_pos + 3 * cn);
                    float fe[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) { float x = row_val(e, P.mu, jn, jt1, jt2); fe[e] = x < 0.f ? -Dc * x : 0.f; }
                    // sum_e f_e (n +- mu t_k) in frame coordinates, then to world
                    const V3 F = frame_world(contact_frame<OBJ>(s, c), v3(fe[0] + fe[1] + fe[2] + fe[3], P.mu * (fe[0] - fe[1]), P.mu * (fe[2] - fe[3])));
                    W.a = W.a - cross(p, F); W.l = W.l - F;
                }
                }
            }
        }
        sts6(s.sw + 6 * b, W);
    }
    KP_SYNC();
    subtree_sums<NT>(s, tid);
    for (int d = tid; d < D_NV; d += NT) {
        float v = dot6(lds6(s.cdof + 6 * d), lds6(s.sa + 6 * s.dbody[d]));
        if (with_inertia) v += arm[d] * (va[d] - (vb ? vb[d] : 0.f));
        if (with_forces && d >= 6) { float jr = s.lim_jar[d - 6]; if (jr < 0.f) v += s.lim_D[d - 6] * jr; }      // - sign * (-D jar): lim_D is signed
        if (rhs) v -= rhs[d];
        out[d] = v;
    }
    KP_SYNC();
}

// 0.5 a^T I_b b summed over the bodies + 0.5 arm x y over the dofs: lane partial of  0.5 x^T M y  for the generalized vectors x, y
// whose body spatial accelerations are acca / accb (xa - xb and ya - yb give the dof vectors; xb / yb may be null)
template <int NT, class SL>
// arm: the dof armature (the layout's own copy, or the model table where the layout keeps none)
__device__ __forceinline__ float quad_form_M(const SL& s, const float* arm, const float* acca, const float* accb, const float* xa, const float* xb, const float* ya, const float* yb, int tid) {
    float c = 0.f;
    if (tid < D_NB) c += 0.5f * dot6(lds6(acca + 6 * tid), inert_mul(s.cinert + 10 * tid, lds6(accb + 6 * tid)));
    for (int i = tid; i < D_NV; i += NT) c += 0.5f * arm[i] * (xa[i] - (xb ? xb[i] : 0.f)) * (ya[i] - (yb ? yb[i] : 0.f));
    return c;
}

// spatial "acceleration" of every body induced by a generalized vector (what aba_solve leaves in sv).  Each body's own share
// da_b = sum_j vec_j cdof_j is formed body-parallel first, so the level-synchronous chain is one 6-vector add per level (as in the kinematics).
template <int NT, class SL>
__device__ __forceinline__ void spatial_accumulate(SL& s, const float* vec, int depth, int tid, float* out = nullptr) {
    if (!out) out = s.sv;
    S6 da = S6{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};
    if (tid < D_NB) {
        const int b = tid, nd = b == 0 ? 6 : 3, d0 = b == 0 ? 0 : 6 + 3 * (b - 1);
        for (int j = 0; j < nd; j++) da = da + vec[d0 + j] * lds6(s.cdof + 6 * (d0 + j));
        if (b == 0) sts6(out, da);
    }
    KP_SYNC();
#pragma nounroll
    for (int lev = 1; lev < D_NLEV; lev++) {
        if (depth == lev) sts6(out + 6 * tid, lds6(out + 6 * s.bpar[tid]) + da);
        KP_SYNC();
    }
}

// records the active pyramid rows of every contact; returns 1 on the lanes that saw a change since the last call.  deep: running
// maximum of the tree level of the hulls that carry an active row (first_clean_level's contact half, from the same row values)
template <int NT, class SL>
__device__ __forceinline__ float active_set_changed(SL& s, const Params& P, int tid, float& deep) {
    float changed = 0.f;
    for (int c = tid; c < s.ncon; c += NT) {
        const float jn = s.jar3[3 * c], jt1 = s.jar3[3 * c + 1], jt2 = s.jar3[3 * c + 2];
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) m |= (row_val(e, P.mu, jn, jt1, jt2) < 0.f ? 1u : 0u) << e;
        if (m != s.con_act[c]) changed = 1.f;
        s.con_act[c] = (unsigned char)m;
        const int b = s.con_body[c];
        if (m != 0u && b < D_NB) deep = fmaxf(deep, (float)s.bdep[b]);      // object-side contacts do not touch the humanoid tree
    }
    return changed;
}

// first tree level below every active constraint: 1 + the deepest level holding a body with an active contact row or a dof with
// an active joint limit (jar < 0); the root level always counts as dirty.  deep: per-lane maxima gathered by active_set_changed and
// the gradient loop
template <int NT, class SL>
__device__ __forceinline__ int first_clean_level(SL& s, float deep, int tid) {
    float m = -wave_min(-deep);
    if (NT > 64) {
        KP_SYNC();
        if ((tid & 63) == 0) s.red[tid >> 6] = m;
        KP_SYNC();
        m = s.red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; w++) m = fmaxf(m, s.red[w]);
        KP_SYNC();
    }
    return (int)m + 1;
}

// exact minimiser of the cost along the search direction: root of phi'(alpha) = g0 + alpha h0 + sum_rows D (jar + alpha jv)_- jv by
// safeguarded Newton steps on the piecewise-linear phi'.  Every lane keeps its rows in registers for the whole search -- one contact
// (four pyramid rows a + alpha b with a = row(jar), b = row(jv)) and ceil(69 / NT) joint-limit rows -- so an evaluation is a few
// FMAs per row and two wave sums, without LDS traffic.  rowcost: the rows' share of the cost at the returned alpha.
// ROWCOST = false leaves rowcost alone (rounds 1 - 2: the object kernel kept its full cost evaluation because three more live values across
// the search moved spills into its articulated-body loops; since the round-3 register discipline both solvers take the closed form)
template <int NT, bool ROWCOST = true, class SL>
// want_rc0: rc0 receives the rows' share of the cost at alpha = 0, i.e. at the iterate the search starts from (the solve's first iteration has no
// previous line search to take it from)
__device__ __forceinline__ float line_search(SL& s, const Params& P, float g0, float h0, int tid, float& rowcost, bool want_rc0, float& rc0) {
    static_assert(SL::MAXCON <= 64 && NT >= 64, "one contact per lane");
    float ra[4], rb[4], rD[4], Dc;
    {
        const bool ok = tid < s.ncon;
        const int k = ok ? tid : 0;
        Dc = ok ? s.con_D[k] : 0.f;
        const float jn = ok ? s.jar3[3 * k] : 0.f, jt1 = ok ? s.jar3[3 * k + 1] : 0.f, jt2 = ok ? s.jar3[3 * k + 2] : 0.f;
        const float vn = ok ? s.jv3[3 * k] : 0.f, vt1 = ok ? s.jv3[3 * k + 1] : 0.f, vt2 = ok ? s.jv3[3 * k + 2] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) { ra[e] = row_val(e, P.mu, jn, jt1, jt2); rb[e] = row_val(e, P.mu, vn, vt1, vt2); rD[e] = Dc * rb[e]; }
    }
    constexpr int LR = (D_NU + NT - 1) / NT;
    float la[LR], lb[LR], lD[LR], lW[LR];
#pragma unroll
    for (int n = 0; n < LR; n++) {
        const int j = tid + n * NT, jj = j < D_NU ? j : 0;
        const bool ok = j < D_NU && s.lim_D[jj] != 0.f;
        const float w = ok ? fabsf(s.lim_D[jj]) : 0.f;
        if (ROWCOST) lW[n] = w;
        la[n] = ok ? s.lim_jar[jj] : 0.f; lb[n] = ok ? s.lim_jv[jj] : 0.f; lD[n] = w * lb[n];
    }
    if (ROWCOST && want_rc0) {
        float rc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) if (ra[e] < 0.f) rc += 0.5f * Dc * ra[e] * ra[e];
#pragma unroll
        for (int n = 0; n < LR; n++) if (la[n] < 0.f) rc += 0.5f * lW[n] * la[n] * la[n];
        rc0 = block_sum<NT>(s, rc, tid);
    }
    // The search direction solves H search = -grad with the Hessian of the current active set, so phi'(0) = -phi''(0) and the Newton
    // step from alpha = 0 is 1: the first evaluation happens there, bracketed by lo = 0 (descent direction).
    float alpha = 1.f, lo = 0.f, hi = 3.0e38f;
    for (int ls = 0; ls < 20; ls++) {
        float d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) { const float x = fmaf(alpha, rb[e], ra[e]); if (x < 0.f) { d1 += x * rD[e]; d2 += rb[e] * rD[e]; } }
#pragma unroll
        for (int n = 0; n < LR; n++) { const float x = fmaf(alpha, lb[n], la[n]); if (x < 0.f) { d1 += x * lD[n]; d2 += lb[n] * lD[n]; } }
        d1 = block_sum<NT>(s, d1, tid); d2 = block_sum<NT>(s, d2, tid);
        const float dphi = g0 + alpha * h0 + d1, ddphi = h0 + d2;
        if (!(ddphi > 0.f)) { if (ls == 0) alpha = 0.f; break; }
        if (dphi < 0.f) lo = alpha; else hi = alpha;
        float an = alpha - dphi * rcp_nr(ddphi);
        if (!(an > lo && an < hi)) an = hi < 1.0e38f ? 0.5f * (lo + hi) : 2.0f * alpha + 1.0f;
        const float step = an - alpha;
        alpha = an;
        if (fabsf(step) <= 1e-6f * fabsf(alpha)) break;
    }
    if (ROWCOST) {   // constraint part of the cost at the step taken: sum over the rows of 0.5 D (a + alpha b)_-^2 (the caller adds the Gauss term in closed form)
        float rc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) { const float x = fmaf(alpha, rb[e], ra[e]); if (x < 0.f) rc += 0.5f * Dc * x * x; }
#pragma unroll
        for (int n = 0; n < LR; n++) { const float x = fmaf(alpha, lb[n], la[n]); if (x < 0.f) rc += 0.5f * lW[n] * x * x; }
        rowcost = block_sum<NT>(s, rc, tid);
    }
    return alpha;
}

// Constraint solve: Newton on the primal problem (mj_solNewton) with an exact line search, WITHOUT qacc_smooth.  Returns iterations.
// mj_fwdConstraint evaluates two starting points -- qacc_smooth and the warm start -- and iterates from the cheaper one (rounds 1 - 2 did the
// same); the primal problem is strictly convex, so its minimiser does not depend on the starting point, and its cost
//     0.5 (a - a_s)^T M (a - a_s) + s(J a - aref)  =  0.5 a^T M a - a^T qfrc_smooth + s(J a - aref) + const
// needs a_s = M^-1 qfrc_smooth neither for the gradient (M a - qfrc_smooth - J^T f, in body form: I_b sacc_b + fb_b - contact wrenches, projected)
// nor for the termination tests (|gradient| and the cost DIFFERENCE of consecutive iterates, exact in closed form along the search direction).
// So the solve starts from the warm start always, and the substep's "smooth" articulated-body factorisation + solve (a fifth of a substep) is not
// run at all: the FIRST Newton factorisation walks every tree level and so leaves M's own factors on the clean levels, which the later
// factorisations of the substep reuse exactly as they reused the smooth solve's.  Without constraint rows qacc = M^-1 qfrc_smooth is one plain solve.
// Results agree with the two-candidate form to the solver's tolerance; the iteration path is MuJoCo's whenever MuJoCo starts from its warm start.
// On entry: s.qacc = warm start, jv3 / lim_jv = aref, s.applied ++ s.ctrl = qfrc_applied + qfrc_actuator, s.fb = bias wrenches.
template <int NT, class SL>
__device__ __forceinline__ int solve_constraints_direct(SL& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap, const float* arm) {
    if (s.ncon == 0 && s.nlim == 0) {
        aba_solve<NT, false>(s, P, L8, s.applied, s.qacc, false, tid, D_NLEV, s.fb);
        return 0;
    }
    float* sacc;               // [24][6] spatial accelerations of the bodies induced by the iterate qacc: over Mv + mres; lean layout: over qpos | qvel (restored by step_body)
    if constexpr (SL::LEAN) sacc = s.qpos; else sacc = s.Mv;
    float* grad = s.qacc_s;    // the words qacc_smooth would occupy
    spatial_accumulate<NT>(s, s.qacc, depth, tid, sacc);
    eval_rows<NT, false>(s, s.qacc, s.jar3, s.lim_jar, true, tid, sacc);
    float rowcost = 0.f;       // the rows' share of the cost at the iterate (the first line search evaluates it at alpha = 0)
    int it = 0, lev_hist = 1;
    bool done = false;          // left the loop through one of mj_solNewton's termination tests (not the iteration cap)
    const unsigned conlev = contact_levels(s, L8);
    // active set of the iterate: pyramid rows (con_act) and joint limits (their D as extra armature); changed = it differs from the set of the last call
    float changed = 0.f, deep = 0.f;
    auto active_set = [&]() {
        changed = 0.f; deep = 0.f;
        for (int i = tid; i < D_NV; i += NT) {
            const float ex = (i >= 6 && s.lim_jar[i - 6] < 0.f) ? fabsf(s.lim_D[i - 6]) : 0.f;
            const float st = SL::LEAN ? arm[i] + ex : ex;          // lean layout: extra holds armature + extra, the sum aba_elim3 forms on the full layout
            if (st != s.extra[i]) changed = 1.f;
            s.extra[i] = st;
            if (ex != 0.f) deep = fmaxf(deep, (float)s.bdep[s.dbody[i]]);     // active joint limit: its body's level is dirty
        }
        changed += active_set_changed<NT>(s, P, tid, deep);
        changed = (NT == 64) ? (__ballot(changed > 0.f) != 0ull ? 1.f : 0.f) : block_sum<NT>(s, changed, tid);
        KP_SYNC();
    };
    active_set();
    for (; it < P.max_iter; it++) {
        wrench_project<NT, false>(s, P, sacc, s.qacc, nullptr, grad, true, true, tid, arm, nullptr, s.fb, s.applied);
        float g2 = 0.f;
        for (int i = tid; i < D_NV; i += NT) { const float g = grad[i]; g2 += g * g; s.x[i] = -g; }
        g2 = block_sum<NT>(s, g2, tid);
        KP_SYNC();
        if (P.scale * sqrtf(g2) < P.tol) { done = true; break; }
        if (it == 0 || changed > 0.f) {
            const int clean = first_clean_level<NT>(s, deep, tid);
            lev_hist = max(lev_hist, clean);
            // first factorisation of the substep: every level (its clean levels ARE M's factors from then on)
            aba_solve<NT, false>(s, P, L8, s.x, s.search, true, tid, it == 0 ? D_NLEV : lev_hist, nullptr, conlev);
            nfact++;
        }
        else aba_resolve(s, L8, s.x, nullptr, s.search);
        eval_rows<NT, false>(s, s.search, s.jv3, s.lim_jv, false, tid);           // aref (in jv3) is folded into jar3 by now
        // phi(alpha) = cost(qacc + alpha search): smooth part g0 = search^T (M qacc - qfrc_smooth), h0 = search^T M search, in body form
        float g0 = 2.0f * quad_form_M<NT>(s, arm, s.sv, sacc, s.search, nullptr, s.qacc, nullptr, tid);
        float h0 = 2.0f * quad_form_M<NT>(s, arm, s.sv, s.sv, s.search, nullptr, s.search, nullptr, tid);
        for (int i = tid; i < D_NB * 6; i += NT) g0 += s.sv[i] * s.fb[i];
        for (int i = tid; i < D_NV; i += NT) g0 -= s.search[i] * s.applied[i];
        g0 = block_sum<NT>(s, g0, tid); h0 = block_sum<NT>(s, h0, tid);
        float rownew, rc0 = 0.f;
        const float alpha = line_search<NT>(s, P, g0, h0, tid, rownew, it == 0, rc0);
        if (it == 0) rowcost = rc0;
        if (!(alpha > 0.f)) { done = true; break; }
        for (int i = tid; i < D_NV; i += NT) s.qacc[i] += alpha * s.search[i];
        for (int i = tid; i < D_NB * 6; i += NT) sacc[i] += alpha * s.sv[i];
        for (int k = tid; k < 3 * s.ncon; k += NT) s.jar3[k] += alpha * s.jv3[k];
        for (int j = tid; j < D_NU; j += NT) if (s.lim_D[j] != 0.f) s.lim_jar[j] += alpha * s.lim_jv[j];
        KP_SYNC();
        // cost(old) - cost(new): the smooth part is an exact quadratic along the search direction, the rows' share comes out of the line search
        const float improvement = P.scale * ((rowcost - rownew) - alpha * (g0 + 0.5f * alpha * h0));
        rowcost = rownew;
        if (improvement < P.tol) { it++; done = true; break; }
        // The Newton step of a model that was exact: the search direction solved H search = -gradient for the active set of the old iterate and the new
        // iterate has the same active set -- on a piecewise-quadratic cost its gradient is then (1 - alpha) x the old one, alpha = 1 up to the line
        // search's rounding.  When that bound already passes mj_solNewton's |gradient| < tolerance test, the pass that would evaluate the gradient at
        // the top of the next iteration (body wrenches, subtree sums, projection: a sixth of an iteration's work) is not run to confirm it.
        active_set();
        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }     // gradient(new) = (1 - alpha) gradient(old) on an unchanged active set
    }
    if (!done) ncap++;
    return it;
}

// ---------------------------------------------------------------- dynamic free objects (OBJ kernels; one wave per env)
// An object is one more rigid body whose unknown in the Newton solve is its spatial acceleration about o (a linear
// re-parametrisation of its six free-joint accelerations, under which Newton steps and the exact line search are invariant).
// Oracle counterpart: kpo_obj_forward + the object columns of kpo_make_constraint / kpo_solve_constraint.

// 6x6 entry of a 10-float spatial inertia [Ixx Iyy Izz Ixy Ixz Iyz | h | m]
__device__ __forceinline__ float inert_entry(const float* I, int r, int c) {
    if (r > c) { const int t = r; r = c; c = t; }
    if (c < 3) return r == c ? I[r] : I[2 + r + c];
    if (r >= 3) return r == c ? I[9] : 0.f;
    const int l = c - 3;
    if (r == l) return 0.f;
    return ((l == (r + 2) % 3) ? 1.f : -1.f) * I[6 + (3 - r - l)];
}
__device__ __forceinline__ S6 unit6(int j) { return S6{v3(j == 0, j == 1, j == 2), v3(j == 3, j == 4, j == 5)}; }

// D F G F^T a: the active pyramid rows of contact c applied to a point acceleration (world components)
__device__ __forceinline__ V3 con_DG(const EnvLdsObj& s, const Params& P, int c, V3 a) {
    const float mu = P.mu, mu2 = mu * mu, Dc = s.con_D[c], jn = s.jar3[3 * c], jt1 = s.jar3[3 * c + 1], jt2 = s.jar3[3 * c + 2];
    const float a0 = row_val(0, mu, jn, jt1, jt2) < 0.f, a1 = row_val(1, mu, jn, jt1, jt2) < 0.f;
    const float a2 = row_val(2, mu, jn, jt1, jt2) < 0.f, a3 = row_val(3, mu, jn, jt1, jt2) < 0.f;
    const float gnn = a0 + a1 + a2 + a3, g11 = mu2 * (a0 + a1), g22 = mu2 * (a2 + a3), gn1 = mu * (a0 - a1), gn2 = mu * (a2 - a3);
    const Frame fr = contact_frame<true>(s, c);
    const V3 pf = frame_comp(fr, a);
    return frame_world(fr, v3(Dc * (gnn * pf.x + gn1 * pf.y + gn2 * pf.z), Dc * (gn1 * pf.x + g11 * pf.y), Dc * (gn2 * pf.x + g22 * pf.z)));
}
// K_c acc: the wrench (about o) contact c's active rows put on a body accelerating with acc
__device__ __forceinline__ S6 con_K(const EnvLdsObj& s, const Params& P, int c, S6 acc, V3 o) {
    const V3 p = ld3(s.con_pos + 3 * c) - o;
    const V3 w = con_DG(s, P, c, acc.l + cross(acc.a, p));
    return S6{cross(p, w), w};
}
// contact force (world) of contact c at the current residuals, acting on the entity that carries the vertex
__device__ __forceinline__ V3 con_force(const EnvLdsObj& s, const Params& P, int c) {
    const float Dc = s.con_D[c], jn = s.jar3[3 * c], jt1 = s.jar3[3 * c + 1], jt2 = s.jar3[3 * c + 2];
    float fe[4];
#pragma unroll
    for (int e = 0; e < 4; e++) { const float x = row_val(e, P.mu, jn, jt1, jt2); fe[e] = x < 0.f ? -Dc * x : 0.f; }
    return frame_world(contact_frame<true>(s, c), v3(fe[0] + fe[1] + fe[2] + fe[3], P.mu * (fe[0] - fe[1]), P.mu * (fe[2] - fe[3])));
}

// pose-dependent quantities of the objects (lane = slot), their world geoms (lane = geom); spatial velocity -> sv[24 + k]
__device__ __forceinline__ void obj_forward(EnvLdsObj& s, const DevTables& T, const Params& P, int tid) {
    const V3 o = ld3(s.xpos);
    if (tid < s.nobj) {
        const int k = tid;
        float* q = s.oq + 7 * k;
        const Q4 qq = qnormalize(Q4{q[3], q[4], q[5], q[6]});
        q[3] = qq.w; q[4] = qq.x; q[5] = qq.y; q[6] = qq.z;
        float R[9];
        q2mat(qq, R);
#pragma unroll
        for (int i = 0; i < 9; i++) s.oR[9 * k + i] = R[i];
        const float* C = s.oc + 13 * k;
        const float mass = C[0], arm = C[12];
        const V3 xp = ld3(q), com = xp + mulmat(R, ld3(C + 1));
        const float I3[9] = {C[4], C[7], C[8], C[7], C[5], C[9], C[8], C[9], C[6]};
        float Tm[9], W[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Tm[3 * i + j] = R[3 * i] * I3[j] + R[3 * i + 1] * I3[3 + j] + R[3 * i + 2] * I3[6 + j];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) W[3 * i + j] = Tm[3 * i] * R[3 * j] + Tm[3 * i + 1] * R[3 * j + 1] + Tm[3 * i + 2] * R[3 * j + 2];
        const V3 rr = com - o, ra = xp - o;
        const float r2 = dot(rr, rr), a2 = dot(ra, ra);
        float* ci = s.oI + 10 * k;
        ci[0] = W[0] + mass * (r2 - rr.x * rr.x); ci[1] = W[4] + mass * (r2 - rr.y * rr.y); ci[2] = W[8] + mass * (r2 - rr.z * rr.z);
        ci[3] = W[1] - mass * rr.x * rr.y; ci[4] = W[2] - mass * rr.x * rr.z; ci[5] = W[5] - mass * rr.y * rr.z;
        ci[6] = mass * rr.x; ci[7] = mass * rr.y; ci[8] = mass * rr.z; ci[9] = mass;
        // the free joint's armature (diagonal in joint coordinates) = a point mass at the body origin + an isotropic rotor
        float* ce = s.oIe + 10 * k;
        ce[0] = ci[0] + arm * (a2 - ra.x * ra.x) + arm; ce[1] = ci[1] + arm * (a2 - ra.y * ra.y) + arm; ce[2] = ci[2] + arm * (a2 - ra.z * ra.z) + arm;
        ce[3] = ci[3] - arm * ra.x * ra.y; ce[4] = ci[4] - arm * ra.x * ra.z; ce[5] = ci[5] - arm * ra.y * ra.z;
        ce[6] = ci[6] + arm * ra.x; ce[7] = ci[7] + arm * ra.y; ce[8] = ci[8] + arm * ra.z; ce[9] = ci[9] + arm;
        const V3 vl = ld3(s.ov + 6 * k), ww = mulmat(R, ld3(s.ov + 6 * k + 3));
        const S6 cv = S6{ww, vl - cross(ww, ra)};
        const S6 ca = S6{v3(0.f, 0.f, 0.f), v3(-P.gx, -P.gy, -P.gz) + cross(vl, ww)};
        sts6(s.ofb + 6 * k, inert_mul(ci, ca) + cross_force(cv, inert_mul(ci, cv)));
        sts6(s.sv + 6 * (D_NB + k), cv);
        // warm start: joint-space acceleration of the previous solve -> spatial acceleration in this pose
        const V3 al_q = ld3(s.oqa + 6 * k), aa = mulmat(R, ld3(s.oqa + 6 * k + 3));
        sts6(s.oa + 6 * k, S6{aa, al_q - cross(aa, ra)});
    }
    KP_SYNC();
    if (tid >= s.ngeom_static && tid < s.ngeom) {
        const int k = s.gobj[tid];
        const float* l = T.obj_geoms + 18 * s.ggi[tid] + 1;      // body-frame geom: type, size[3], pos[3], mat[9] (L2-resident table)
        const float* R = s.oR + 9 * k;
        float* g = s.geom + 17 * tid;
        g[0] = l[0]; g[1] = l[1]; g[2] = l[2]; g[3] = l[3];
        st3(g + 4, ld3(s.oq + 7 * k) + mulmat(R, ld3(l + 4)));
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) g[7 + 3 * i + j] = R[3 * i] * l[7 + j] + R[3 * i + 1] * l[10 + j] + R[3 * i + 2] * l[13 + j];
        g[16] = s.oc[13 * k + 10];
    }
    KP_SYNC();
}

// The <= 12 x 12 SPD object system, one row per lane, held in registers (row stride 13 in s.Sm, right-hand side in column n).
// Gaussian elimination with the pivot row broadcast by v_readlane; rows >= n are identity padding.  refactor = false reuses the
// multipliers / upper triangle a previous call stored back to s.Sm and only pushes a new right-hand side through them.
// NN = 6 or 12: compiled for one and for two simulated objects (the elimination is unrolled over NN columns; most scenes hold one object)
template <int NN>
__device__ __forceinline__ void dense_solve_n(EnvLdsObj& s, int n, int tid, bool refactor) {
    constexpr int ST = 6 * D_MAXOBJ + 1;
    float a[NN + 1];
    const bool rowok = tid < n;
#pragma unroll
    for (int c = 0; c < NN; c++) a[c] = (rowok && c < n) ? s.Sm[tid * ST + c] : (c == tid ? 1.f : 0.f);
    a[NN] = rowok ? s.Sm[tid * ST + n] : 0.f;
#pragma unroll
    for (int j = 0; j < NN; j++) {
        const float pinv = 1.0f / __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[j]), j));
        const bool lower = tid > j;
        const float l = refactor ? (lower ? a[j] * pinv : 0.f) : (lower ? a[j] : 0.f);
        if (refactor) {
#pragma unroll
            for (int c = j + 1; c < NN; c++) a[c] -= l * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[c]), j));
            if (lower) a[j] = l;
        }
        a[NN] -= l * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[NN]), j));
    }
#pragma unroll
    for (int j = NN - 1; j >= 0; j--) {
        const float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[NN] / a[j]), j));
        if (tid == j) a[NN] = xj;
        if (tid < j) a[NN] -= a[j] * xj;
    }
    if (rowok) {
        if (refactor) {
#pragma unroll
            for (int c = 0; c < NN; c++) if (c < n) s.Sm[tid * ST + c] = a[c];
        }
        s.Sm[tid * ST + n] = a[NN];
    }
    KP_SYNC();
}
__device__ __forceinline__ void dense_solve(EnvLdsObj& s, int n, int tid, bool refactor) {
    if (n <= 6) dense_solve_n<6>(s, n, tid, refactor); else dense_solve_n<6 * D_MAXOBJ>(s, n, tid, refactor);
}

// per contact (lane = contact): the active-row matrix D F G F^T and the contact force at the current residuals
template <int NT>
__device__ __forceinline__ void con_prepare(EnvLdsObj& s, const Params& P, int tid) {
    for (int c = tid; c < s.ncon; c += NT) {
        const V3 mx = con_DG(s, P, c, v3(1.f, 0.f, 0.f)), my = con_DG(s, P, c, v3(0.f, 1.f, 0.f)), mz = con_DG(s, P, c, v3(0.f, 0.f, 1.f));
        float* m = s.cM + 6 * c;
        m[0] = mx.x; m[1] = my.y; m[2] = mz.z; m[3] = mx.y; m[4] = mx.z; m[5] = my.z;
        st3(s.jv3 + 3 * c, con_force(s, P, c));    // contact forces live in jv3 (the previous search direction's rows: dead since the update) until this iteration's eval_rows
    }
    KP_SYNC();
}
// K_c acc from the prepared matrix
__device__ __forceinline__ S6 con_Kp(const EnvLdsObj& s, int c, S6 acc, V3 o) {
    const V3 p = ld3(s.con_pos + 3 * c) - o;
    const V3 a = acc.l + cross(acc.a, p);
    const float* m = s.cM + 6 * c;
    const V3 w = v3(m[0] * a.x + m[3] * a.y + m[4] * a.z, m[3] * a.x + m[1] * a.y + m[5] * a.z, m[4] * a.x + m[5] * a.y + m[2] * a.z);
    return S6{cross(p, w), w};
}
__device__ __forceinline__ S6 wave_sum6(S6 v) {
    return S6{v3(wave_sum(v.a.x), wave_sum(v.a.y), wave_sum(v.a.z)), v3(wave_sum(v.l.x), wave_sum(v.l.y), wave_sum(v.l.z))};
}

// sum of V per-lane values over the 64 lanes (lane = contact): four DPP exchanges inside each 16-lane row, the four row sums meet
// in LDS.  Deterministic order.  Lane v < V returns sum v.
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}
template <int V>
__device__ __forceinline__ float lanes_sum(EnvLdsObj& s, const float* vals, int tid) {
    static_assert(V <= 21, "red[] holds 4 x 21 partials");
#pragma unroll
    for (int v = 0; v < V; v++) {
        const float r = row_sum16(vals[v]);
        if ((tid & 15) == 0) s.red[(tid >> 4) * 21 + v] = r;
    }
    KP_SYNC();
    const float out = tid < V ? (s.red[tid] + s.red[21 + tid]) + (s.red[42 + tid] + s.red[63 + tid]) : 0.f;
    KP_SYNC();
    return out;                                                     // lane v < V holds sum v
}
// symmetric 6 x 6 index (r <= c) -> 0..20, row-major upper triangle
__device__ __forceinline__ constexpr int sym6(int r, int c) { return r <= c ? (r * (13 - r)) / 2 + (c - r) : (c * (13 - c)) / 2 + (r - c); }

// object rows of the Newton system: Sm = blockdiag(I_eff) + sum of the active contact matrices K_c = P M_c P^T over the contacts that
// touch an object (own block +K_c, object-object block -K_c); column n = sign * rhs.  Lane = contact: every lane forms the 21 unique
// entries of its K_c = [[X M X^T, X M], [M X^T, M]] (X = [p]x), then one wave-wide vector sum per block.
__device__ __forceinline__ void obj_hessian(EnvLdsObj& s, const Params& P, const float* rhs, float sign, int tid) {
    constexpr int ST = 6 * D_MAXOBJ + 1;
    const int nobj = s.nobj, n = 6 * nobj;
    const V3 o = ld3(s.xpos);
    const bool have = tid < s.ncon;
    const int c = have ? tid : 0;
    const int A = have ? s.con_body[c] : -1, B = have ? (int)s.con_b2[c] : -1;
    float K[21];
    {
        const V3 p = ld3(s.con_pos + 3 * c) - o;
        const float* m = s.cM + 6 * c;
        const V3 m0 = v3(m[0], m[3], m[4]), m1 = v3(m[3], m[1], m[5]), m2 = v3(m[4], m[5], m[2]);     // columns (= rows) of M
        const V3 b0 = cross(p, m0), b1 = cross(p, m1), b2 = cross(p, m2);                               // B = X M, column j = p x M[:, j]
        // rows of B: B_i = (b0[i], b1[i], b2[i]); A = X M X^T, row i = p x B_i
        const V3 a0 = cross(p, v3(b0.x, b1.x, b2.x)), a1 = cross(p, v3(b0.y, b1.y, b2.y)), a2 = cross(p, v3(b0.z, b1.z, b2.z));
        K[sym6(0, 0)] = a0.x; K[sym6(0, 1)] = a0.y; K[sym6(0, 2)] = a0.z; K[sym6(1, 1)] = a1.y; K[sym6(1, 2)] = a1.z; K[sym6(2, 2)] = a2.z;
        K[sym6(0, 3)] = b0.x; K[sym6(0, 4)] = b1.x; K[sym6(0, 5)] = b2.x;
        K[sym6(1, 3)] = b0.y; K[sym6(1, 4)] = b1.y; K[sym6(1, 5)] = b2.y;
        K[sym6(2, 3)] = b0.z; K[sym6(2, 4)] = b1.z; K[sym6(2, 5)] = b2.z;
        K[sym6(3, 3)] = m[0]; K[sym6(3, 4)] = m[3]; K[sym6(3, 5)] = m[4]; K[sym6(4, 4)] = m[1]; K[sym6(4, 5)] = m[5]; K[sym6(5, 5)] = m[2];
    }
    // lane v < 21 owns unique entry v: its (r, cc)
    int er = 0, ec = tid;
#pragma unroll
    for (int r = 0; r < 5; r++) if (ec >= 6 - er && er == r) { ec -= 6 - er; er++; }
    ec += er;
    for (int blk = 0; blk < (nobj == 2 ? 3 : 1); blk++) {           // blocks (0,0), (1,1), (0,1)
        const int k = blk == 1 ? 1 : 0;
        float w;
        if (blk < 2) w = (have && (A == D_NB + k || B == D_NB + k)) ? 1.f : 0.f;
        else w = (have && A >= D_NB && B >= D_NB) ? -1.f : 0.f;
        float vals[21];
#pragma unroll
        for (int v = 0; v < 21; v++) vals[v] = w * K[v];
        const float sum = lanes_sum<21>(s, vals, tid);
        if (tid < 21) {
            if (blk < 2) {
                const float v = sum + inert_entry(s.oIe + 10 * k, er, ec);
                s.Sm[(6 * k + er) * ST + 6 * k + ec] = v; s.Sm[(6 * k + ec) * ST + 6 * k + er] = v;
            } else {                                                 // K symmetric: block (0,1) = -K, block (1,0) = -K^T = -K
                s.Sm[er * ST + 6 + ec] = sum; s.Sm[ec * ST + 6 + er] = sum; s.Sm[(6 + ec) * ST + er] = sum; s.Sm[(6 + er) * ST + ec] = sum;
            }
        }
    }
    if (tid < n) s.Sm[tid * ST + n] = sign * rhs[tid];
    KP_SYNC();
}

// out[6k..] = sum over the hull contacts whose surface belongs to object k of K_c sv[hull]   (= -H_oh y for the y behind sv)
__device__ __forceinline__ void obj_coupling_u(EnvLdsObj& s, const Params& P, float* out, int tid) {
    const V3 o = ld3(s.xpos);
    const bool have = tid < s.con_start[D_NB];
    const int c = have ? tid : 0;
    const int B = have ? (int)s.con_b2[c] : -1;
    const S6 w = con_Kp(s, c, lds6(s.sv + 6 * s.con_body[c]), o);
    float vals[12];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float f = B == D_NB + k ? 1.f : 0.f;
        vals[6 * k] = f * w.a.x; vals[6 * k + 1] = f * w.a.y; vals[6 * k + 2] = f * w.a.z; vals[6 * k + 3] = f * w.l.x; vals[6 * k + 4] = f * w.l.y; vals[6 * k + 5] = f * w.l.z;
    }
    const float sum = lanes_sum<12>(s, vals, tid);
    if (tid < 6 * s.nobj) out[tid] = sum;
    KP_SYNC();
}
// sw[b] = sign * sum over hull b's contacts with an object surface (slot ksel, or any if ksel < 0) of K_c acc[slot]
__device__ __forceinline__ void hull_coupling_wrench(EnvLdsObj& s, const Params& P, int ksel, const float* acc, float sign, int tid) {
    if (tid < D_NB) {
        const V3 o = ld3(s.xpos);
        S6 W = S6{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};
        for (int c = s.con_start[tid]; c < s.con_start[tid + 1]; c++) {
            const int B = s.con_b2[c];
            if (B >= D_NB && (ksel < 0 || B == D_NB + ksel)) W = W + con_Kp(s, c, lds6(acc + 6 * (B - D_NB)), o);
        }
        sts6(s.sw + 6 * tid, sign * W);
    }
    KP_SYNC();
}
// object part of the gradient: g_k = mres_k - sum_{vertex side = k} [p x F; F] + sum_{surface side = k} [p x F; F]
__device__ __forceinline__ void obj_gradient(EnvLdsObj& s, int tid) {
    const V3 o = ld3(s.xpos);
    const bool have = tid < s.ncon;
    const int c = have ? tid : 0;
    const int A = have ? s.con_body[c] : -1, B = have ? (int)s.con_b2[c] : -1;
    const V3 F = ld3(s.jv3 + 3 * c), p = ld3(s.con_pos + 3 * c) - o;
    const V3 t = cross(p, F);
    float vals[12];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float f = A == D_NB + k ? -1.f : (B == D_NB + k ? 1.f : 0.f);
        vals[6 * k] = f * t.x; vals[6 * k + 1] = f * t.y; vals[6 * k + 2] = f * t.z; vals[6 * k + 3] = f * F.x; vals[6 * k + 4] = f * F.y; vals[6 * k + 5] = f * F.z;
    }
    const float sum = lanes_sum<12>(s, vals, tid);
    if (tid < 6 * s.nobj) s.ogr[tid] = s.omres[tid] + sum;
    KP_SYNC();
}
// spatial -> joint-space acceleration of the objects, then the same semi-implicit Euler step as the humanoid root
__device__ __forceinline__ void obj_integrate(EnvLdsObj& s, const Params& P, int tid) {
    if (tid < s.nobj) {
        const int k = tid;
        const float* R = s.oR + 9 * k;
        const S6 a = lds6(s.oa + 6 * k);
        const V3 ra = ld3(s.oq + 7 * k) - ld3(s.xpos);
        const V3 ql = a.l + cross(a.a, ra);
        const V3 qa = v3(R[0] * a.a.x + R[3] * a.a.y + R[6] * a.a.z, R[1] * a.a.x + R[4] * a.a.y + R[7] * a.a.z, R[2] * a.a.x + R[5] * a.a.y + R[8] * a.a.z);
        sts6(s.oqa + 6 * k, S6{ql, qa});
        float* v = s.ov + 6 * k;
        float* q = s.oq + 7 * k;
        v[0] += P.h * ql.x; v[1] += P.h * ql.y; v[2] += P.h * ql.z; v[3] += P.h * qa.x; v[4] += P.h * qa.y; v[5] += P.h * qa.z;
        q[0] += P.h * v[0]; q[1] += P.h * v[1]; q[2] += P.h * v[2];
        const V3 w = ld3(v + 3);
        const float n = sqrtf(dot(w, w));
        Q4 qr = Q4{1.f, 0.f, 0.f, 0.f};
        if (n >= 1e-15f) { float sn, cs; sincosf(0.5f * P.h * n, &sn, &cs); const float kk = sn / n; qr = Q4{cs, w.x * kk, w.y * kk, w.z * kk}; }
        const Q4 qn = qmul(qnormalize(Q4{q[3], q[4], q[5], q[6]}), qr);
        q[3] = qn.w; q[4] = qn.x; q[5] = qn.y; q[6] = qn.z;
    }
}

// Schur-complement columns of the object block, all at once.  Column j of  S = H_oo - H_oh H_hh^-1 H_ho  needs z_j = H_hh^-1 (H_ho e_j):
// a bias-force-only pass through the articulated-body factorisation whose right-hand side is a set of body wrenches on the hulls that
// touch object k(j) (-K_c e_j), and of whose result only the spatial accelerations of the touching hulls are read (H_oh z_j = sum K_c a).
// So only the union of the touching hulls' paths to the root takes part, and the columns are independent: lane = (column, row of the
// 6-vector), 8 columns x 8 rows per round, the bodies of the path visited one after the other (leaves -> root in descending body order,
// root -> leaves ascending; bodies are in depth-first order, so a per-depth accumulator is all the hand-up needs).  Replaces 6 n_obj
// passes of aba_resolve + hull_coupling_wrench + obj_coupling_u (12 passes per factorisation in the push scene) by one or two rounds.
// Scratch: the 552 contiguous floats jv3 | lim_jv | sa | sw (dead between the gradient and the row evaluation): per-depth accumulators
// [9][nc][6] + the joint-space right-hand sides u of the path's dofs [npd][nc]; nc = columns per round is what fits.
constexpr int D_SCHUR_SCRATCH = D_MAXCON * 3 + 72 + 144 + 144;
__device__ __forceinline__ void schur_columns(EnvLdsObj& s, const Params& P, unsigned cmask, int tid) {
    static_assert(offsetof(EnvLds, lim_jv) == offsetof(EnvLds, jv3) + sizeof(float) * D_MAXCON * 3 && offsetof(EnvLds, sa) == offsetof(EnvLds, lim_jv) + sizeof(float) * 72 &&
                  offsetof(EnvLds, sw) == offsetof(EnvLds, sa) + sizeof(float) * 144, "jv3 | lim_jv | sa | sw must be contiguous");
    constexpr int ST = 6 * D_MAXOBJ + 1;
    const int jl = tid >> 3, r = tid & 7;
    const bool rowok = r < 6;
    const int rc = rowok ? r : 5;
    const float rmask = rowok ? 1.f : 0.f;
    const V3 o = ld3(s.xpos);
    // hulls that press on an object with an active row, and the union of their paths to the root (subtree of b = bodies [b, b + bsub[b]))
    bool mine = false;
    if (tid < D_NB)
        for (int c = s.con_start[tid]; c < s.con_start[tid + 1]; c++) { const int B = s.con_b2[c]; if (B >= D_NB && ((cmask >> (B - D_NB)) & 1u)) mine = true; }
    const unsigned touch = (unsigned)__ballot(mine);
    const unsigned subtree = tid < D_NB ? ((s.bsub[tid] >= 32 ? 0xFFFFFFFFu : ((1u << s.bsub[tid]) - 1u)) << tid) : 0u;
    const unsigned path = (unsigned)__ballot((touch & subtree) != 0u);
    if (path == 0u) return;
    const int npd = 3 * __popc(path) + 3;                       // dofs on the path (the root has six)
    const int ncols = 6 * __popc(cmask);
    // columns per round: the accumulators [9][nc][6] and the first s0 rows of the right-hand sides [npd][nc] share the 552 floats of jv3 | lim_jv | sa | sw,
    // the remaining rows go to sv[0, 144) (the hulls' spatial accelerations of y0: read by obj_coupling_u before this call, rewritten by the
    // back-substitution pass after it).  Round 4: with the 552 floats alone a humanoid lying on the table (14 - 20 bodies on the path) got 4 - 5
    // columns per round, i.e. three rounds for the push scene's twelve columns; now six, i.e. two.
    int ncmax = min(8, ncols);
    while (ncmax > 1 && (54 * ncmax > D_SCHUR_SCRATCH || (npd - (D_SCHUR_SCRATCH - 54 * ncmax) / ncmax) * ncmax > 144)) ncmax--;
    if (ncols > ncmax) ncmax = min(ncmax, (ncols + ((ncols + ncmax - 1) / ncmax) - 1) / ((ncols + ncmax - 1) / ncmax));      // same number of rounds, evenly filled
    float* ACC = s.jv3;                                          // [9 levels][ncmax][6]
    float* UUa = ACC + 54 * ncmax;                               // rows [0, s0) of [npd][ncmax]
    float* UUb = s.sv;                                           // rows [s0, npd)
    const int s0 = (D_SCHUR_SCRATCH - 54 * ncmax) / ncmax;
#define KP_UU(slot) ((slot) < s0 ? UUa + (slot) * ncmax : UUb + ((slot) - s0) * ncmax)
    const int nobj = s.nobj;
    struct Fac3 { float c[3], U[3], Di[3]; };                     // motion axis, U = IA s and 1 / D of a body's three dofs: this lane's row
    auto ldfac = [&](int d0) {
        Fac3 f;
#pragma unroll
        for (int j = 0; j < 3; j++) { f.c[j] = s.cdof[6 * (d0 + j) + rc]; f.U[j] = s.U[6 * (d0 + j) + rc]; f.Di[j] = s.Dinv[d0 + j]; }
        return f;
    };
    for (int c0 = 0; c0 < ncols; c0 += ncmax) {
        const int nc = min(ncmax, ncols - c0);
        const bool colok = jl < nc;
        const int jc = colok ? jl : 0;
        const int gcol = (cmask == 2u ? 6 : 0) + c0 + jc;        // column of the object system
        const int kcol = gcol / 6, icol = gcol - 6 * kcol;
        // ---- leaves -> root: bias forces.  The per-depth accumulators belong to lane (column, row) alone and live in its registers (accd[depth], indexed by
        // the wave-uniform depth of the body being visited): the walk is a serial chain over the path's bodies, and a read-modify-write of an LDS
        // accumulator per body put two LDS round trips on that chain (round 4: the walk was latency-bound; 176 k cycles per substep on the env that ends the
        // objects launch, whose path is the whole tree)
        float accd[D_NLEV];
#pragma unroll
        for (int k = 0; k < D_NLEV; k++) accd[k] = 0.f;
        // The walk is a serial chain over the path's bodies and every body needs nine factor words (cdof, U, 1 / D of its three dofs) + its depth from LDS:
        // loaded at the top of the body's own turn they put one LDS round trip per body on the chain (16 - 20 bodies x two directions x one or two rounds
        // x every factorisation on the envs the objects launch ends on).  They do not depend on the chain: the NEXT body's words are requested
        // before the current body's arithmetic and have arrived when its turn comes (same operations on the same values: results unchanged).
        unsigned todo = path;
        int b = 31 - __clz((int)todo);
        Fac3 cur = ldfac(b == 0 ? 3 : 6 + 3 * (b - 1));
        int dv = (int)s.bdep[b];
        while (true) {
            todo &= ~(1u << b);
            const int nb = todo ? 31 - __clz((int)todo) : 0;
            const Fac3 nxt = ldfac(nb == 0 ? 3 : 6 + 3 * (nb - 1));
            const int ndv = (int)s.bdep[nb];
            const int d = __builtin_amdgcn_readfirstlane(dv);
            float pA = accd[d];
            if ((touch >> b) & 1u) {                             // right-hand side: + (K_c e_icol)[r] for this hull's contacts on object kcol
                V3 Pi, Pr;                                       // rows of P = [[p]x ; 1]: K_c = P M P^T, entry (r, i) = P_r . (M P_i)
                for (int c = s.con_start[b]; c < s.con_start[b + 1]; c++) {
                    if ((int)s.con_b2[c] - D_NB != kcol) continue;
                    const V3 p = ld3(s.con_pos + 3 * c) - o;
                    if (icol == 0) Pi = v3(0.f, -p.z, p.y); else if (icol == 1) Pi = v3(p.z, 0.f, -p.x); else if (icol == 2) Pi = v3(-p.y, p.x, 0.f);
                    else Pi = v3(icol == 3 ? 1.f : 0.f, icol == 4 ? 1.f : 0.f, icol == 5 ? 1.f : 0.f);
                    if (r == 0) Pr = v3(0.f, -p.z, p.y); else if (r == 1) Pr = v3(p.z, 0.f, -p.x); else if (r == 2) Pr = v3(-p.y, p.x, 0.f);
                    else Pr = v3(r == 3 ? 1.f : 0.f, r == 4 ? 1.f : 0.f, r == 5 ? 1.f : 0.f);
                    const float* m = s.cM + 6 * c;
                    const V3 w = v3(m[0] * Pi.x + m[3] * Pi.y + m[4] * Pi.z, m[3] * Pi.x + m[1] * Pi.y + m[5] * Pi.z, m[4] * Pi.x + m[5] * Pi.y + m[2] * Pi.z);
                    pA += rmask * dot(Pr, w);
                }
            }
            const int slot0 = b == 0 ? 0 : 3 + 3 * __popc(path & ((1u << b) - 1u));
            for (int g = 0; g < (b == 0 ? 2 : 1); g++) {
                const int sl = b == 0 ? (g == 0 ? 3 : 0) : slot0;
                const Fac3 f = g == 0 ? cur : ldfac(0);          // the root's second group (its translations) is loaded in its turn
#pragma unroll
                for (int j = 2; j >= 0; j--) {
                    const float u = -sum8(rmask * f.c[j] * pA);
                    pA += rmask * f.U[j] * (u * f.Di[j]);
                    if (colok && r == 0) KP_UU(sl + j)[jc] = u;
                }
            }
            accd[d] = 0.f;
            if (b > 0) accd[d - 1] += (colok && rowok) ? pA : 0.f;
            if (!todo) break;
            b = nb; cur = nxt; dv = ndv;
        }
        KP_SYNC();
        // ---- root -> leaves: spatial accelerations (accd[depth] now holds the last visited body's at that depth: a body's parent is the last one
        // visited one level up, bodies being in depth-first order); at a touching hull, H_oh z accumulates per object
        float cpl0 = 0.f, cpl1 = 0.f;
        todo = path;
        b = __ffs((int)todo) - 1;
        auto ldrhs = [&](int bb, int g, float* uu) {                // the joint-space right-hand sides of a body's group, this lane's column
            const int sl = bb == 0 ? (g == 0 ? 0 : 3) : 3 + 3 * __popc(path & ((1u << bb) - 1u));
#pragma unroll
            for (int j = 0; j < 3; j++) uu[j] = KP_UU(sl + j)[jc];
        };
        cur = ldfac(b == 0 ? 0 : 6 + 3 * (b - 1));
        float cuu[3];
        ldrhs(b, 0, cuu);
        dv = (int)s.bdep[b];
        while (true) {
            todo &= todo - 1u;
            const int nb = todo ? __ffs((int)todo) - 1 : b;
            const Fac3 nxt = ldfac(nb == 0 ? 0 : 6 + 3 * (nb - 1));
            float nuu[3];
            ldrhs(nb, 0, nuu);
            const int ndv = (int)s.bdep[nb];
            const int d = __builtin_amdgcn_readfirstlane(dv);
            float a = (b > 0 && colok && rowok) ? accd[d - 1] : 0.f;
            for (int g = 0; g < (b == 0 ? 2 : 1); g++) {
                Fac3 f = cur;
                float uu[3] = {cuu[0], cuu[1], cuu[2]};
                if (g == 1) { f = ldfac(3); ldrhs(0, 1, uu); }
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float qdd = (uu[j] - sum8(rmask * f.U[j] * a)) * f.Di[j];
                    a += qdd * f.c[j];
                }
            }
            accd[d] = a;
            if ((touch >> b) & 1u) {
                KP_SYNC();
                if (colok && rowok) ACC[jc * 6 + r] = a;                    // the column's whole 6-vector is needed by every lane of the column: through LDS, for the touching hulls only
                KP_SYNC();
                const S6 ab = lds6(ACC + jc * 6);
                for (int c = s.con_start[b]; c < s.con_start[b + 1]; c++) {
                    const int B = s.con_b2[c];
                    if (B < D_NB) continue;
                    const S6 w = con_Kp(s, c, ab, o);
                    const float wr = r == 0 ? w.a.x : r == 1 ? w.a.y : r == 2 ? w.a.z : r == 3 ? w.l.x : r == 4 ? w.l.y : w.l.z;
                    if (B == D_NB) cpl0 += wr; else cpl1 += wr;
                }
            }
            if (!todo) break;
            b = nb; cur = nxt; dv = ndv;
            cuu[0] = nuu[0]; cuu[1] = nuu[1]; cuu[2] = nuu[2];
        }
        if (colok && rowok) {
            s.Sm[r * ST + gcol] += cpl0;
            if (nobj > 1) s.Sm[(6 + r) * ST + gcol] += cpl1;
        }
        KP_SYNC();
    }
#undef KP_UU
}

// the constraint solve with free objects in the scene: same Newton iteration as solve_constraints on the joint unknowns
// (q_humanoid, a_object...).  The Newton system is solved exactly by block elimination: the articulated-body pass factorises
// the humanoid block, 6 n_obj + 1 bias-only passes form the Schur complement on the objects when a hull touches one.
template <int NT>
__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap, const float* arm) {
    constexpr int ST = 6 * D_MAXOBJ + 1;
    const int nobj = s.nobj, no6 = 6 * nobj;
    // smooth acceleration of the objects: I_eff a = -bias wrench
    if (no6 > 0) {
        {
            const int n = no6;
            if (tid < n) {
                const int k = tid / 6, r = tid - 6 * k;
                float* row = s.Sm + ST * tid;
                for (int c = 0; c < n; c++) row[c] = (c / 6 == k) ? inert_entry(s.oIe + 10 * k, r, c - 6 * k) : 0.f;
                row[n] = -s.ofb[tid];
            }
            KP_SYNC();
        }
        dense_solve(s, no6, tid, true);
        if (tid < no6) s.oas[tid] = s.Sm[ST * tid + no6];
        KP_SYNC();
    }
    // The humanoid's share of the problem in the direct form of solve_constraints_direct: no qacc_smooth, no smooth articulated-body solve, the
    // start is the warm start, the first factorisation walks every level.  The objects keep their (cheap) smooth accelerations oas: their
    // share of the Gauss term stays 0.5 (oa - oas)^T I_eff (oa - oas), carried as omres = I_eff (oa - oas).
    if (s.ncon == 0 && s.nlim == 0) {
        aba_solve<NT, true>(s, P, L8, s.applied, s.qacc, false, tid, D_NLEV, s.fb);
        if (tid < no6) s.oa[tid] = s.oas[tid];
        KP_SYNC();
        return 0;
    }
    float* sacc = s.Mv;        // [24 + 2][6] over Mv + mres (+ x[0..3]): spatial accelerations induced by the iterate (hulls: of qacc; object slots: oa, for the first row evaluation)
    float* grad = s.qacc_s;    // the words qacc_smooth would occupy
    static_assert(offsetof(EnvLds, x) == offsetof(EnvLds, Mv) + 152 * sizeof(float), "sacc's object slots continue into x");
    spatial_accumulate<NT>(s, s.qacc, depth, tid, sacc);
    if (tid < no6) sacc[6 * D_NB + tid] = s.oa[tid];
    if (tid < nobj) sts6(s.omres + 6 * tid, inert_mul(s.oIe + 10 * tid, lds6(s.oa + 6 * tid) + (-1.0f) * lds6(s.oas + 6 * tid)));
    KP_SYNC();
    eval_rows<NT, true>(s, s.qacc, s.jar3, s.lim_jar, true, tid, sacc);
    float rowcost = 0.f;       // the rows' share of the cost at the iterate (the first line search evaluates it at alpha = 0)
    int it = 0, lev_hist = 1;
    bool done = false;          // left the loop through one of mj_solNewton's termination tests (not the iteration cap)
    const unsigned conlev = contact_levels(s, L8);
    // active set of the iterate (see solve_constraints_direct)
    float changed = 0.f, deep = 0.f;
    auto active_set = [&]() {
        changed = 0.f; deep = 0.f;
        for (int i = tid; i < D_NV; i += NT) {
            const float ex = (i >= 6 && s.lim_jar[i - 6] < 0.f) ? fabsf(s.lim_D[i - 6]) : 0.f;
            if (ex != s.extra[i]) changed = 1.f;
            s.extra[i] = ex;
            if (ex != 0.f) deep = fmaxf(deep, (float)s.bdep[s.dbody[i]]);     // active joint limit: its body's level is dirty
        }
        changed += active_set_changed<NT>(s, P, tid, deep);
        changed = (NT == 64) ? (__ballot(changed > 0.f) != 0ull ? 1.f : 0.f) : block_sum<NT>(s, changed, tid);     // one wave: a ballot is the whole reduction
        KP_SYNC();
    };
    active_set();
    for (; it < P.max_iter; it++) {
        // gradient: humanoid dofs (mres - J^T f) and object wrenches
        con_prepare<NT>(s, P, tid);                  // lane = contact: M_c at this iterate (object rows of the Hessian AND the hulls' contact inertia in aba_solve) and the contact force
        wrench_project<NT, true>(s, P, sacc, s.qacc, nullptr, grad, true, true, tid, arm, s.jv3, s.fb, s.applied);
        if (nobj > 0) obj_gradient(s, tid);
        float g2 = 0.f;
        for (int i = tid; i < D_NV; i += NT) { const float g = grad[i]; g2 += g * g; s.x[i] = -g; }
        if (tid < nobj) {   // gradient in joint coordinates: [g_l ; R^T (g_a + r x g_l)], r = o - body origin
            const S6 g = lds6(s.ogr + 6 * tid);
            const V3 t = g.a + cross(ld3(s.xpos) - ld3(s.oq + 7 * tid), g.l);
            g2 += dot(g.l, g.l) + dot(t, t);
        }
        g2 = block_sum<NT>(s, g2, tid);
        KP_SYNC();
        if (P.scale * sqrtf(g2) < P.tol) { done = true; break; }
        // search direction
        const bool refactor = it == 0 || changed > 0.f;
        nfact += refactor;
        int cslot = -1;                                   // object slot this lane's hull contact presses on with an active row
        for (int c = tid; c < s.con_start[D_NB]; c += NT) {
            if (s.con_b2[c] < D_NB) continue;
            const float jn = s.jar3[3 * c], jt1 = s.jar3[3 * c + 1], jt2 = s.jar3[3 * c + 2];
            bool act = false;
#pragma unroll
            for (int e = 0; e < 4; e++) act |= row_val(e, P.mu, jn, jt1, jt2) < 0.f;
            if (act) cslot = s.con_b2[c] - D_NB;
        }
        const unsigned cmask = (__ballot(cslot == 0) != 0ull ? 1u : 0u) | (__ballot(cslot == 1) != 0ull ? 2u : 0u);
        const bool couple = cmask != 0u;
        // H = [[H_hh, H_ho], [H_oh, H_oo]] by block elimination.  refactor: one articulated-body factorisation of H_hh (+ the object
        // rows), then -- when a hull presses on an object -- the Schur-complement columns (schur_columns); the dense object system; and
        // one more pass through the factorisation for the back-substitution
        if (refactor) {
            if (no6 > 0) obj_hessian(s, P, s.ogr, -1.0f, tid);
            lev_hist = max(lev_hist, first_clean_level<NT>(s, deep, tid));   // the clean range only shrinks within a substep
            // the substep's first factorisation walks every level: its clean levels hold M's own factors from then on
            aba_solve<NT, true>(s, P, L8, s.x, s.search, true, tid, it == 0 ? D_NLEV : lev_hist, nullptr, conlev);     // y0 = H_hh^-1 (-g_h); sv = its spatial accelerations
            if (couple) { obj_coupling_u(s, P, s.ot, tid); if (tid < no6) s.Sm[ST * tid + no6] += s.ot[tid]; KP_SYNC(); }     // rhs_o -= H_oh y0
        }
        if (refactor && couple) schur_columns(s, P, cmask, tid);          // S = H_oo - H_oh H_hh^-1 H_ho, all columns in one or two rounds
        if (!refactor) {                                                  // factors reused: y0 = H_hh^-1 (-g_h) through them, then the object right-hand side
            aba_resolve(s, L8, s.x, nullptr, s.search);
            if (no6 > 0) {
                if (tid < no6) s.Sm[ST * tid + no6] = -s.ogr[tid];
                KP_SYNC();
                if (couple) { obj_coupling_u(s, P, s.ot, tid); if (tid < no6) s.Sm[ST * tid + no6] += s.ot[tid]; KP_SYNC(); }
            }
        }
        if (no6 > 0) {
            dense_solve(s, no6, tid, refactor);
            if (tid < no6) s.osrch[tid] = s.Sm[ST * tid + no6];
            KP_SYNC();
            if (couple) {                                                 // back-substitution: search = H_hh^-1 (-g_h - H_ho da)
                hull_coupling_wrench(s, P, -1, s.osrch, 1.0f, tid);
                aba_resolve(s, L8, s.x, s.sw, s.search);
            }
        }
        if (tid < no6) s.sv[6 * D_NB + tid] = s.osrch[tid];
        KP_SYNC();
        eval_rows<NT, true>(s, s.search, s.jv3, s.lim_jv, false, tid);
        if (tid < nobj) sts6(s.oMv + 6 * tid, inert_mul(s.oIe + 10 * tid, lds6(s.osrch + 6 * tid)));
        KP_SYNC();
        float g0 = 2.0f * quad_form_M<NT>(s, arm, s.sv, sacc, s.search, nullptr, s.qacc, nullptr, tid);      // search^T (M qacc - qfrc_smooth) for the hulls ...
        float h0 = 2.0f * quad_form_M<NT>(s, arm, s.sv, s.sv, s.search, nullptr, s.search, nullptr, tid);
        for (int i = tid; i < D_NB * 6; i += NT) g0 += s.sv[i] * s.fb[i];
        for (int i = tid; i < D_NV; i += NT) g0 -= s.search[i] * s.applied[i];
        if (tid < no6) { g0 += s.osrch[tid] * s.omres[tid]; h0 += s.osrch[tid] * s.oMv[tid]; }                // ... and search^T I_eff (oa - oas) for the objects
        g0 = block_sum<NT>(s, g0, tid); h0 = block_sum<NT>(s, h0, tid);
        float rownew, rc0 = 0.f;
        const float alpha = line_search<NT>(s, P, g0, h0, tid, rownew, it == 0, rc0);
        if (it == 0) rowcost = rc0;
        if (!(alpha > 0.f)) { done = true; break; }
        for (int i = tid; i < D_NV; i += NT) s.qacc[i] += alpha * s.search[i];
        for (int i = tid; i < D_NB * 6; i += NT) sacc[i] += alpha * s.sv[i];
        if (tid < no6) { s.oa[tid] += alpha * s.osrch[tid]; s.omres[tid] += alpha * s.oMv[tid]; }
        for (int k = tid; k < 3 * s.ncon; k += NT) s.jar3[k] += alpha * s.jv3[k];
        for (int j = tid; j < D_NU; j += NT) if (s.lim_D[j] != 0.f) s.lim_jar[j] += alpha * s.lim_jv[j];
        KP_SYNC();
        // cost(old) - cost(new) in closed form: the Gauss term of hulls and objects is quadratic along the search direction (g0, h0 hold both
        // shares), the rows' share comes out of the line search's registers
        const float improvement = P.scale * ((rowcost - rownew) - alpha * (g0 + 0.5f * alpha * h0));
        rowcost = rownew;
        if (improvement < P.tol) { it++; done = true; break; }
        active_set();
        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }     // gradient(new) = (1 - alpha) gradient(old): see solve_constraints_direct
    }
    if (!done) ncap++;          // the solver stopped at opt.iterations: counted per env in diag (flags >> 8)
    return it;
}

// ---------------------------------------------------------------- the kernel
// FWD = true compiles the forward-only launch (sim.forward(): derived quantities at a new state, no substeps) as its own
// small kernel, so that the control-step kernel's launches are the only ones under its name in a profile.
// Global accesses to state that one job of kp_step_queue_kernel writes and a later job (another wave, maybe another XCD with its own
// L2) reads.  Relaxed agent-scope atomics are plain loads / stores with the sc1 bit: they go through to the memory side instead of
// living in an XCD's L2, so the hand-over needs no L2 write-back / invalidate fence.  Outside the queue kernel: ordinary accesses.
template <bool Q, typename V> __device__ __forceinline__ V gld(const V* p) {
    if constexpr (Q) return __hip_atomic_load(const_cast<V*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool Q, typename V> __device__ __forceinline__ void gst(V* p, V v) {
    if constexpr (Q) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// LEAN: the floor scenes' job-queue layout (EnvLdsLean).  Returns 1 when the lean layout could not hold a substep's contacts: the job has written nothing to HBM at
// that point and is re-run on the full layout (kp_step_overflow_kernel); 0 otherwise.
template <int NT, bool OBJ, bool FWD, bool Q = false, bool LEAN = false>
__device__ __forceinline__ int step_body(const StepArgs& A, const int env_in, const int part) {
    static_assert(!LEAN || (Q && !OBJ && !FWD && NT == 64), "the lean layout serves the floor scenes' job queue only");
    // this job's share of the control step; A stays the kernel's read-only argument block (a by-value copy that the job modifies is a private copy per job)
    const int n_substeps = FWD ? 0 : (part >= 0 ? (int)(((part < 8 ? A.part_sub_lo >> (8 * part) : A.part_sub_hi >> (8 * (part - 8)))) & 255ull) : A.n_substeps);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using SL = typename std::conditional<OBJ, EnvLdsObj, typename std::conditional<LEAN, EnvLdsLean, EnvLds>::type>::type;
    SL& s = *reinterpret_cast<SL*>(smem_raw);
    const int tid0 = threadIdx.x;
    // laundered per job: the per-lane table addresses of the load section below are invariant over the queue kernel's job loop, so LLVM formed them all at
    // kernel entry and kept ~20 of them (64-bit, per lane) in scratch for the whole launch, reloading them in every job.  With the store section's
    // addresses (below) that made the object queue kernel's scratch 464 B per lane = 6.6 MB per XCD on 224 resident waves, more than the 4 MB of L2
    // behind them: every job's spills and reloads went to HBM (298 MB per launch, 22 x the algorithmic bytes; one workgroup per env: 176 B per lane,
    // 45 MB).  Now 176 B per lane and 101 MB per launch (profiles/r04/traffic_with_and_without_queue.log).
    const int tid = kp_launder(tid0);
    const int env = env_in;
    if (env >= A.n_envs) return 0;
    if (A.env_mask && !A.env_mask[env]) return 0;
    const unsigned long long t_launch = __builtin_readcyclecounter();
    const DevTables& T = A.T;
    const Params& P = A.P;

    // ---- load: derived state first (the state the last forward pass ran on), then the real state
    const float* tq_row = A.target_qpos ? A.target_qpos + (size_t)env * D_NQ : nullptr;
    const float* act_row = A.action ? A.action + (size_t)env * D_NV : nullptr;
    // Torque hand-over between the jobs of a control step (stale mode): the stable-PD torque of substep k is a function of the kinematics of
    // substep k - 1, which sit in the LDS of the job that ran k - 1.  That job therefore also computes the torque of k (it is the same
    // solve its successor would start with) and hands over 78 floats; the successor then needs neither the derived state nor the forward
    // pass on it (17.7 k cycles of a 49 k hand-over).  The first job of a control step still starts from the derived state the previous
    // control step left.  Same code site, same operands: results do not depend on how the control step is cut.
    const bool torque_in = Q && P.stale && P.actuation && part > 0 && A.spd_next != nullptr;
    const bool torque_out = Q && P.stale && P.actuation && part >= 0 && part + 1 < A.n_parts && A.spd_next != nullptr;
    for (int i = tid; i < D_NQ; i += NT) s.qpos[i] = gld<Q>((torque_in ? A.qpos : A.qpos_d) + (size_t)env * D_NQ + (unsigned)(i));
    for (int i = tid; i < D_NV; i += NT) {
        s.qvel[i] = gld<Q>((torque_in ? A.qvel : A.qvel_d) + (size_t)env * D_NV + (unsigned)(i));
        if constexpr (LEAN) s.extra[i] = T.dof_armature[i]; else { s.arm[i] = T.dof_armature[i]; s.extra[i] = 0.f; }
        s.dbody[i] = T.dof_body[i]; s.qacc[i] = gld<Q>(A.warm + (size_t)env * D_NV + (unsigned)(i));
    }
    if (tid < D_NB) { s.bpar[tid] = (unsigned char)(T.body_parent[tid] < 0 ? 0 : T.body_parent[tid]); s.bsub[tid] = T.body_subtree[tid]; s.bdep[tid] = T.body_depth[tid]; }
    if (tid < 6) s.applied[tid] = 0.f;
    if (torque_in) for (int i = tid; i < 78; i += NT) s.applied[i] = gld<Q>(A.spd_next + (size_t)env * 80 + (unsigned)(i));
    if (tid < 25) s.IAa[22 * tid + 21] = 0.f;
    if (tid < 22) s.IAa[22 * 24 + tid] = 0.f;
    if (tid < 6) s.pAa[6 * 24 + tid] = 0.f;
    if (tid == 0) { s.ncon = 0; s.nlim = 0; s.flag = 0; }
    if constexpr (OBJ) {
        static_assert(NT == 64, "the object kernel runs one wavefront per environment");
        for (int i = tid; i < D_MAXGEOM * 17; i += NT) s.geom[i] = A.geoms[(size_t)env * D_MAXGEOM * 17 + (unsigned)(i)];
        const int ngs = A.ngeom[env];
        int nobj = 0, ng = ngs;
        if (tid < D_MAXGEOM) s.gobj[tid] = -1;
        for (int k = 0; k < D_MAXOBJ; k++) {
            const int oi = A.obj_slot ? A.obj_slot[(size_t)env * D_MAXOBJ + k] : -1;
            if (oi < 0 || oi >= T.n_obj) break;
            nobj = k + 1;
            if (tid < 7) s.oq[7 * k + tid] = gld<Q>(A.obj_qpos + (size_t)env * 35 + (unsigned)(7 * oi + tid));
            if (tid < 6) { s.ov[6 * k + tid] = gld<Q>(A.obj_qvel + (size_t)env * 30 + (unsigned)(6 * oi + tid)); s.oqa[6 * k + tid] = gld<Q>(A.obj_warm + (size_t)env * 6 * D_MAXOBJ + (unsigned)(6 * k + tid));
                           s.oqa_prev[6 * k + tid] = (A.warm_extrap != 0.f && Q && part > 0 && A.obj_warm2) ? gld<Q>(A.obj_warm2 + (size_t)env * 6 * D_MAXOBJ + (unsigned)(6 * k + tid)) : 0.f; }
            if (tid < 13) s.oc[13 * k + tid] = T.obj_inertial[13 * oi + tid];
            for (int gi = T.obj_geom_adr[oi]; gi < T.obj_geom_adr[oi + 1] && ng < D_MAXGEOM; gi++, ng++) {
                if (tid == 0) { s.gobj[ng] = (signed char)k; s.ggi[ng] = (unsigned char)gi; }
            }
        }
        if (tid == 0) { s.ngeom_static = ngs; s.ngeom = ng; s.nobj = nobj; }
    }
    KP_SYNC();
    float qd_save_q[(D_NQ + NT - 1) / NT], qd_save_v[(D_NV + NT - 1) / NT];
    int niter_total = 0, maxcon = 0, nfact_total = 0, ncap_total = 0;
    // Extrapolated start of the Newton solve (model option warm_extrap; round 5): see the solve's call site below.  The previous-but-one solution is
    // available from the control step's second substep on -- within a job in the words of qacc_s, across jobs through A.warm2 -- so whether a
    // substep extrapolates depends on its index in the control step only, never on how the control step is cut into jobs.
    const float beta = A.warm_extrap;
    bool have_prev = false;
    // lean layout (qacc_s is the solve's own vector there): a_{k-2} rides in a second HBM row of the env (A.warm3) between the solves of a job; the hand-over row
    // A.warm2 is read at the job's start and written at its end only, so that a job cut short by a contact overflow leaves no trace
    if (beta != 0.f && Q && part > 0 && A.warm2) {
        for (int i = tid; i < D_NV; i += NT) {
            const float v = gld<Q>(A.warm2 + (size_t)env * D_NV + (unsigned)(i));
            if constexpr (LEAN) gst<Q>(A.warm3 + (size_t)env * D_NV + (unsigned)(i), v); else s.qacc_s[i] = v;
        }
        have_prev = true;
    }
    const float* armv;            // dof armature as the solve reads it: the layout's copy, or the model table (lean layout: folded into extra for the eliminations)
    if constexpr (LEAN) armv = T.dof_armature; else armv = s.arm;
    auto store_readouts = [&](int lane, int e) {
        for (int i = lane; i < 72; i += NT) A.xpos[(size_t)e * 72 + (unsigned)(i)] = s.xpos[i];
        if (lane < D_NB) {                   // xipos = xpos + R ipos of the same forward pass
            const V3 xi = ld3(s.xpos + 3 * lane) + qrot(Q4{s.xquat[4 * lane], s.xquat[4 * lane + 1], s.xquat[4 * lane + 2], s.xquat[4 * lane + 3]}, ld3(T.body_ipos + 3 * lane));
            st3(A.xipos + (size_t)e * 72 + (unsigned)(3 * lane), xi);
        }
        for (int i = lane; i < 96; i += NT) A.xquat[(size_t)e * 96 + (unsigned)(i)] = s.xquat[i];
    };
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    const bool prof = A.prof != nullptr;
#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }
    unsigned long long tstart = 0;
    // One loop, one inlined copy of every phase.  Pass -1 is the entry pass: the forward pass on the derived state the job was handed
    // (qpos_d / qvel_d), after which the real state is loaded; passes 0 .. n - 1 are the substeps; in fresh mode (stale_kinematics = 0)
    // pass n is the forward pass on the final state.  Running the entry pass through the SAME code as a substep's forward pass is what
    // makes a control step cut into jobs bit-identical to an uncut one (two inlined copies need not contract their FMAs alike), and it
    // keeps the kernel's code a third shorter.
    int rem_after = 0;          // substeps of the control step's later jobs (queue_prio = 3)
    if constexpr (Q) for (int p = part + 1; p < A.n_parts; p++) rem_after += (int)(((p < 8 ? A.part_sub_lo >> (8 * p) : A.part_sub_hi >> (8 * (p - 8)))) & 255ull);
    const int last_pass = (n_substeps > 0 && (!P.stale || torque_out)) ? n_substeps : n_substeps - 1;
    for (int sub = torque_in ? 0 : -1; sub <= last_pass; sub++) {
        // per-lane invariants are re-derived from a laundered lane index every pass (see kp_launder): table addresses, the body's tree
        // level and offset, and -- at the top of each solve -- the Lane8 schedule
        const int tid = kp_launder(tid0);
        const int depth = tid < D_NB ? (int)s.bdep[tid] : -1;
        const V3 bpos = tid < D_NB ? ld3(T.body_pos + 3 * tid) : v3(0.f, 0.f, 0.f);
        const bool substep = sub >= 0 && sub < n_substeps;
        if constexpr (Q) {
            // queue_prio = 3: issue priority by the env's distance from the end of its control step, set anew at every substep -- the waves of a SIMD then progress
            // together: an env that started late or runs heavy (and so has more substeps left than its neighbours) is preferred until it has caught up
            if (A.queue_prio == 3 && substep) {
                const int lvl = (4 * (rem_after + n_substeps - sub) - 1) / A.n_substeps;
                if (lvl >= 3) __builtin_amdgcn_s_setprio(3); else if (lvl == 2) __builtin_amdgcn_s_setprio(2); else if (lvl == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            }
        }
        if (prof && sub == 0) tstart = __builtin_readcyclecounter();
        if (prof) t0 = __builtin_readcyclecounter();
        // stale mode: the controller sees M / bias of the previous forward pass (cinert, cdof, bias still in LDS)
        const bool spd_pass = torque_out && sub == n_substeps;           // the extra pass of a job that is not the control step's last: the successor's first torque
        if ((substep || spd_pass) && P.stale && P.actuation && !(torque_in && sub == 0)) { Lane8 La; La.init(kp_launder(tid), T.sched8); spd_torque_rfc<NT, OBJ>(s, T, P, La, tid, tq_row, act_row); }
        if (spd_pass) {
            for (int i = tid; i < 78; i += NT) gst<Q>(A.spd_next + (size_t)env * 80 + (unsigned)(i), s.applied[i]);
            break;
        }
        if (substep && !P.actuation) {              // model option "actuation" = 0: ctrl = qfrc_applied = 0 (torque-free motion; tests)
            for (int i = tid; i < 78; i += NT) s.applied[i] = 0.f;          // applied[6] ++ ctrl[72]
            KP_SYNC();
        }
        KP_T(0)
        // ---- mj_forward at the current state
        if (sub >= 0) {
#pragma unroll
            for (int n = 0; n < (D_NQ + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NQ) qd_save_q[n] = s.qpos[i]; }
#pragma unroll
            for (int n = 0; n < (D_NV + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NV) qd_save_v[n] = s.qvel[i]; }
        }
        forward_kin_bias<NT>(s, T, P, depth, bpos, tid);
        if (sub < 0) {          // entry pass: the derived state is what the forward pass just ran on (quaternion normalised); now the real state
#pragma unroll
            for (int n = 0; n < (D_NQ + NT - 1) / NT; n++) { int i = tid + n * NT; qd_save_q[n] = i < D_NQ ? s.qpos[i] : 0.f; }
#pragma unroll
            for (int n = 0; n < (D_NV + NT - 1) / NT; n++) { int i = tid + n * NT; qd_save_v[n] = i < D_NV ? s.qvel[i] : 0.f; }
            KP_SYNC();
            if (n_substeps > 0) {
                for (int i = tid; i < D_NQ; i += NT) s.qpos[i] = gld<Q>(A.qpos + (size_t)env * D_NQ + (unsigned)(i));
                for (int i = tid; i < D_NV; i += NT) s.qvel[i] = gld<Q>(A.qvel + (size_t)env * D_NV + (unsigned)(i));
                KP_SYNC();
            }
            continue;
        }
        if (!substep) continue;         // fresh mode's exit pass: outputs are the kinematics of the final state
        if constexpr (OBJ) {
            if (beta != 0.f) {          // the objects' share of the extrapolated start, in their joint coordinates (obj_forward turns oqa into the spatial start oa)
                if (tid < 6 * s.nobj) { const float cur = s.oqa[tid]; if (have_prev) s.oqa[tid] = cur + beta * (cur - s.oqa_prev[tid]); s.oqa_prev[tid] = cur; }
                KP_SYNC();
            }
            obj_forward(s, T, P, tid);
        }
        KP_T(1)
        collide<NT, OBJ>(s, T, P, tid);
        if constexpr (LEAN) {
            if (s.ncon > min(SL::MAXCON, A.lean_cap)) return 1;      // wave-uniform; nothing of this job has been stored yet (lean_cap: model option, tests lower it to exercise the hand-over)
            // stale kinematics: the control step's read-outs are those of its last substep's forward pass; the solve below re-uses the words of xquat
            if (P.stale && sub == n_substeps - 1 && part == A.n_parts - 1) store_readouts(tid, env);
        }
        if (A.dbg_contacts && sub == n_substeps - 1) {      // test hook: the contact set of the last collision pass (con_D still holds the distance)
            float* o = A.dbg_contacts + (size_t)env * (1 + D_MAXCON * 9);
            if (tid == 0) o[0] = (float)s.ncon;
            for (int c = tid; c < s.ncon; c += NT) {
                float* r = o + 1 + 9 * c;
                r[0] = (float)s.con_body[c]; r[2] = s.con_D[c]; r[3] = s.con_pos[3 * c]; r[4] = s.con_pos[3 * c + 1]; r[5] = s.con_pos[3 * c + 2];
                if constexpr (OBJ) { const EnvLdsObj& so = as_obj(s); r[1] = so.con_b2[c] < 0 ? -1.f : (float)so.con_b2[c]; r[6] = so.con_n[3 * c]; r[7] = so.con_n[3 * c + 1]; r[8] = so.con_n[3 * c + 2]; }
                else { r[1] = -1.f; r[6] = 0.f; r[7] = 0.f; r[8] = 1.f; }
            }
        }
        KP_T(2)
        make_constraint<NT, OBJ>(s, T, P, tid);                 // needs sv = cvel: before any aba_solve
        KP_T(3)
        if (!P.stale && P.actuation) { Lane8 La; La.init(kp_launder(tid), T.sched8); spd_torque_rfc<NT, OBJ>(s, T, P, La, tid, tq_row, act_row); }
        for (int i = tid; i < D_NV; i += NT) s.extra[i] = LEAN ? T.dof_armature[i] : 0.f;
        KP_SYNC();
        Lane8 L8; L8.init(kp_launder(tid), T.sched8);          // lives through the Newton solve
        // Start of the Newton solve.  MuJoCo starts from the previous substep's solution a_{k-1} (qacc_warmstart); with warm_extrap = beta != 0 the start is
        // a_{k-1} + beta (a_{k-1} - a_{k-2}).  The problem is strictly convex: the minimiser and the termination tests do not depend on the starting point, only
        // the path does (INTEGRATION deviation 6 already starts from the warm start where MuJoCo might pick qacc_smooth).  Where accelerations change
        // smoothly from substep to substep -- falls, impacts, a humanoid going down on the table -- the extrapolated point is closer to the solution and has its
        // active set more often; on quiet standing states the change is contact chatter and extrapolating it gains nothing or costs.  Measured (MI355X, 60 timed
        // steps, profiles/r05/warm_extrap_*.log), launch ms / Newton iterations per substep at beta = 0 | 0.5 | 0.75 | 1: objects 4.77 / 1.95 | 4.61 / 1.87 |
        // 4.52 / 1.87 | 4.62 / 1.94; random_init 3.17 / 2.96 | 2.75 / 2.30 | 2.51 / 1.88 | 2.75 / 2.27; tracked (the metric) 2.616 / 2.06 | 2.621 / 2.02 | 2.636 / 2.05 |
        // 2.697 / 2.14; wild_eval + 2 ... 4 % at 0.5 - 0.75.  Extrapolating only across large changes (a threshold on max |a_{k-1} - a_{k-2}|) is worse than either:
        // the large changes are the contact events, where the extrapolation is wrong.  Defaults: 0.75 when the scene's free objects are simulated, 0 (MuJoCo's
        // start) for floor scenes -- a caller whose envs are mostly falling (a policy at random init) sets 0.75.  a_{k-2} rides in the words of qacc_s between
        // solves (dead outside the
        // solve, where they hold the gradient); a_{k-1} is kept in two registers across the solve.  The objects' accelerations likewise (oqa / oqa_prev, in their joint coordinates).
        float keep0 = 0.f, keep1 = 0.f;
        if (beta != 0.f) {
            const int i0 = tid, i1 = tid + NT;
            auto prev = [&](int i) { if constexpr (LEAN) return gld<Q>(A.warm3 + (size_t)env * D_NV + (unsigned)(i)); else return s.qacc_s[i]; };
            if (i0 < D_NV) { keep0 = s.qacc[i0]; if (have_prev) s.qacc[i0] = keep0 + beta * (keep0 - prev(i0)); }
            if (NT < D_NV && i1 < D_NV) { keep1 = s.qacc[i1]; if (have_prev) s.qacc[i1] = keep1 + beta * (keep1 - prev(i1)); }
            KP_SYNC();
        }
        if constexpr (OBJ) {
            KP_T(4)                                            // no smooth solve: the Newton solve starts from the warm start (solve_constraints_direct)
            niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total, s.arm);
        } else {
            KP_T(4)
            niter_total += solve_constraints_direct<NT>(s, P, L8, depth, tid, nfact_total, ncap_total, armv);
            if constexpr (LEAN) {
                // the solve's body accelerations lived in the words of qpos | qvel: both come back from the registers that hold this substep's forward-pass state,
                // the root quaternion normalised as forward_kin_bias left it
#pragma unroll
                for (int n = 0; n < (D_NQ + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NQ) s.qpos[i] = qd_save_q[n]; }
#pragma unroll
                for (int n = 0; n < (D_NV + NT - 1) / NT; n++) { int i = tid + n * NT; if (i < D_NV) s.qvel[i] = qd_save_v[n]; }
                KP_SYNC();
                if (tid == 0) { const Q4 q = qnormalize(Q4{s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]}); s.qpos[3] = q.w; s.qpos[4] = q.x; s.qpos[5] = q.y; s.qpos[6] = q.z; }
                KP_SYNC();
            }
        }
        if (beta != 0.f) {
            const int i0 = tid, i1 = tid + NT;
            if constexpr (LEAN) {
                if (i0 < D_NV) gst<Q>(A.warm3 + (size_t)env * D_NV + (unsigned)(i0), keep0);
                if (NT < D_NV && i1 < D_NV) gst<Q>(A.warm3 + (size_t)env * D_NV + (unsigned)(i1), keep1);
            } else {
                if (i0 < D_NV) s.qacc_s[i0] = keep0;
                if (NT < D_NV && i1 < D_NV) s.qacc_s[i1] = keep1;
            }
            have_prev = true;
            KP_SYNC();
        }
        KP_T(5)
        maxcon = max(maxcon, s.ncon);
        // ---- semi-implicit Euler (mj_Euler, no damping)
        if constexpr (OBJ) obj_integrate(s, P, tid);       // reads xpos[0] (o) of this substep's forward pass
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
        KP_T(6)
    }
    if (prof && tid == 0 && n_substeps > 0) {
        pc[7] = __builtin_readcyclecounter() - tstart;
        for (int k = 0; k < 8; k++) A.prof[8 * (size_t)env + k] = pc[k];
    }
    // ---- store.  The row addresses are re-derived from laundered copies of the env and lane indices: the ones the load section formed at the top of the job
    // would otherwise stay live through the whole job (37 spilled 64-bit addresses per lane in the object queue kernel)
    const int tidS = kp_launder(tid0);
    const int envS = kp_launder_uniform(env);
    bool bad = false;
    for (int i = tidS; i < D_NQ; i += NT) { float v = s.qpos[i]; bad |= !(fabsf(v) < 1e10f); if (n_substeps > 0) gst<Q>(A.qpos + (size_t)envS * D_NQ + (unsigned)(i), v); }
    for (int i = tidS; i < D_NV; i += NT) { float v = s.qvel[i]; bad |= !(fabsf(v) < 1e10f); if (n_substeps > 0) { gst<Q>(A.qvel + (size_t)envS * D_NV + (unsigned)(i), v); gst<Q>(A.warm + (size_t)envS * D_NV + (unsigned)(i), s.qacc[i]); if (Q && beta != 0.f && A.warm2 && part >= 0 && part + 1 < A.n_parts) { float pv; if constexpr (LEAN) pv = gld<Q>(A.warm3 + (size_t)envS * D_NV + (unsigned)(i)); else pv = s.qacc_s[i]; gst<Q>(A.warm2 + (size_t)envS * D_NV + (unsigned)(i), pv); } } }
    if (!torque_out) {       // the derived state is read by the control step's NEXT first job only (a job that hands its torque over has no reader for it)
#pragma unroll
        for (int n = 0; n < (D_NQ + NT - 1) / NT; n++) { int i = tidS + n * NT; if (i < D_NQ) gst<Q>(A.qpos_d + (size_t)envS * D_NQ + (unsigned)(i), qd_save_q[n]); }
#pragma unroll
        for (int n = 0; n < (D_NV + NT - 1) / NT; n++) { int i = tidS + n * NT; if (i < D_NV) gst<Q>(A.qvel_d + (size_t)envS * D_NV + (unsigned)(i), qd_save_v[n]); }
    }
    // read-outs: the last job of the control step only (earlier jobs' copies could land later from another L2); lean layout in stale mode: already stored (above)
    if ((!Q || part == A.n_parts - 1) && !(LEAN && P.stale && n_substeps > 0)) store_readouts(tidS, envS);
    if constexpr (OBJ) {
        if (n_substeps > 0) {
            for (int k = 0; k < s.nobj; k++) {
                const int oi = A.obj_slot[(size_t)envS * D_MAXOBJ + k];
                if (tidS < 7) { const float v = s.oq[7 * k + tidS]; bad |= !(fabsf(v) < 1e10f); gst<Q>(A.obj_qpos + (size_t)envS * 35 + (unsigned)(7 * oi + tidS), v); }
                if (tidS < 6) { gst<Q>(A.obj_qvel + (size_t)envS * 30 + (unsigned)(6 * oi + tidS), s.ov[6 * k + tidS]); gst<Q>(A.obj_warm + (size_t)envS * 6 * D_MAXOBJ + (unsigned)(6 * k + tidS), s.oqa[6 * k + tidS]);
                                if (Q && A.warm_extrap != 0.f && A.obj_warm2 && part >= 0 && part + 1 < A.n_parts) gst<Q>(A.obj_warm2 + (size_t)envS * 6 * D_MAXOBJ + (unsigned)(6 * k + tidS), s.oqa_prev[6 * k + tidS]); }
            }
        }
    }
    if (bad) atomicOr(&s.flag, 1);
    KP_SYNC();
    if (tidS == 0 && A.diag && n_substeps > 0) {
        int* dg = A.diag + 4 * (size_t)envS;
        if (part > 0) {      // later job of the same control step: accumulate
            const int d3 = gld<Q>(dg + 3), d2 = gld<Q>(dg + 2);
            niter_total += gld<Q>(dg + 1); s.flag |= d2 & 255; ncap_total += d2 >> 8;
            maxcon = max(maxcon, d3 & 255); nfact_total += d3 >> 8;
        }
        gst<Q>(dg + 0, s.ncon); gst<Q>(dg + 1, niter_total); gst<Q>(dg + 2, (s.flag & 255) | (ncap_total << 8)); gst<Q>(dg + 3, maxcon | (nfact_total << 8));
    }
    if (tidS == 0 && A.cost && n_substeps > 0)
        gst<Q>(A.cost + envS, (part > 0 ? gld<Q>(A.cost + envS) : 0u) + (unsigned)((__builtin_readcyclecounter() - t_launch) >> 10));
    return 0;
}

// mj_fullM(model, M, data.qM)[:75, :75] and data.qfrc_bias[:75] as the reference's compute_desired_accel reads them
// (uhc/envs/humanoid_im.py:422-426): the dense joint-space inertia matrix (armature included) and the bias force of the state the
// derived quantities belong to (qpos_d / qvel_d: what mujoco-py's data holds between sim.step() calls).  The simulator itself
// never forms either (every solve is an articulated-body pass); this read-out builds column j of M as the projection of the body
// wrenches I_b a_b(e_j) summed over subtrees (a composite-rigid-body product M e_j), and qfrc_bias as the same projection of the
// RNE body wrenches.  Off the hot path: 76 projections per env.
__global__ __launch_bounds__(64) void kp_mass_kernel(StepArgs A, float* __restrict__ Mout, float* __restrict__ bias_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    EnvLds& s = *reinterpret_cast<EnvLds*>(smem_raw);
    constexpr int NT = 64;
    const int tid = threadIdx.x, env = blockIdx.x;
    if (env >= A.n_envs) return;
    const DevTables& T = A.T;
    const Params& P = A.P;
    const int depth = tid < D_NB ? T.body_depth[tid] : -1;
    const V3 bpos = tid < D_NB ? ld3(T.body_pos + 3 * tid) : v3(0.f, 0.f, 0.f);
    for (int i = tid; i < D_NQ; i += NT) s.qpos[i] = A.qpos_d[(size_t)env * D_NQ + i];
    for (int i = tid; i < D_NV; i += NT) { s.qvel[i] = A.qvel_d[(size_t)env * D_NV + i]; s.arm[i] = T.dof_armature[i]; s.dbody[i] = T.dof_body[i]; }
    if (tid < D_NB) { s.bpar[tid] = (unsigned char)(T.body_parent[tid] < 0 ? 0 : T.body_parent[tid]); s.bsub[tid] = T.body_subtree[tid]; s.bdep[tid] = T.body_depth[tid]; }
    KP_SYNC();
    forward_kin_bias<NT>(s, T, P, depth, bpos, tid);
    // qfrc_bias: subtree sums of the RNE body wrenches projected on the motion axes (backward half of mj_rne)
    for (int it = tid; it < D_NB * 6; it += NT) {
        const int b = it / 6, c = it - 6 * b, n = s.bsub[b];
        float acc = 0.f;
        for (int k = b; k < b + n; k++) acc += s.fb[6 * k + c];
        s.sa[it] = acc;
    }
    KP_SYNC();
    if (bias_out) for (int d = tid; d < D_NV; d += NT) bias_out[(size_t)env * D_NV + d] = dot6(lds6(s.cdof + 6 * d), lds6(s.sa + 6 * s.dbody[d]));
    KP_SYNC();
    if (!Mout) return;
    for (int j = 0; j < D_NV; j++) {
        for (int i = tid; i < D_NV; i += NT) s.x[i] = i == j ? 1.f : 0.f;
        KP_SYNC();
        spatial_accumulate<NT>(s, s.x, depth, tid);
        wrench_project<NT, false>(s, P, s.sv, s.x, nullptr, s.qacc_s, true, false, tid, s.arm);
        for (int d = tid; d < D_NV; d += NT) Mout[((size_t)env * D_NV + d) * D_NV + j] = s.qacc_s[d];
        KP_SYNC();
    }
}

// Launch order for the next control step: envs sorted by the cycles they took in the last one, longest first (counting sort on
// 256 bins scaled to the maximum, one workgroup).  With N envs on 8 x 256 wave slots the launch otherwise ends on whichever
// long env (many contacts, many Newton iterations) happened to start in the second round.
__global__ __launch_bounds__(1024) void k_lpt_order(int n, const unsigned* __restrict__ cost, int* __restrict__ order) {
    __shared__ unsigned hist[256], base[256], cmax;
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) cmax = 1;
    __syncthreads();
    unsigned m = 0;
    for (int i = tid; i < n; i += 1024) m = max(m, cost[i]);
    atomicMax(&cmax, m);
    __syncthreads();
    const float sc = 255.f / (float)cmax;
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[255 - (int)((float)cost[i] * sc)], 1u);
    __syncthreads();
    if (tid == 0) { unsigned a = 0; for (int b = 0; b < 256; b++) { base[b] = a; a += hist[b]; } }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&base[255 - (int)((float)cost[i] * sc)], 1u)] = i;
}

template <int NT, bool OBJ>
__global__ __launch_bounds__(NT, (NT == 64 ? 2 : 1)) void kp_step_kernel(StepArgs A) {
    step_body<NT, OBJ, false>(A, A.order ? A.order[blockIdx.x] : (int)blockIdx.x, -1);
}
template <int NT, bool OBJ>
__global__ __launch_bounds__(NT, (NT == 64 ? 2 : 1)) void kp_forward_kernel(StepArgs A) { step_body<NT, OBJ, true>(A, (int)blockIdx.x, -1); }

// Same control step, scheduled in finer grains.  4096 envs on 8 x 256 wave slots are two rounds of whole-control-step jobs whose
// lengths spread 2.9 M .. 5.4 M cycles, so a third of a kp_step_kernel launch is its tail (tools/launch_balance.py).  Here one
// resident wavefront per slot pulls jobs (env, part) from a FIFO in HBM: a job is a few substeps of one env, and finishing it
// publishes the env's next part at the tail.  An env therefore migrates between waves (its state already round-trips through HBM at
// job boundaries exactly as it does between launches), the makespan becomes sum / slots + about one job, and the arithmetic is
// that of kp_step_kernel bit for bit.  Progress: indices are claimed in order, so a wave that waits for entry idx waits for a publish by
// a wave that is running a job; if nothing is running every entry below n_envs * n_parts has been published.  A bounded wait (2 s) turns
// any violation of that argument into an error flag (jobctr[2]) instead of a hung queue.
// Memory ordering of the hand-over, two variants (model option "queue_fence"):
//   1 (default)  the state arrays a later job reads are written / read as relaxed agent-scope atomics (sc1 write-through accesses), and the
//                publish is bracketed by __builtin_amdgcn_fence(release, agent) on the producing wave and fence(acquire, agent) after the
//                consuming load: a release / acquire pair, correct by the HIP memory model.
//   0            the round-1 form without the two fences: s_waitcnt vmcnt(0) alone orders the sc1 stores before the publish.  Correct on
//                gfx950 by what sc1 means in hardware only.
// Both are bit-identical in results (test_job_queue_schedule_is_bit_identical).  Measured on MI355X, 4096 envs standing + contact
// (tools/queue_fence_bench.py, profiles/r02/queue_fence_bench.log): 3.818 ms (0) vs 3.837 ms (1) per launch -- the fences cost 0.5 %, so
// the variant that is correct by construction is the default.
// LEAN (floor scenes): the EnvLdsLean layout and a register budget for three waves per SIMD; a job whose contacts do not fit the layout is handed, with the env's
// remaining jobs, to kp_step_overflow_kernel (launched right behind this kernel, same stream).
template <bool OBJ, bool LEAN = false>
__global__ __launch_bounds__(64, (LEAN ? 3 : 2)) void kp_step_queue_kernel(StepArgs A) {
    // jobctr: [0] head (claimed), [1] tail (published), [2] stalled flag; on cache lines of their own, away from the head / tail every claim and publish hits:
    //         [16] jobs that were never queued because the finishing wave ran them itself; [32] time (40 ns units) and [33] substeps of the jobs finished so
    //         far in this launch; [48], [49] the same sums of the previous launch (the mean behind "heavy": constant during the launch, read once per wave)
    const unsigned total = (unsigned)A.n_envs * (unsigned)A.n_parts;
    const unsigned prev_tot = A.jobctr[48], prev_cnt = A.jobctr[49];
    for (;;) {
        unsigned idx = 0;
        if (threadIdx.x == 0) idx = atomicAdd(&A.jobctr[0], 1u);
        idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
        if (idx >= total) return;
        unsigned e, spins = 0;
        unsigned long long t_wait = 0;
        while ((e = __hip_atomic_load(&A.jobq[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0xFFFFFFFFu) {
            // Back-off: the first polls come 0.9 us apart (a job published mid-launch is picked up at once); a wave that has waited longer is in the
            // launch's tail, where the work left is the costliest envs' chains that their own waves keep (queue_heavy) -- it polls every 3.4 ... 27 us.
            // Every poll is a read that goes through to memory (agent-scope atomic) on behalf of a wave that has nothing to do.
            if (spins < 16u) __builtin_amdgcn_s_sleep(32);
            else for (unsigned k = 0; k <= min((spins - 16u) >> 3, 7u); k++) __builtin_amdgcn_s_sleep(127);
            // entries at and beyond total - jobctr[16] will never be published (the counter only grows): nothing left for this wave
            if (idx >= total - __hip_atomic_load(&A.jobctr[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            // a bounded wait (2 s of the 100 MHz clock) turns any violation of the progress argument into an error flag instead of a hung queue
            if (++spins == 1u) t_wait = __builtin_amdgcn_s_memrealtime();
            else if ((spins & 63u) == 0u && __builtin_amdgcn_s_memrealtime() - t_wait > 200000000ull) { if (threadIdx.x == 0) atomicExch(&A.jobctr[2], 1u); return; }
        }
        asm volatile("" ::: "memory");                        // the job's (sc1) state loads stay behind the load that saw the entry
        if (A.queue_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // memory-model variant: acquire side of the publish below
        e = (unsigned)__builtin_amdgcn_readfirstlane((int)e);
        const int env = (int)(e & 0xFFFFFFu);
        int part = (int)(e >> 24);
        // With more envs than slots the launch ends on the envs whose first job had to wait for a slot: their chain starts one job late, and every pass through
        // the FIFO (behind the early envs' later jobs) delays it further.  queue_late: such an env is never queued again.
        const bool late = A.queue_late && part == 0 && idx >= gridDim.x;
        // (running a late env's whole control step as ONE job, without the hand-overs between its parts, was measured and dropped: 2.216 vs 2.194 ms,
        // profiles/r06/lean_schedule_knobs5.log)
        // Issue priority: the launch ends on its costliest envs' serial chains, and a wave shares its SIMD's issue slots with one other wave.  A wave that
        // runs a job of an env known to be heavy -- one of the first queue entries when the first jobs were queued longest-env-first (k_lpt_order), or an
        // env it kept because its last job ran long (below) -- raises its own priority, so the SIMD's arbiter prefers it over its neighbour
        // (s_setprio: scheduling only, results do not depend on it); every other job runs at the default priority.
        if (A.queue_prio == 1) { if (A.order_valid && part == 0 && idx < (unsigned)A.n_envs / 16u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
        // queue_prio = 2: most remaining work first.  With more envs than slots the envs whose first job starts late are the ones the launch ends on; a wave on an
        // env's first job outranks its SIMD neighbours on second jobs, those outrank last jobs (at the launch's start every wave is on a first job: no preference)
        if (A.queue_prio == 2) { if (part == 0) __builtin_amdgcn_s_setprio(2); else if (part + 1 < A.n_parts) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        for (;;) {
            const unsigned long long tj = __builtin_amdgcn_s_memrealtime();          // 100 MHz ticks: only ratios of job times are used
            const int overflow = step_body<64, OBJ, false, true, LEAN>(A, env, part);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every lane's write-through state store has been acknowledged ...
            if (A.queue_fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // memory-model variant: release all of this wave's stores at agent scope
            if (LEAN && overflow) {
                // more contacts than the lean layout holds: this job (which has stored nothing) and the env's later jobs go to the overflow list; none of them
                // will be published to this queue, which the waiting waves learn from jobctr[16] like they do for jobs a wave kept
                if (threadIdx.x == 0) {
                    A.ovfq[atomicAdd(&A.jobctr[64], 1u)] = (unsigned)env | ((unsigned)part << 24);
                    atomicAdd(&A.jobctr[16], (unsigned)(A.n_parts - 1 - part));
                }
                break;
            }
            if (part + 1 >= A.n_parts) break;
            // An env whose jobs run long is the one the launch will end on: with two envs per slot every hand-over through the FIFO costs it about one
            // job's length of waiting.  The wave that finds its env heavy -- cycles per substep above queue_heavy % of the launch's running mean --
            // keeps it: it runs the next job itself, at once, and the queue gets one entry fewer (jobctr[3]).  Who runs a job never changes its result.
            int keep = 0;
            if (A.queue_heavy > 0 && threadIdx.x == 0) {
                const unsigned nsub = (unsigned)(((part < 8 ? A.part_sub_lo >> (8 * part) : A.part_sub_hi >> (8 * (part - 8)))) & 255ull);
                const unsigned dt = (unsigned)((__builtin_amdgcn_s_memrealtime() - tj) >> 2);      // 40 ns units: the launch's sum stays far below 2^32
                atomicAdd(&A.jobctr[32], dt); atomicAdd(&A.jobctr[33], nsub);
                // the yardstick is the PREVIOUS launch's mean (k_queue_init moves it to jobctr[48..49]): this launch's own running mean is made of the jobs
                // that finished first, i.e. of the light ones.  Jobs of fewer than four substeps are too noisy a sample of their env (one extra Newton
                // iteration in one substep is + 25 %)
                keep = prev_cnt >= 512u && nsub >= 4u && (unsigned long long)dt * prev_cnt * 100ull > (unsigned long long)prev_tot * nsub * (unsigned)A.queue_heavy;
            }
            keep = __builtin_amdgcn_readfirstlane(keep) | (int)late;
            if (!keep) {
                if (threadIdx.x == 0) {                                    // ... before the env's next job becomes visible
                    const unsigned pos = atomicAdd(&A.jobctr[1], 1u);
                    __hip_atomic_store(&A.jobq[pos], (unsigned)env | ((unsigned)(part + 1) << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
            if (threadIdx.x == 0) atomicAdd(&A.jobctr[16], 1u);
            if (A.queue_prio == 1) __builtin_amdgcn_s_setprio(3);
            part++;
            if (A.queue_prio == 2) { if (part + 1 < A.n_parts) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }      // a kept (heavy) env stays one rank above its part
        }
    }
}

// The jobs the lean queue kernel could not hold (more than EnvLdsLean::MAXCON contacts in a substep), run on the full layout: a wave claims an entry and runs that
// job and the env's remaining jobs itself.  Launched behind every lean queue launch; with an empty list (the rule: floor scenes hold 7 - 10 contacts) every wave
// leaves at its first read.  The arithmetic is step_body's: which kernel ran a job changes nothing but the contact capacity.
__global__ __launch_bounds__(64, 2) void kp_step_overflow_kernel(StepArgs A) {
    for (;;) {
        unsigned idx = 0;
        if (threadIdx.x == 0) idx = atomicAdd(&A.jobctr[65], 1u);
        idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
        if (idx >= __hip_atomic_load(&A.jobctr[64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&A.ovfq[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int env = (int)(e & 0xFFFFFFu);
        for (int part = (int)(e >> 24); part < A.n_parts; part++) {
            step_body<64, false, false, true, false>(A, env, part);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (A.queue_fence) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
        }
    }
}

// order (optional): the envs' first jobs enter the FIFO longest-env-first (k_lpt_order on the previous control step's cycles) instead of in
// env order, so that the env whose three jobs take longest does not also start last
__global__ void k_queue_init(int n_envs, unsigned total, unsigned* __restrict__ jobq, unsigned* __restrict__ jobctr, const int* __restrict__ order) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) jobq[i] = i < (unsigned)n_envs ? (order ? (unsigned)order[i] : i) : 0xFFFFFFFFu;
    if (i == 0) { jobctr[0] = 0u; jobctr[1] = (unsigned)n_envs; jobctr[16] = 0u; jobctr[64] = 0u; jobctr[65] = 0u; jobctr[48] = jobctr[32]; jobctr[49] = jobctr[33]; jobctr[32] = 0u; jobctr[33] = 0u; }
}

}  // namespace kp
