// kp_sim.hip -- host side of libkinpoly_sim.so: model upload, per-env state buffers, kernel launches,
// and the extern "C" entry points declared in include/kinpoly_sim.h.  No torch types anywhere.
#include <hip/hip_runtime.h>

#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <unistd.h>
#include <string>
#include <vector>

#include "../../include/kinpoly_sim.h"
#include "kp_model.hpp"
#include "kp_obs_kernels.hpp"
#include "kp_rollout_kernels.hpp"
#include "kp_policy_kernels.hpp"
#include "kp_step_kernel.hpp"

namespace {
thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }
#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define HIP_OK_NULL(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fail(std::string(#expr) + ": " + hipGetErrorString(e_)); return nullptr; } } while (0)
}  // namespace

struct kp_model {
    kp::HostModel h;
    int contact = 1, limits = 1, stale = 1, solver_iter = 100, threads = 64, dynamic_objects = 1, lpt_order = -1, substeps_per_job = 4, queue_slots = 0, job_taper = 1, queue_fence = 1, queue_heavy = 160, queue_prio = -1, queue_late = -1, job_auto = 1, lean_queue = 1, lean_adaptive = 1, lds_pad = 0, lean_cap = kp::EnvLdsLean::MAXCON;
    float warm_extrap = -1.f;      // < 0: automatic (0.75 when the scene's free objects are simulated, 0 otherwise); see kp_step_kernel.hpp
    int planemesh_max = 3; double planemesh_tol = 0.3;   // mjc_PlaneConvex's maxplanemesh / tolplanemesh (the blob's `planemesh`)
    int actuation = 1;            // 0: no stable-PD torque, no residual force (ctrl = qfrc_applied = 0): torque-free flight for the energy test
    double solver_tol = 1e-8, gravity_z = -9.81, gravity_x = 0.0, gravity_y = 0.0;   // solver_iter / solver_tol: mjOption.iterations / tolerance of the reference model (kp_model_load)
};

struct kp_sim {
    const kp_model* model = nullptr;
    int n = 0, device = 0;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;
    kp::DevTables T{};
    kp::Params P{};
    float *qpos = nullptr, *qvel = nullptr, *qpos_d = nullptr, *qvel_d = nullptr, *warm = nullptr, *warm2 = nullptr;
    float *xpos = nullptr, *xquat = nullptr, *xipos = nullptr, *scratch = nullptr;
    float *prev_bquat = nullptr, *prev_hpos = nullptr, *diffw = nullptr;
    float *t_qpos = nullptr, *t_wbpos = nullptr, *t_wbquat = nullptr, *t_bquat = nullptr, *t_com = nullptr;
    int* diag = nullptr;
    int* order = nullptr; unsigned* cost = nullptr;   // launch order of the control-step kernel (k_lpt_order)
    unsigned *jobq = nullptr, *jobctr = nullptr, *ovfq = nullptr;      // ovfq: jobs handed from the lean queue kernel to kp_step_overflow_kernel
    float* warm3 = nullptr;
    // how many envs the lean layout handed to the overflow kernel, read back WITHOUT waiting: every lean launch copies its count to pinned memory and records an
    // event; a later launch whose predecessor's event has completed looks at the number.  When more than 1 / 64 of the envs overflow (a policy at random init resets
    // every env onto a garbage pose, half buried, with 30 - 70 contacts) the overflow kernel's second pass over them costs more than the lean layout saves: the
    // next 64 launches use the full layout, then the lean one is tried again.  Results do not depend on the layout.
    unsigned* ovf_host = nullptr; hipEvent_t ovf_ev[2] = {nullptr, nullptr}; bool ovf_pending[2] = {false, false};
    long launch_index = 0, lean_off_until = -1; int lean_fallbacks = 0;
    unsigned ovf_last = 0;            // the most recent count that has arrived: sizes the overflow kernel's grid (an empty 2048-workgroup launch costs 34 us)      // job FIFO of kp_step_queue_kernel
    float* spd_next = nullptr;                        // [N, 80] torque hand-over between the jobs of a control step
    int jobq_cap = 0, wave_slots = 2048;
    int q_nsub = -1, q_obj = -1;                      // what the queue's "heavy job" yardstick (jobctr[32..33] -> [48..49]) was measured on
    unsigned long long* prof = nullptr;
    float* dbg_contacts = nullptr;
    float *obj_qpos = nullptr, *geoms = nullptr;      // [N,35], [N,8,17]
    float *obj_qvel = nullptr, *obj_warm = nullptr, *obj_warm2 = nullptr;   // [N,30], [N,12], [N,12]
    signed char* obj_slot = nullptr;                  // [N,2]
    int* ngeom = nullptr;
    const float* d_obj_geoms = nullptr; const float* d_obj_mass = nullptr; int n_obj_geoms = 0, n_obj = 0;
    bool has_objects = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    std::vector<hipEvent_t> ring;  // event pairs of the recorded launches
    int ring_used = 0;
    bool ring_on = false;
    hipEvent_t last0 = nullptr, last1 = nullptr;
};

namespace {

template <typename T, typename S>
const T* upload(kp_sim* s, const std::vector<S>& src, bool* ok) {
    std::vector<T> tmp(src.size());
    for (size_t i = 0; i < src.size(); i++) tmp[i] = (T)src[i];
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(sizeof(T) * tmp.size(), 16)) != hipSuccess) { *ok = false; return nullptr; }
    s->allocs.push_back(d);
    if (hipMemcpy(d, tmp.data(), sizeof(T) * tmp.size(), hipMemcpyHostToDevice) != hipSuccess) { *ok = false; return nullptr; }
    return (const T*)d;
}

float* dalloc(kp_sim* s, size_t count, bool* ok) {
    void* d = nullptr;
    if (hipMalloc(&d, sizeof(float) * count) != hipSuccess) { *ok = false; return nullptr; }
    s->allocs.push_back(d);
    hipMemset(d, 0, sizeof(float) * count);
    return (float*)d;
}

bool build_tables(kp_sim* s) {
    const kp::HostModel& m = s->model->h;
    using namespace kp;
    bool ok = true;
    auto& T = s->T;
    T.body_pos = upload<float>(s, m.body_pos, &ok); T.body_ipos = upload<float>(s, m.body_ipos, &ok);
    T.body_mass = upload<float>(s, m.body_mass, &ok); T.body_inertia = upload<float>(s, m.body_inertia, &ok);
    T.body_rbound = upload<float>(s, m.body_rbound, &ok); T.mesh_rbound = upload<float>(s, m.mesh_rbound, &ok);
    std::vector<double> invw(NB), liminvw(NU), lo(NU), hi(NU);
    for (int b = 0; b < NB; b++) invw[b] = m.body_invweight0[2 * b];
    for (int j = 0; j < NU; j++) { liminvw[j] = m.dof_invweight0[6 + j]; lo[j] = m.jnt_range[2 * j]; hi[j] = m.jnt_range[2 * j + 1]; }
    T.body_invw = upload<float>(s, invw, &ok); T.lim_invw = upload<float>(s, liminvw, &ok);
    T.jnt_lo = upload<float>(s, lo, &ok); T.jnt_hi = upload<float>(s, hi, &ok);
    T.dof_armature = upload<float>(s, m.dof_armature, &ok);
    T.kp = upload<float>(s, m.kp, &ok); T.kd = upload<float>(s, m.kd, &ok); T.tlim = upload<float>(s, m.torque_lim, &ok);
    T.ascale = upload<float>(s, m.a_scale, &ok);
    s->diffw = const_cast<float*>(upload<float>(s, m.body_diffw, &ok));
    T.verts = upload<float>(s, m.verts, &ok);
    T.vert_adr = upload<uint16_t>(s, m.vert_adr, &ok);
    T.vert_nbr_adr = upload<uint16_t>(s, m.vert_nbr_adr, &ok); T.vert_nbr = upload<uint8_t>(s, m.vert_nbr, &ok);
    T.dof_body = upload<uint8_t>(s, m.dof_body, &ok);
    T.body_parent = upload<int8_t>(s, m.body_parent, &ok); T.body_depth = upload<uint8_t>(s, m.body_depth, &ok);
    T.body_subtree = upload<uint8_t>(s, m.body_subtree, &ok); T.jnt_limited = upload<uint8_t>(s, m.jnt_limited, &ok);
    // layout assumptions of the kernels: bodies in DFS order (subtree(b) = [b, b + size)), body b > 0 owns dofs 6+3(b-1)..+2
    for (int b = 1; b < NB; b++) {
        if (m.body_parent[b] >= b) return false;
        for (int j = 0; j < 3; j++) if (m.dof_body[6 + 3 * (b - 1) + j] != b) return false;
    }
    for (int b = 0; b < NB; b++)
        for (int k = b + 1; k < NB; k++) {
            bool desc = false;
            for (int p = m.body_parent[k]; p >= 0; p = m.body_parent[p]) if (p == b) desc = true;
            if (desc != (k < b + m.body_subtree[b])) return false;
        }
    std::vector<int> lev_start(D_NLEV + 2, 0), lev_body;
    for (int lev = 0; lev <= D_NLEV; lev++) {
        lev_start[lev] = (int)lev_body.size();
        for (int b = 0; b < NB; b++) if (m.body_depth[b] == lev) lev_body.push_back(b);
    }
    lev_start[D_NLEV + 1] = (int)lev_body.size();
    if ((int)lev_body.size() != NB || lev_start[D_NLEV] != NB) return false;  // tree deeper than D_NLEV levels
    T.lev_start = upload<uint8_t>(s, lev_start, &ok); T.lev_body = upload<uint8_t>(s, lev_body, &ok);
    // static schedule of the 8-lanes-per-body tree passes (Lane8, kp_step_kernel.hpp): lane = 8 * slot + row, 28 words per lane, built
    // here once so that the kernels load it instead of deriving it (they re-read it at the top of every solve to keep it out of the
    // registers in between)
    std::vector<unsigned> tab(64 * 28, 0u);
    unsigned multi = 0;
    for (int lev = 0; lev < D_NLEV; lev++)
        for (int i = lev_start[lev]; i < lev_start[lev + 1]; i++) {
            int nc = 0;
            for (int k = lev_body[i] + 1; k < NB; k++) if (m.body_parent[k] == lev_body[i]) nc++;
            if (nc > 3) return false;
            if (nc > 1) multi |= 1u << lev;
        }
    for (int tid = 0; tid < 64; tid++) {
        const int slot = tid >> 3, r = tid & 7;
        unsigned long long sb = 0, sp = 0, sc[3] = {0, 0, 0};
        for (int lev = 0; lev < D_NLEV; lev++) {
            const int nb = lev_start[lev + 1] - lev_start[lev];
            if (nb > 8) return false;
            unsigned long long b = 31, par = 31, ch[3] = {24, 24, 24};      // no body in this slot: 31; absent child: the zero record 24
            if (slot < nb) {
                const int body = lev_body[lev_start[lev] + slot];
                b = (unsigned)body; par = m.body_parent[body] < 0 ? 31u : (unsigned)m.body_parent[body];
                int nc = 0;
                for (int k = body + 1; k < NB; k++) if (m.body_parent[k] == body) ch[nc++] = (unsigned)k;
            }
            sb |= b << (5 * lev); sp |= par << (5 * lev);
            for (int c = 0; c < 3; c++) sc[c] |= ch[c] << (5 * lev);
        }
        unsigned* w = tab.data() + 28 * tid;
        w[0] = (unsigned)sb; w[1] = (unsigned)(sb >> 32); w[2] = (unsigned)sp; w[3] = (unsigned)(sp >> 32);
        for (int c = 0; c < 3; c++) { w[4 + 2 * c] = (unsigned)sc[c]; w[5 + 2 * c] = (unsigned)(sc[c] >> 32); }
        w[10] = multi;
        for (int k = 0; k < 8; k++) {
            const int kx = k < 4 ? k : 11 - k;          // XOR order {0,1,2,3,7,6,5,4}: register k holds column r ^ kx
            const int c = r ^ kx;
            int idx = 0; float sg = 0.f;
            if (r < 6 && c < 6) {                       // entry (r, c) of the 6x6 spatial inertia from the 10 floats [Ixx Iyy Izz Ixy Ixz Iyz | h | m]
                if (r < 3 && c < 3) { idx = (r == c) ? r : r + c + 2; sg = 1.f; }
                else if (r >= 3 && c >= 3) { idx = 9; sg = (r == c) ? 1.f : 0.f; }
                else {
                    const int a = r < 3 ? r : c, l = (r < 3 ? c : r) - 3;   // [h]x(a, l)
                    if (a != l) { idx = 6 + (3 - a - l); sg = (l == (a + 2) % 3) ? 1.f : -1.f; }
                }
            }
            const int rr = r < c ? r : c, cc = r < c ? c : r;
            const int idx21 = (r < 6 && c < 6) ? (rr * (13 - rr)) / 2 + (cc - rr) : 21;   // 21 = the always-zero slot of a record
            w[12 + k] = (unsigned)c | ((unsigned)idx << 4) | ((unsigned)idx21 << 8);
            std::memcpy(&w[20 + k], &sg, 4);
        }
    }
    T.sched8 = upload<uint32_t>(s, tab, &ok);
    T.obj_inertial = nullptr; T.obj_geoms = nullptr; T.obj_geom_adr = nullptr; T.n_obj = 0;
    if (!m.obj_geoms.empty() && m.obj_inertial.size() == 13 * m.obj_mass.size()) {
        T.obj_inertial = upload<float>(s, m.obj_inertial, &ok); T.obj_geoms = upload<float>(s, m.obj_geoms, &ok);
        T.obj_geom_adr = upload<int>(s, m.obj_geom_adr, &ok); T.n_obj = (int)m.obj_mass.size();
    }
    // scalar parameters
    auto& P = s->P;
    const auto& o = m.opt;
    auto clampimp = [](double v) { return std::min(0.9999, std::max(0.0001, v)); };
    P.h = (float)o[OPT_TIMESTEP]; P.gx = (float)s->model->gravity_x; P.gy = (float)s->model->gravity_y; P.gz = (float)s->model->gravity_z;
    double tc = std::max(o[OPT_SOLREF_TC], 2 * o[OPT_TIMESTEP]), dr = o[OPT_SOLREF_DR], dmax = clampimp(o[OPT_IMP_DW]);
    P.K = (float)(1.0 / (dmax * dmax * tc * tc * dr * dr)); P.B = (float)(2.0 / (dmax * tc));
    P.imp_d0 = (float)clampimp(o[OPT_IMP_D0]); P.imp_dw = (float)dmax; P.imp_w = (float)o[OPT_IMP_W];
    P.imp_mid = (float)clampimp(o[OPT_IMP_MID]); P.imp_pow = (float)std::max(1.0, o[OPT_IMP_POW]);
    P.mu = (float)o[OPT_FRIC]; P.margin = (float)o[OPT_MARGIN];
    P.pm_max = s->model->planemesh_max; P.pm_tol = (float)s->model->planemesh_tol;
    // mj_solNewton's termination scale: mean inertia and dof count of the WHOLE reference scene (objects included)
    P.scale = (float)(1.0 / (o[OPT_MEANINERTIA] * (o.size() > OPT_NV_FULL ? o[OPT_NV_FULL] : NV)));
    P.rfc_scale = (float)o[OPT_RFC_SCALE]; P.rfc_lim = (float)o[OPT_RFC_LIM];
    double bn = o[OPT_BR_W] * o[OPT_BR_W] + o[OPT_BR_X] * o[OPT_BR_X] + o[OPT_BR_Y] * o[OPT_BR_Y] + o[OPT_BR_Z] * o[OPT_BR_Z];
    P.br_inv[0] = (float)(o[OPT_BR_W] / bn); P.br_inv[1] = (float)(-o[OPT_BR_X] / bn);
    P.br_inv[2] = (float)(-o[OPT_BR_Y] / bn); P.br_inv[3] = (float)(-o[OPT_BR_Z] / bn);
    P.tol = (float)s->model->solver_tol; P.max_iter = s->model->solver_iter;
    P.contact = s->model->contact; P.limits = s->model->limits; P.stale = s->model->stale; P.actuation = s->model->actuation;
    return ok;
}

// Job sizes of kp_step_queue_kernel for a control step of nsub substeps: `spj` substeps for the last job and `taper` more for each job
// before it -- the FIFO runs all envs' first jobs, then all second jobs ...: long jobs first keep the hand-overs few, short jobs
// last keep the end of the launch short -- what is left over becomes the first job (or joins it if shorter than spj); at most 16 jobs.
// Default (spj 4, taper 1): 15 = 6 + 5 + 4.
int job_schedule(int nsub, int spj, int taper, int* sizes) {
    int parts = 0, rem = nsub, size = spj;
    while (rem > 0 && parts < 16) {
        int take = parts == 15 ? rem : std::min(rem, size);
        if (rem - take > 0 && rem - take < spj) take = rem;
        sizes[parts++] = take; rem -= take;
        size += taper;
    }
    std::reverse(sizes, sizes + parts);
    return parts;
}

int launch_step(kp_sim* s, const float* action, int nsub, const uint8_t* mask, bool time_it) {
    kp::StepArgs A{};
    A.T = s->T; A.P = s->P; A.n_envs = s->n; A.n_substeps = nsub;
    A.qpos = s->qpos; A.qvel = s->qvel; A.qpos_d = s->qpos_d; A.qvel_d = s->qvel_d; A.warm = s->warm; A.warm2 = s->warm2;
    A.target_qpos = s->t_qpos; A.action = action; A.env_mask = mask;
    A.xpos = s->xpos; A.xquat = s->xquat; A.xipos = s->xipos; A.diag = s->diag; A.prof = s->prof;
    A.geoms = s->geoms; A.ngeom = s->ngeom; A.dbg_contacts = s->dbg_contacts;
    A.obj_slot = s->obj_slot; A.obj_qpos = s->obj_qpos; A.obj_qvel = s->obj_qvel; A.obj_warm = s->obj_warm; A.obj_warm2 = s->obj_warm2;
    A.order = nullptr; A.cost = s->cost;
    const bool obj = s->has_objects;
    if (obj && s->model->threads != 64) return fail("object contact needs threads_per_env = 64");
    A.warm_extrap = s->model->warm_extrap < 0.f ? (obj ? 0.75f : 0.f) : s->model->warm_extrap;      // the Newton solve's starting point (kp_step_kernel.hpp)
    size_t lds = obj ? sizeof(kp::EnvLdsObj) : sizeof(kp::EnvLds);
    hipEvent_t e0 = s->ev0, e1 = s->ev1;
    if (time_it) {      // a launch that is being captured into a hipGraph carries no timing events (they could not be read back)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) time_it = false;
    }
    if (time_it && s->ring_on && s->ring_used < 4096) {
        if ((int)s->ring.size() < 2 * (s->ring_used + 1)) {
            hipEvent_t a, b;
            HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
            s->ring.push_back(a); s->ring.push_back(b);
        }
        e0 = s->ring[2 * s->ring_used]; e1 = s->ring[2 * s->ring_used + 1];
        s->ring_used++;
    }
    if (time_it) HIP_OK(hipEventRecord(e0, s->stream));
    // lpt_order: 1 on, 0 off, -1 (default) on when the scene's objects are simulated: an env whose hulls press on objects stays the launch's
    // longest for many control steps, so starting it first shortens the launch (objects workload 7.21 -> 6.62 ms); floor-only costs are not
    // predictable from step to step (correlation 0.25 .. 0.55) and gain nothing
    if (nsub > 0 && (s->model->lpt_order > 0 || (s->model->lpt_order < 0 && s->has_objects))) {      // longest env first (inside the timed bracket): order the workgroups / the queue's first jobs by the cycles of the previous control step
        hipLaunchKernelGGL(kp::k_lpt_order, dim3(1), dim3(1024), 0, s->stream, s->n, s->cost, s->order);
        A.order = s->order;
    }
#define KP_LAUNCH(NT_) do { if (nsub > 0) hipLaunchKernelGGL((kp::kp_step_kernel<NT_, false>), dim3(s->n), dim3(NT_), lds, s->stream, A); \
                            else hipLaunchKernelGGL((kp::kp_forward_kernel<NT_, false>), dim3(s->n), dim3(NT_), lds, s->stream, A); } while (0)
    // more envs than resident wave slots: schedule the control step as jobs of substeps_per_job substeps pulled from a FIFO by one
    // resident wave per slot (kp_step_queue_kernel) instead of one workgroup per env, which ends on a long tail
    // Schedule defaults.  Full layout (objects; floor with lean_queue = 0): 6 + 5 + 4 substeps, FIFO, no issue priorities (rounds 2 - 5).  Lean layout: 3072 slots hold
    // three quarters of BASELINE's 4096 envs, so a quarter of the envs start one job late and the launch ends on them (sum of env cycles / slots 1.83 ms, launch
    // 2.44 ms).  Three measures take the launch to 2.19 ms (profiles/r06/lean_schedule_knobs*.log): equal jobs 5 + 5 + 5 (the late envs start earlier), a late
    // env is never queued again (queue_late), and every wave's issue priority follows its env's distance from the end of the control step (queue_prio = 3).
    // Options the caller sets (substeps_per_job / job_taper, queue_prio, queue_late) are obeyed.
    // resident waves: LDS is allocated in 1 280-byte granules, 128 per CU (tools/micro/lds_granule_probe.hip); the register budgets allow 8 waves per CU
    // (full layout, <= 256 VGPRs) and 12 (lean layout, 168 VGPRs)
    // floor scenes: the job queue runs on the lean layout (EnvLdsLean) with a register budget for three waves per SIMD; model option lean_queue = 0 keeps the
    // full layout (two waves per SIMD; A / B measurements).  lds_pad: allocate at least that many bytes per env (experiments: fewer envs per CU with the same binary)
    bool lean = !obj && s->model->lean_queue && s->model->threads == 64;
    bool capturing = false;
    { hipStreamCaptureStatus cap = hipStreamCaptureStatusNone; capturing = hipStreamIsCapturing(s->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone; }
    if (lean && nsub > 0 && s->ovf_host && !capturing) {
        for (int k = 0; k < 2; k++)
            if (s->ovf_pending[k] && hipEventQuery(s->ovf_ev[k]) == hipSuccess) {
                s->ovf_pending[k] = false;
                s->ovf_last = s->ovf_host[k];
                if ((size_t)s->ovf_host[k] * 64 > (size_t)s->n) { s->lean_off_until = s->launch_index + 64; s->lean_fallbacks++; }
            }
        if (s->model->lean_adaptive && s->launch_index < s->lean_off_until) lean = false;
    }
    if (nsub > 0) s->launch_index++;
    size_t lds_q = lean ? sizeof(kp::EnvLdsLean) : lds;
    if (s->model->lds_pad > 0) lds_q = std::max(lds_q, (size_t)s->model->lds_pad);
    const int per_cu = std::min(lean ? 12 : 8, 128 / (int)((lds_q + 1279) / 1280));
    const int slots = s->model->queue_slots > 0 ? s->model->queue_slots : s->wave_slots / 8 * per_cu;
    const bool lean_auto = lean && s->model->job_auto;
    const int spj = lean_auto ? 5 : s->model->substeps_per_job, taper = lean_auto ? 0 : s->model->job_taper;
    int sizes[16], parts = 0;
    if (spj > 0 && nsub > 0) {
        parts = job_schedule(nsub, spj, taper, sizes);
        if (const char* e = std::getenv("KP_JOB_SCHEDULE")) {      // experiments: explicit comma-separated job sizes
            int tmp[16], np = 0, sum = 0;
            for (const char* c = e; *c && np < 16;) { tmp[np] = std::atoi(c); sum += tmp[np++]; while (*c && *c != ',') c++; if (*c) c++; }
            if (sum == nsub) { parts = np; std::copy(tmp, tmp + np, sizes); }
        }
    }
    // the queue pays from the env count on that one-workgroup-per-env launches (full layout) cannot hold resident at once; the lean queue then runs with
    // every env resident up to its own slot count and in rounds beyond it
    const int resident_full = s->wave_slots / 8 * std::min(8, 128 / (int)((lds + 1279) / 1280));
    const bool queue = nsub > 0 && parts > 1 && s->model->threads == 64 && s->n > std::min(slots, resident_full) && s->n <= 0xFFFFFF && !s->prof;   // env ids take 24 bits of a queue entry
    A.jobq = s->jobq; A.jobctr = s->jobctr; A.ovfq = s->ovfq; A.lean_cap = s->model->lean_cap; A.warm3 = s->warm3; A.spd_next = s->spd_next; A.n_parts = queue ? parts : 1; A.queue_fence = s->model->queue_fence; A.queue_heavy = s->model->queue_heavy; A.queue_prio = s->model->queue_prio >= 0 ? s->model->queue_prio : (lean ? 3 : 0); A.queue_late = s->model->queue_late >= 0 ? s->model->queue_late : (lean ? 1 : 0); A.order_valid = A.order != nullptr;
    A.part_sub_lo = A.part_sub_hi = 0;
    for (int k = 0; queue && k < parts; k++) (k < 8 ? A.part_sub_lo : A.part_sub_hi) |= (unsigned long long)(sizes[k] & 255) << (8 * (k & 7));
    if (queue) {
        // the yardstick of queue_heavy is the previous launch's mean job time per substep: it only means something for the same kernel on the same
        // schedule, so a change of the substep count or of the kernel instantiation starts it from nothing (that launch keeps no env)
        if (nsub != s->q_nsub || (int)obj + 2 * (int)lean != s->q_obj) {
            HIP_OK(hipMemsetAsync(s->jobctr + 32, 0, 2 * sizeof(unsigned), s->stream));
            s->q_nsub = nsub; s->q_obj = (int)obj + 2 * (int)lean;
        }
        const unsigned total = (unsigned)s->n * (unsigned)parts;
        hipLaunchKernelGGL(kp::k_queue_init, dim3((total + 255) / 256), dim3(256), 0, s->stream, s->n, total, s->jobq, s->jobctr, A.order);   // inside the timed bracket
        A.order = nullptr;                                              // the queue kernel addresses envs by their queue entry
        if (obj) hipLaunchKernelGGL((kp::kp_step_queue_kernel<true>), dim3(slots), dim3(64), lds_q, s->stream, A);
        else if (!lean) hipLaunchKernelGGL((kp::kp_step_queue_kernel<false>), dim3(slots), dim3(64), lds_q, s->stream, A);
        else {
            hipLaunchKernelGGL((kp::kp_step_queue_kernel<false, true>), dim3(slots), dim3(64), lds_q, s->stream, A);
            // the jobs whose contacts did not fit the lean layout (normally none: every wave leaves at its first read)
            // grid: twice the overflow count last seen (the scenes that overflow keep doing so for a while), at least 128 workgroups -- an unexpected burst is
            // slow once and sized right from the next-but-one launch on; launching full residency every time costs 34 us per control step for nothing
            const int ovf_grid = std::min(std::min(s->n, resident_full), std::max(128, 2 * (int)std::min<unsigned>(s->ovf_last, 1u << 20)));
            hipLaunchKernelGGL(kp::kp_step_overflow_kernel, dim3(ovf_grid), dim3(64), lds, s->stream, A);
            if (s->ovf_host && !capturing) {
                const int k = (int)(s->launch_index & 1);
                if (!s->ovf_pending[k]) {
                    HIP_OK(hipMemcpyAsync(s->ovf_host + k, s->jobctr + 64, sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
                    HIP_OK(hipEventRecord(s->ovf_ev[k], s->stream));
                    s->ovf_pending[k] = true;
                }
            }
        }
    } else
    switch (s->model->threads) {
        case 64:
            if (obj && nsub > 0) hipLaunchKernelGGL((kp::kp_step_kernel<64, true>), dim3(s->n), dim3(64), lds, s->stream, A);
            else if (obj) hipLaunchKernelGGL((kp::kp_forward_kernel<64, true>), dim3(s->n), dim3(64), lds, s->stream, A);
            else KP_LAUNCH(64);
            break;
        case 128: KP_LAUNCH(128); break;
        case 256: KP_LAUNCH(256); break;
        default: return fail("threads_per_env must be 64, 128 or 256");
    }
#undef KP_LAUNCH
    HIP_OK(hipGetLastError());
    if (time_it) {
        HIP_OK(hipEventRecord(e1, s->stream));
        s->last0 = e0; s->last1 = e1; s->timed = true;
    }
    return 0;
}

}  // namespace

extern "C" {

const char* kp_last_error(void) { return g_err.c_str(); }
const char* kp_version(void) { return "kinpoly_sim 0.1 (gfx950)"; }

kp_model* kp_model_load(const char* path) {
    kp_model* m = new kp_model();
    if (!kp::load_kpm(path, m->h)) { fail("kp_model_load: " + m->h.error); delete m; return nullptr; }
    m->gravity_z = m->h.opt[kp::OPT_GZ]; m->gravity_x = m->h.opt[kp::OPT_GX]; m->gravity_y = m->h.opt[kp::OPT_GY];
    m->solver_tol = m->h.opt[kp::OPT_SOLVER_TOL];
    m->solver_iter = (int)m->h.opt[kp::OPT_SOLVER_ITER];      // 100: MuJoCo's default, which the reference never overrides
    if (m->solver_iter < 1) m->solver_iter = 100;
    m->planemesh_max = (int)m->h.planemesh[0]; m->planemesh_tol = m->h.planemesh[1];
    return m;
}
void kp_model_free(kp_model* m) { delete m; }

int kp_model_set_option(kp_model* m, const char* name, double v) {
    if (!m || !name) return fail("kp_model_set_option: null argument");
    std::string k(name);
    if (k == "contact") m->contact = v != 0;
    else if (k == "limits") m->limits = v != 0;
    else if (k == "gravity_z") m->gravity_z = v;
    else if (k == "gravity_x") m->gravity_x = v;
    else if (k == "gravity_y") m->gravity_y = v;
    else if (k == "actuation") m->actuation = v != 0;
    else if (k == "stale_kinematics") m->stale = v != 0;
    else if (k == "solver_iter") m->solver_iter = (int)v;
    else if (k == "solver_tol") m->solver_tol = v;
    else if (k == "dynamic_objects") m->dynamic_objects = v != 0;
    else if (k == "planemesh_max") { if (v < 1 || v > 8) return fail("planemesh_max must be 1 .. 8"); m->planemesh_max = (int)v; }
    else if (k == "planemesh_tol") { if (v < 0) return fail("planemesh_tol must be >= 0"); m->planemesh_tol = v; }
    else if (k == "lpt_order") m->lpt_order = v < 0 ? -1 : (v != 0);
    else if (k == "job_taper") { m->job_taper = std::max(0, std::min(8, (int)v)); m->job_auto = 0; }
    else if (k == "queue_fence") m->queue_fence = v != 0;
    else if (k == "queue_heavy") m->queue_heavy = std::max(0, (int)v);
    else if (k == "queue_late") m->queue_late = v < 0 ? -1 : (v != 0);
    else if (k == "queue_prio") m->queue_prio = std::max(-1, std::min(3, (int)v));
    else if (k == "warm_extrap") m->warm_extrap = (float)v;
    else if (k == "lean_queue") m->lean_queue = v != 0;
    else if (k == "lean_adaptive") m->lean_adaptive = v != 0;
    else if (k == "lean_max_contacts") { if (v < 0 || v > kp::EnvLdsLean::MAXCON) return fail("lean_max_contacts must be 0 .. " + std::to_string(kp::EnvLdsLean::MAXCON)); m->lean_cap = (int)v; }
    else if (k == "lds_pad") { if (v < 0 || v > 65536) return fail("lds_pad must be 0 .. 65536 bytes"); m->lds_pad = (int)v; }
    else if (k == "queue_slots") { if (v < 0) return fail("queue_slots must be >= 0 (0 = resident wave slots of the device)"); m->queue_slots = (int)v; }
    else if (k == "substeps_per_job") { if (v < 0 || v > 255) return fail("substeps_per_job must be 0 (whole control step per workgroup) .. 255"); m->substeps_per_job = (int)v; m->job_auto = 0; }
    else if (k == "threads_per_env") { if (v != 64 && v != 128 && v != 256) return fail("threads_per_env must be 64, 128 or 256"); m->threads = (int)v; }
    else return fail("kp_model_set_option: unknown option " + k);
    return 0;
}
double kp_model_get_option(const kp_model* m, const char* name) {
    std::string k(name ? name : "");
    if (!m) return NAN;
    if (k == "contact") return m->contact;
    if (k == "limits") return m->limits;
    if (k == "gravity_z") return m->gravity_z;
    if (k == "gravity_x") return m->gravity_x;
    if (k == "gravity_y") return m->gravity_y;
    if (k == "actuation") return m->actuation;
    if (k == "stale_kinematics") return m->stale;
    if (k == "solver_iter") return m->solver_iter;
    if (k == "solver_tol") return m->solver_tol;
    if (k == "dynamic_objects") return m->dynamic_objects;
    if (k == "planemesh_max") return m->planemesh_max;
    if (k == "planemesh_tol") return m->planemesh_tol;
    if (k == "lpt_order") return m->lpt_order;
    if (k == "substeps_per_job") return m->substeps_per_job;
    if (k == "queue_slots") return m->queue_slots;
    if (k == "lean_queue") return m->lean_queue;
    if (k == "lean_adaptive") return m->lean_adaptive;
    if (k == "lean_max_contacts") return m->lean_cap;
    if (k == "lds_pad") return m->lds_pad;
    if (k == "lds_bytes_per_env_lean") return (double)sizeof(kp::EnvLdsLean);
    if (k == "job_taper") return m->job_taper;
    if (k == "job_auto") return m->job_auto;       // 1: the job sizes are the layout's defaults (lean queue: 5 + 5 + 5; otherwise substeps_per_job / job_taper = 6 + 5 + 4)
    if (k == "queue_fence") return m->queue_fence;
    if (k == "queue_heavy") return m->queue_heavy;
    if (k == "queue_prio") return m->queue_prio;
    if (k == "queue_late") return m->queue_late;
    if (k == "warm_extrap") return m->warm_extrap;
    if (k == "threads_per_env") return m->threads;
    if (k == "timestep") return m->h.opt[kp::OPT_TIMESTEP];
    if (k == "lds_bytes_per_env") return (double)sizeof(kp::EnvLds);
    if (k == "lds_bytes_per_env_objects") return (double)sizeof(kp::EnvLdsObj);
    return NAN;
}

kp_sim* kp_sim_create(const kp_model* m, int n_envs, int device_id, void* stream) {
    if (!m || n_envs <= 0) { fail("kp_sim_create: bad arguments"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fail("kp_sim_create: no HIP device available (the simulator has no CPU fallback)"); return nullptr; }
    HIP_OK_NULL(hipSetDevice(device_id));
    kp_sim* s = new kp_sim();
    s->model = m; s->n = n_envs; s->device = device_id; s->stream = (hipStream_t)stream;
    bool ok = build_tables(s);
    size_t N = n_envs;
    s->qpos = dalloc(s, N * 76, &ok); s->qvel = dalloc(s, N * 75, &ok); s->qpos_d = dalloc(s, N * 76, &ok);
    s->qvel_d = dalloc(s, N * 75, &ok); s->warm = dalloc(s, N * 75, &ok); s->warm2 = dalloc(s, N * 75, &ok);
    s->xpos = dalloc(s, N * 72, &ok); s->xquat = dalloc(s, N * 96, &ok); s->xipos = dalloc(s, N * 72, &ok);
    s->t_qpos = dalloc(s, N * 76, &ok); s->t_wbpos = dalloc(s, N * 72, &ok); s->t_wbquat = dalloc(s, N * 96, &ok);
    s->t_bquat = dalloc(s, N * 96, &ok); s->t_com = dalloc(s, N * 72, &ok);
    s->scratch = dalloc(s, N * 96, &ok);
    s->prev_bquat = dalloc(s, N * 96, &ok); s->prev_hpos = dalloc(s, N * 7, &ok);
    s->diag = (int*)dalloc(s, N * 4, &ok);
    s->order = (int*)dalloc(s, N, &ok); s->cost = (unsigned*)dalloc(s, N, &ok);
    s->jobq_cap = (int)N * 16; s->jobq = (unsigned*)dalloc(s, (size_t)s->jobq_cap, &ok); s->jobctr = (unsigned*)dalloc(s, 128, &ok); s->ovfq = (unsigned*)dalloc(s, N, &ok); s->warm3 = dalloc(s, N * 75, &ok); s->spd_next = (float*)dalloc(s, N * 80, &ok);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) s->wave_slots = prop.multiProcessorCount * 8;
    }
    s->obj_qpos = dalloc(s, N * 35, &ok); s->geoms = dalloc(s, N * kp::D_MAXGEOM * 17, &ok); s->ngeom = (int*)dalloc(s, N, &ok);
    s->obj_qvel = dalloc(s, N * 30, &ok); s->obj_warm = dalloc(s, N * 6 * kp::D_MAXOBJ, &ok); s->obj_warm2 = dalloc(s, N * 6 * kp::D_MAXOBJ, &ok); s->obj_slot = (signed char*)dalloc(s, (N * kp::D_MAXOBJ + 3) / 4 + 1, &ok);
    if (!m->h.obj_geoms.empty()) {
        s->d_obj_geoms = upload<float>(s, m->h.obj_geoms, &ok); s->d_obj_mass = upload<float>(s, m->h.obj_mass, &ok);
        s->n_obj_geoms = (int)(m->h.obj_geoms.size() / 18); s->n_obj = (int)m->h.obj_mass.size();
    }
    if (const char* e = std::getenv("KP_PROFILE")) if (e[0] == '1') s->prof = (unsigned long long*)dalloc(s, N * 16, &ok);
    if (hipHostMalloc((void**)&s->ovf_host, 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) s->ovf_host = nullptr;
    else { s->ovf_host[0] = s->ovf_host[1] = 0u; if (hipEventCreateWithFlags(&s->ovf_ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->ovf_ev[1], hipEventDisableTiming) != hipSuccess) { hipHostFree(s->ovf_host); s->ovf_host = nullptr; } }
    if (!ok || hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) {
        fail("kp_sim_create: device allocation / table build failed");
        kp_sim_destroy(s);
        return nullptr;
    }
    return s;
}

void kp_sim_destroy(kp_sim* s) {
    if (!s) return;
    hipSetDevice(s->device);
    hipStreamSynchronize(s->stream);
    for (void* p : s->allocs) hipFree(p);
    for (hipEvent_t e : s->ring) hipEventDestroy(e);
    if (s->ovf_host) { hipHostFree(s->ovf_host); for (hipEvent_t e : s->ovf_ev) if (e) hipEventDestroy(e); }
    if (s->ev0) hipEventDestroy(s->ev0);
    if (s->ev1) hipEventDestroy(s->ev1);
    delete s;
}
int kp_sim_n_envs(const kp_sim* s) { return s ? s->n : -1; }

int kp_sim_set_stream(kp_sim* s, void* stream) {
    if (!s) return fail("kp_sim_set_stream: null argument");
    s->stream = (hipStream_t)stream;
    return 0;
}

const uint32_t* kp_sim_status_device(kp_sim* s) { return s ? s->jobctr : nullptr; }

int kp_sim_contacts(kp_sim* s, float* out_host) {
    if (!s) return fail("kp_sim_contacts: null argument");
    HIP_OK(hipSetDevice(s->device));
    const size_t count = (size_t)s->n * (1 + kp::D_MAXCON * 9);
    if (!s->dbg_contacts) {           // first call arms the recording: later kp_sim_step_ctrl launches store their last contact set
        bool ok = true;
        s->dbg_contacts = dalloc(s, count, &ok);
        if (!ok) return fail("kp_sim_contacts: allocation failed");
        if (out_host) std::memset(out_host, 0, sizeof(float) * count);
        return 0;
    }
    if (!out_host) return 0;
    HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipMemcpy(out_host, s->dbg_contacts, sizeof(float) * count, hipMemcpyDeviceToHost));
    return 0;
}

int kp_sim_mass_matrix(kp_sim* s, float* M, float* bias) {
    if (!s || (!M && !bias)) return fail("kp_sim_mass_matrix: null argument");
    HIP_OK(hipSetDevice(s->device));
    kp::StepArgs A{};
    A.T = s->T; A.P = s->P; A.n_envs = s->n; A.n_substeps = 0;
    A.qpos_d = s->qpos_d; A.qvel_d = s->qvel_d;
    hipLaunchKernelGGL(kp::kp_mass_kernel, dim3(s->n), dim3(64), sizeof(kp::EnvLds), s->stream, A, M, bias);
    HIP_OK(hipGetLastError());
    return 0;
}

// masked row copy kernel (set_state writes only the masked envs)
__global__ void k_copy_rows(int n, int dim, const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dst2,
                            const uint8_t* __restrict__ mask) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * dim) return;
    int e = (int)(t / dim);
    if (mask && !mask[e]) return;
    float v = src ? src[t] : 0.f;
    dst[t] = v;
    if (dst2) dst2[t] = v;
}

int kp_sim_set_state(kp_sim* s, const float* qpos, const float* qvel, const uint8_t* mask) {
    if (!s || !qpos || !qvel) return fail("kp_sim_set_state: null argument");
    HIP_OK(hipSetDevice(s->device));
    int n = s->n;
    hipLaunchKernelGGL(k_copy_rows, dim3((n * 76 + 255) / 256), dim3(256), 0, s->stream, n, 76, qpos, s->qpos, s->qpos_d, mask);
    hipLaunchKernelGGL(k_copy_rows, dim3((n * 75 + 255) / 256), dim3(256), 0, s->stream, n, 75, qvel, s->qvel, s->qvel_d, mask);
    hipLaunchKernelGGL(k_copy_rows, dim3((n * 75 + 255) / 256), dim3(256), 0, s->stream, n, 75, (const float*)nullptr, s->warm, (float*)nullptr, mask);
    HIP_OK(hipGetLastError());
    return launch_step(s, nullptr, 0, mask, false);  // sim.forward(): derived quantities at the new state
}

int kp_sim_set_target(kp_sim* s, const float* tq, const uint8_t* mask) {
    if (!s || !tq) return fail("kp_sim_set_target: null argument");
    HIP_OK(hipSetDevice(s->device));
    kp::TargetBufs B{s->t_qpos, s->t_wbpos, s->t_wbquat, s->t_bquat, s->t_com};
    hipLaunchKernelGGL(kp::k_target_fk, dim3((s->n + 3) / 4), dim3(256), 0, s->stream, s->n, tq, mask, B, s->T.body_pos, s->T.body_ipos, s->T.body_parent, s->T.body_depth);
    HIP_OK(hipGetLastError());
    return 0;
}

// objects that are not parked (convert_obj_qpos parks the inactive ones 100+ m away): dynamic mode -> they become the env's
// free bodies (slots in object order, velocities zeroed as reset_model does); static mode -> their world-frame geoms are frozen
// row (optional): env e takes row row[e] of obj_qpos_in (a table of context rows, kp_sim_reset_rows) instead of row e; obj7 / one_hot (optional):
// obj7[e] <- get_obj_qpos(action_one_hot) of the fresh block, i.e. the 7 floats at the action's slot, untouched when the row has no action
__global__ void k_set_objects(int n, const float* __restrict__ obj_qpos_all, const uint8_t* __restrict__ mask, float* __restrict__ obj_qpos,
                              float* __restrict__ geoms, int* __restrict__ ngeom, const float* __restrict__ og, const float* __restrict__ omass,
                              int n_og, int n_obj, int dynamic, signed char* __restrict__ slot, float* __restrict__ obj_qvel, float* __restrict__ obj_warm,
                              const int* __restrict__ row, const float* __restrict__ one_hot, float* __restrict__ obj7) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (mask && !mask[e]) return;
    int ng = 0, ns = 0;
    const size_t r = row ? (size_t)row[e] : (size_t)e;
    const float* src = obj_qpos_all + r * 35;
    if (obj7) {
        const int st = kp::obj_action_start(one_hot ? one_hot + r * 4 : nullptr);
        if (st >= 0) for (int i = 0; i < 7; i++) obj7[(size_t)e * 7 + i] = obj_qpos_all[r * 35 + st + i];
        else { for (int i = 0; i < 7; i++) obj7[(size_t)e * 7 + i] = 0.f; obj7[(size_t)e * 7 + 3] = 1.f; }
    }
    for (int i = 0; i < 35; i++) obj_qpos[(size_t)e * 35 + i] = src[i];
    for (int i = 0; i < 30; i++) obj_qvel[(size_t)e * 30 + i] = 0.f;
    for (int i = 0; i < 6 * kp::D_MAXOBJ; i++) obj_warm[(size_t)e * 6 * kp::D_MAXOBJ + i] = 0.f;
    for (int k = 0; k < kp::D_MAXOBJ; k++) slot[(size_t)e * kp::D_MAXOBJ + k] = -1;
    if (dynamic) {
        for (int oi = 0; oi < n_obj && oi < 5 && ns < kp::D_MAXOBJ; oi++) {
            const float* pose = src + 7 * oi;
            if (sqrtf(pose[0] * pose[0] + pose[1] * pose[1] + pose[2] * pose[2]) > 50.0f) continue;
            slot[(size_t)e * kp::D_MAXOBJ + ns++] = (signed char)oi;
        }
        ngeom[e] = 0;
        return;
    }
    for (int gi = 0; gi < n_og && ng < kp::D_MAXGEOM; gi++) {
        const float* g = og + 18 * gi;
        const int oi = (int)g[0];
        if (oi >= n_obj || oi >= 5) continue;
        const float* pose = src + 7 * oi;
        if (sqrtf(pose[0] * pose[0] + pose[1] * pose[1] + pose[2] * pose[2]) > 50.0f) continue;
        kp::Q4 q = kp::qnormalize(kp::Q4{pose[3], pose[4], pose[5], pose[6]});
        float R[9], Rg[9];
        kp::q2mat(q, R);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rg[3 * i + j] = R[3 * i] * g[8 + j] + R[3 * i + 1] * g[11 + j] + R[3 * i + 2] * g[14 + j];
        kp::V3 p = kp::mulmat(R, kp::ld3(g + 5));
        float* o = geoms + ((size_t)e * kp::D_MAXGEOM + ng) * 17;
        o[0] = g[1]; o[1] = g[2]; o[2] = g[3]; o[3] = g[4];
        o[4] = pose[0] + p.x; o[5] = pose[1] + p.y; o[6] = pose[2] + p.z;
        for (int k = 0; k < 9; k++) o[7 + k] = Rg[k];
        o[16] = 1.0f / omass[oi];
        ng++;
    }
    ngeom[e] = ng;
}

int kp_sim_set_objects(kp_sim* s, const float* obj_qpos, const uint8_t* mask) {
    if (!s || !obj_qpos) return fail("kp_sim_set_objects: null argument");
    if (!s->d_obj_geoms) return fail("kp_sim_set_objects: the model blob has no object geoms");
    const int dynamic = s->model->dynamic_objects && s->T.obj_inertial != nullptr;
    HIP_OK(hipSetDevice(s->device));
    hipLaunchKernelGGL(k_set_objects, dim3((s->n + 63) / 64), dim3(64), 0, s->stream, s->n, obj_qpos, mask, s->obj_qpos, s->geoms, s->ngeom,
                       s->d_obj_geoms, s->d_obj_mass, s->n_obj_geoms, s->n_obj, dynamic, s->obj_slot, s->obj_qvel, s->obj_warm,
                       (const int*)nullptr, (const float*)nullptr, (float*)nullptr);
    HIP_OK(hipGetLastError());
    s->has_objects = true;
    return 0;
}

int kp_sim_fk(kp_sim* s, int n_rows, const float* qpos, float* qpos_out, float* wbpos, float* wbquat, float* bquat, float* com) {
    if (!s || !qpos || n_rows <= 0) return fail("kp_sim_fk: bad arguments");
    HIP_OK(hipSetDevice(s->device));
    kp::TargetBufs B{qpos_out, wbpos, wbquat, bquat, com};
    hipLaunchKernelGGL(kp::k_target_fk, dim3((n_rows + 3) / 4), dim3(256), 0, s->stream, n_rows, qpos, (const uint8_t*)nullptr, B,
                       s->T.body_pos, s->T.body_ipos, s->T.body_parent, s->T.body_depth);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_sim_fk_backward(kp_sim* s, int n_rows, const float* qpos, const float* wbpos, const float* wbquat, const float* grad_wbpos, float* grad_qpos) {
    if (!s || !qpos || !wbpos || !wbquat || !grad_wbpos || !grad_qpos || n_rows <= 0) return fail("kp_sim_fk_backward: bad arguments");
    HIP_OK(hipSetDevice(s->device));
    hipLaunchKernelGGL(kp::k_fk_wbpos_grad, dim3((n_rows + 3) / 4), dim3(256), 0, s->stream, n_rows, qpos, wbpos, wbquat, grad_wbpos, grad_qpos,
                       s->T.body_parent, s->T.body_subtree);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_sim_step_ctrl(kp_sim* s, const float* action, int nsub, const uint8_t* mask) {
    if (!s || !action || nsub <= 0) return fail("kp_sim_step_ctrl: bad arguments");
    HIP_OK(hipSetDevice(s->device));
    return launch_step(s, action, nsub, mask, true);
}

int kp_sim_step_head(kp_sim* s, const float* act) {
    if (!s || !act) return fail("kp_sim_step_head: null argument");
    HIP_OK(hipSetDevice(s->device));
    kp::TargetBufs B{s->t_qpos, s->t_wbpos, s->t_wbquat, s->t_bquat, s->t_com};
    kp::KinStep K{act, s->xpos, s->xquat, s->prev_bquat, s->prev_hpos, 1.0f / 30.0f};
    hipLaunchKernelGGL(kp::k_target_fk, dim3((s->n + 3) / 4), dim3(256), 0, s->stream, s->n, s->qpos, (const uint8_t*)nullptr, B, s->T.body_pos, s->T.body_ipos,
                       s->T.body_parent, s->T.body_depth, K);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_sim_step_kin(kp_sim* s, const float* act, float* next_qpos) {
    if (!s || !act || !next_qpos) return fail("kp_sim_step_kin: null argument");
    HIP_OK(hipSetDevice(s->device));
    hipLaunchKernelGGL(kp::k_step_kin, dim3((s->n + 63) / 64), dim3(64), 0, s->stream, s->n, s->qpos, act, next_qpos, 1.0f / 30.0f);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_sim_obs_cc(kp_sim* s, float* out, const float* zf_mean, const float* zf_std, float clip) {
    if (!s || !out) return fail("kp_sim_obs_cc: null argument");
    if ((zf_mean == nullptr) != (zf_std == nullptr)) return fail("kp_sim_obs_cc: pass both zf_mean and zf_std or neither");
    HIP_OK(hipSetDevice(s->device));
    kp::ObsCcArgs A;
    A.n = s->n; A.qpos = s->qpos; A.qvel = s->qvel; A.xpos = s->xpos; A.xquat = s->xquat; A.xipos = s->xipos;
    A.t_qpos = s->t_qpos; A.t_wbpos = s->t_wbpos; A.t_wbquat = s->t_wbquat; A.t_com = s->t_com;
    for (int k = 0; k < 4; k++) A.br_inv[k] = s->P.br_inv[k];
    A.zf_mean = zf_mean; A.zf_std = zf_std; A.clip = clip; A.out = out;
    hipLaunchKernelGGL(kp::k_obs_cc, dim3(s->n), dim3(64), 0, s->stream, A);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_field_dim(int f) {
    switch (f) {
        case KP_QPOS: case KP_TARGET_QPOS: case KP_QPOS_D: return 76;
        case KP_QVEL: case KP_QVEL_D: return 75;
        case KP_XPOS: case KP_XIPOS: case KP_TARGET_WBPOS: case KP_TARGET_COM: return 72;
        case KP_XQUAT: case KP_BQUAT: case KP_TARGET_WBQUAT: case KP_TARGET_BQUAT: case KP_PREV_BQUAT: return 96;
        case KP_HEAD: case KP_PREV_HPOS: return 7;
        case KP_OBJ_QPOS: return 35;
        case KP_OBJ_QVEL: return 30;
        case KP_M: return 75 * 75;
        case KP_BIAS: return 75;
        default: return -1;
    }
}

__global__ void k_head(int n, const float* __restrict__ xpos, const float* __restrict__ xquat, float* __restrict__ out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int hb = 13;  // Head (model._body_name2id['Head'] - 1)
    for (int k = 0; k < 3; k++) out[(size_t)e * 7 + k] = xpos[(size_t)e * 72 + 3 * hb + k];
    for (int k = 0; k < 4; k++) out[(size_t)e * 7 + 3 + k] = xquat[(size_t)e * 96 + 4 * hb + k];
}

const float* kp_sim_field_device(kp_sim* s, int field) {
    if (!s) return nullptr;
    switch (field) {
        case KP_QPOS: return s->qpos;
        case KP_QVEL: return s->qvel;
        case KP_XPOS: return s->xpos;
        case KP_XQUAT: return s->xquat;
        case KP_XIPOS: return s->xipos;
        case KP_TARGET_QPOS: return s->t_qpos;
        case KP_QPOS_D: return s->qpos_d;
        case KP_QVEL_D: return s->qvel_d;
        case KP_OBJ_QPOS: return s->obj_qpos;
        case KP_OBJ_QVEL: return s->obj_qvel;
        default: return nullptr;
    }
}

int kp_sim_get(kp_sim* s, int field, float* out) {
    if (!s || !out) return fail("kp_sim_get: null argument");
    HIP_OK(hipSetDevice(s->device));
    const float* src = nullptr;
    switch (field) {
        case KP_QPOS: src = s->qpos; break;
        case KP_QVEL: src = s->qvel; break;
        case KP_XPOS: src = s->xpos; break;
        case KP_XQUAT: src = s->xquat; break;
        case KP_XIPOS: src = s->xipos; break;
        case KP_TARGET_QPOS: src = s->t_qpos; break;
        case KP_TARGET_WBPOS: src = s->t_wbpos; break;
        case KP_TARGET_WBQUAT: src = s->t_wbquat; break;
        case KP_TARGET_BQUAT: src = s->t_bquat; break;
        case KP_TARGET_COM: src = s->t_com; break;
        case KP_QPOS_D: src = s->qpos_d; break;
        case KP_QVEL_D: src = s->qvel_d; break;
        case KP_PREV_BQUAT: src = s->prev_bquat; break;
        case KP_PREV_HPOS: src = s->prev_hpos; break;
        case KP_OBJ_QPOS: src = s->obj_qpos; break;
        case KP_OBJ_QVEL: src = s->obj_qvel; break;
        case KP_BQUAT:
            hipLaunchKernelGGL(kp::k_bquat, dim3((s->n * 24 + 255) / 256), dim3(256), 0, s->stream, s->n, s->qpos, out);
            HIP_OK(hipGetLastError());
            return 0;
        case KP_M: return kp_sim_mass_matrix(s, out, nullptr);
        case KP_BIAS: return kp_sim_mass_matrix(s, nullptr, out);
        case KP_HEAD:
            hipLaunchKernelGGL(k_head, dim3((s->n + 255) / 256), dim3(256), 0, s->stream, s->n, s->xpos, s->xquat, out);
            HIP_OK(hipGetLastError());
            return 0;
        default: return fail("kp_sim_get: unknown field");
    }
    HIP_OK(hipMemcpyAsync(out, src, sizeof(float) * (size_t)s->n * kp_field_dim(field), hipMemcpyDeviceToDevice, s->stream));
    return 0;
}

static kp::CtxDev to_dev(const kp_ctx* c) {
    kp::CtxDev d;
    d.T = c->T; d.head_pose = c->head_pose; d.head_vels = c->head_vels; d.obj_rel = c->obj_head_relative_poses;
    d.action_one_hot = c->action_one_hot; d.gt_bquat = c->gt_bquat; d.gt_wbpos = c->gt_wbpos; d.obj_qpos = c->obj_qpos; d.cur_t = c->cur_t; d.row = c->row;
    return d;
}

int kp_sim_step_begin(kp_sim* s) {
    if (!s) return fail("kp_sim_step_begin: null argument");
    HIP_OK(hipSetDevice(s->device));
    hipLaunchKernelGGL(kp::k_snapshot, dim3((s->n * 24 + 255) / 256), dim3(256), 0, s->stream, s->n, s->qpos, s->xpos, s->xquat, s->prev_bquat, s->prev_hpos);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_sim_obs_ar(kp_sim* s, const kp_ctx* c, float* out) {
    if (!s || !c || !out || !c->head_pose || !c->head_vels || !c->obj_head_relative_poses || !c->action_one_hot || !c->cur_t || c->T < 1)
        return fail("kp_sim_obs_ar: bad arguments");
    HIP_OK(hipSetDevice(s->device));
    hipLaunchKernelGGL(kp::k_obs_ar, dim3((s->n + 63) / 64), dim3(64), 0, s->stream, s->n, to_dev(c), s->qpos, s->xpos, s->xquat, out);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_sim_term_reward(kp_sim* s, const kp_ctx* c, const kp_reward_cfg* w, float* reward, float* info, uint8_t* failp, float* diffs) {
    if (!s || !c || !w || !reward || !info || !failp || !diffs || !c->head_pose || !c->gt_bquat || !c->gt_wbpos || !c->cur_t || c->T < 2)
        return fail("kp_sim_term_reward: bad arguments");
    HIP_OK(hipSetDevice(s->device));
    kp::RewardW W{w->w_hp, w->w_hq, w->w_p, w->w_jp, w->w_act_p, w->w_act_v, w->k_hp, w->k_hq, w->k_p, w->k_jp, w->k_act_p, w->k_act_v,
                  w->dt, w->body_diff_thresh, w->body_diff_gt_thresh, w->use_gt_term};
    hipLaunchKernelGGL(kp::k_term_reward<false>, dim3((s->n + 7) / 8), dim3(256), 0, s->stream, s->n, to_dev(c), W, s->qpos, s->xpos, s->xquat,
                       s->t_wbpos, s->t_bquat, s->prev_bquat, s->prev_hpos, s->diffw, reward, info, failp, diffs, kp::PostStep{});
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_sim_post_step(kp_sim* s, const kp_ctx* c, const kp_reward_cfg* w, int32_t* cur_t, const int32_t* row_len, int env_episode_len,
                     float* reward, float* info, uint8_t* failp, float* diffs, uint8_t* done, uint8_t* end, float* percent, int32_t* done_count, float* obj7) {
    if (!s || !c || !w || !cur_t || !row_len || !reward || !info || !failp || !diffs || !done || !end || !percent || !c->head_pose || !c->gt_bquat || !c->gt_wbpos || c->T < 2)
        return fail("kp_sim_post_step: bad arguments");
    if (c->cur_t != cur_t) return fail("kp_sim_post_step: cur_t must be the buffer the context reads (kp_ctx.cur_t)");
    HIP_OK(hipSetDevice(s->device));
    kp::RewardW W{w->w_hp, w->w_hq, w->w_p, w->w_jp, w->w_act_p, w->w_act_v, w->k_hp, w->k_hq, w->k_p, w->k_jp, w->k_act_p, w->k_act_v,
                  w->dt, w->body_diff_thresh, w->body_diff_gt_thresh, w->use_gt_term};
    kp::PostStep PS{cur_t, row_len, env_episode_len, done, end, percent, done_count, obj7, obj7 ? s->obj_qpos : nullptr};
    hipLaunchKernelGGL(kp::k_term_reward<true>, dim3((s->n + 7) / 8), dim3(256), 0, s->stream, s->n, to_dev(c), W, s->qpos, s->xpos, s->xquat,
                       s->t_wbpos, s->t_bquat, s->prev_bquat, s->prev_hpos, s->diffw, reward, info, failp, diffs, PS);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_sim_reset_rows(kp_sim* s, const float* init_qpos, const float* init_qvel, const int32_t* row, const uint8_t* mask, int32_t* cur_t, int set_target,
                      float* aux_rows, int aux_cols, const float* row_obj_qpos, const float* row_action_one_hot, float* obj7) {
    if (!s || !init_qpos || !init_qvel || (aux_rows && aux_cols <= 0)) return fail("kp_sim_reset_rows: null argument");
    if (obj7 && !row_obj_qpos) return fail("kp_sim_reset_rows: obj7 needs row_obj_qpos");
    HIP_OK(hipSetDevice(s->device));
    if (row_obj_qpos) {       // the object block of reset_model (humanoid_ar_v1.py:377-382) from the env's context row, before sim.forward()
        if (!s->d_obj_geoms) return fail("kp_sim_reset_rows: the model blob has no object geoms");
        const int dynamic = s->model->dynamic_objects && s->T.obj_inertial != nullptr;
        hipLaunchKernelGGL(k_set_objects, dim3((s->n + 63) / 64), dim3(64), 0, s->stream, s->n, row_obj_qpos, mask, s->obj_qpos, s->geoms, s->ngeom,
                           s->d_obj_geoms, s->d_obj_mass, s->n_obj_geoms, s->n_obj, dynamic, s->obj_slot, s->obj_qvel, s->obj_warm, row, row_action_one_hot, obj7);
        HIP_OK(hipGetLastError());
        s->has_objects = true;
    }
    hipLaunchKernelGGL(kp::k_reset_rows, dim3(s->n), dim3(128), 0, s->stream, s->n, init_qpos, init_qvel, row, mask, cur_t, s->qpos, s->qvel, s->qpos_d, s->qvel_d, s->warm,
                       aux_rows, aux_rows ? aux_cols : 0);
    HIP_OK(hipGetLastError());
    if (int rc = launch_step(s, nullptr, 0, mask, false)) return rc;      // sim.forward(): derived quantities at the new state
    if (set_target) {                                                     // target = smpl_humanoid.qpos_fk(init_qpos) (humanoid_ar_v1.py:384-386): the state just written
        kp::TargetBufs B{s->t_qpos, s->t_wbpos, s->t_wbquat, s->t_bquat, s->t_com};
        hipLaunchKernelGGL(kp::k_target_fk, dim3((s->n + 3) / 4), dim3(256), 0, s->stream, s->n, s->qpos, mask, B, s->T.body_pos, s->T.body_ipos, s->T.body_parent, s->T.body_depth);
        HIP_OK(hipGetLastError());
    }
    return 0;
}

int kp_gae_bootstrap(int n, int T, const float* rewards, const float* masks, const float* values, const float* last_values, float gamma, float tau,
                     float* adv, float* ret, void* stream) {
    if (n <= 0 || T <= 0 || !rewards || !masks || !values || !adv || !ret) return fail("kp_gae: bad arguments");
    hipLaunchKernelGGL(kp::k_gae, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, T, rewards, masks, values, last_values, gamma, tau, adv, ret);
    HIP_OK(hipGetLastError());
    return 0;
}
int kp_gae(int n, int T, const float* rewards, const float* masks, const float* values, float gamma, float tau, float* adv, float* ret, void* stream) {
    return kp_gae_bootstrap(n, T, rewards, masks, values, nullptr, gamma, tau, adv, ret, stream);
}

int kp_pool_advance(int n, int n_slots, const uint8_t* done, int32_t* head, int32_t* ahead, int32_t* row, void* stream) {
    if (n <= 0 || n_slots <= 0 || !done || !head || !ahead || !row) return fail("kp_pool_advance: bad arguments");
    hipLaunchKernelGGL(kp::k_pool_advance, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, n_slots, done, head, ahead, row);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_rollout_record_pre(const kp_record_pre* r, void* stream) {
    if (!r || r->n <= 0 || r->T <= 0 || r->t < 0 || r->t >= r->T) return fail("kp_rollout_record_pre: bad arguments");
    if ((r->states && !r->obs) || (r->episode_start && !r->fresh) || (r->curr_qpos && !r->qpos) || (r->meta && !r->row_meta) ||
        (r->gt_target_qpos && (!r->ctx_qpos || !r->cur_t || !r->row_len || r->ctx_T <= 0)))
        return fail("kp_rollout_record_pre: a destination without its source");
    kp::RecordPre R{r->n, r->T, r->t, r->ctx_T, r->obs, r->fresh, r->qpos, r->ctx_qpos, r->row, r->cur_t, r->row_len, r->row_meta,
                    r->states, r->episode_start, r->curr_qpos, r->gt_target_qpos, r->meta};
    hipLaunchKernelGGL(kp::k_record_pre, dim3(r->n), dim3(128), 0, (hipStream_t)stream, R);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_rollout_record_post(const kp_record_post* r, void* stream) {
    if (!r || r->n <= 0 || r->T <= 0 || r->t < 0 || r->t >= r->T) return fail("kp_rollout_record_post: bad arguments");
    if ((r->actions && !r->action) || (r->rewards && !r->reward) || (r->fails && !r->fail) || (r->dones && !r->done) || (r->percents && !r->percent) ||
        (r->c_infos && !r->c_info) || (r->next_states && !r->obs) || (r->res_qpos && !r->qpos) || (r->cc_actions && !r->cc_action) ||
        (r->cc_states && !r->cc_state) || (r->v_metas && !r->meta))
        return fail("kp_rollout_record_post: a destination without its source");
    kp::RecordPost R{r->n, r->T, r->t, r->fr_num, r->action, r->reward, r->fail, r->done, r->percent, r->c_info, r->obs, r->qpos, r->cc_action, r->cc_state, r->meta,
                     r->actions, r->rewards, r->fails, r->dones, r->percents, r->c_infos, r->next_states, r->res_qpos, r->cc_actions, r->cc_states, r->v_metas};
    hipLaunchKernelGGL(kp::k_record_post, dim3(r->n), dim3(128), 0, (hipStream_t)stream, R);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_mcp_compose(int n, int K, int A, const float* logits, const float* prim, const float* noise, int noise_stride, const float* stdv, float* out, void* stream) {
    if (n <= 0 || K <= 0 || K > 64 || A <= 0 || !logits || !prim || !out || (noise && !stdv)) return fail("kp_mcp_compose: bad arguments");
    const size_t tot = (size_t)n * A;
    hipLaunchKernelGGL(kp::k_mcp_compose, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, K, A, logits, prim, noise, noise_stride, stdv, out);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_kin_advance(int n, const float* qpos, const float* kin_action, float dt, float* next_qpos, float* qvel_fd, void* stream) {
    if (n <= 0 || !qpos || !kin_action || !next_qpos || !qvel_fd || !(dt > 0.f)) return fail("kp_kin_advance: bad arguments");
    hipLaunchKernelGGL(kp::k_kin_advance, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, n, qpos, kin_action, dt, next_qpos, qvel_fd);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_mcp_tail(int n, int K, int J, int A, const float* h2, const float* b2, const float* w3, int ldw, const float* b3, const float* logits, const float* noise,
                int noise_stride, const float* stdv, float* out, void* stream) {
    if (n <= 0 || K <= 0 || K > 16 || J <= 0 || J % kp::MCP_CHUNK || A <= 0 || A > 80 || ldw < A || !h2 || !b2 || !w3 || !b3 || !logits || !out || (noise && !stdv))
        return fail("kp_mcp_tail: bad arguments (K <= 16 primitives, hidden width a multiple of 64, A <= 80 actions, ldw >= A)");
    const dim3 grid((unsigned)((n + 15) / 16)), block(64 * kp::MCP_WAVES);
    hipStream_t st = (hipStream_t)stream;
#define KP_TAIL(NT, V) hipLaunchKernelGGL((kp::k_mcp_tail<NT, V>), grid, block, 0, st, n, K, J, A, h2, b2, w3, ldw, b3, logits, noise, noise_stride, stdv, out)
    if (A <= 16) KP_TAIL(1, false);
    else if (A <= 48) KP_TAIL(3, false);
    else if (ldw >= 80 && ldw % 4 == 0 && ((uintptr_t)w3 & 15) == 0) KP_TAIL(5, true);      // rows padded to 80 columns: 16-byte operand loads
    else KP_TAIL(5, false);
#undef KP_TAIL
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_gru_cell_step(int n, int H, int D, const float* gi, const float* gh, const float* b_ih, const float* b_hh, const float* h_in, const float* state,
                     float* h_out, float* xcat, void* stream) {
    if (n <= 0 || H <= 0 || !gi || !gh || !b_ih || !b_hh || !h_in || !h_out || (xcat && (!state || D <= 0 || D > H)))
        return fail("kp_gru_cell_step: bad arguments (the [state | h] row needs state and 0 < D <= H)");
    const size_t tot = (size_t)n * H;
    hipLaunchKernelGGL(kp::k_gru_cell_step, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, H, D, gi, gh, b_ih, b_hh, h_in, state, h_out, xcat);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_gru_gates_forward(int n, int H, const float* gi, const float* gh, const float* hm_prev, const float* next_keep, float* h_out, float* hm_next, void* stream) {
    if (n <= 0 || H <= 0 || !gi || !gh || !hm_prev || !h_out) return fail("kp_gru_gates_forward: bad arguments");
    const size_t tot = (size_t)n * H;
    hipLaunchKernelGGL(kp::k_gru_gates_fwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, H, gi, gh, hm_prev, next_keep, h_out, hm_next);
    HIP_OK(hipGetLastError());
    return 0;
}
int kp_gru_gates_backward(int n, int H, const float* gi, const float* gh, const float* hm_prev, const float* dh_out, const float* carry, const float* carry_keep,
                          float* dgi, float* dgh, float* dhz, void* stream) {
    if (n <= 0 || H <= 0 || !gi || !gh || !hm_prev || !dgi || !dgh || !dhz) return fail("kp_gru_gates_backward: bad arguments");
    const size_t tot = (size_t)n * H;
    hipLaunchKernelGGL(kp::k_gru_gates_bwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, H, gi, gh, hm_prev, dh_out, carry, carry_keep, dgi, dgh, dhz);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_sim_set_full_state(kp_sim* s, const float* qpos, const float* qvel, const float* qpos_d, const float* qvel_d, const uint8_t* mask) {
    if (!s || !qpos || !qvel || !qpos_d || !qvel_d) return fail("kp_sim_set_full_state: null argument");
    HIP_OK(hipSetDevice(s->device));
    int n = s->n;
    hipLaunchKernelGGL(k_copy_rows, dim3((n * 76 + 255) / 256), dim3(256), 0, s->stream, n, 76, qpos, s->qpos, (float*)nullptr, mask);
    hipLaunchKernelGGL(k_copy_rows, dim3((n * 75 + 255) / 256), dim3(256), 0, s->stream, n, 75, qvel, s->qvel, (float*)nullptr, mask);
    hipLaunchKernelGGL(k_copy_rows, dim3((n * 76 + 255) / 256), dim3(256), 0, s->stream, n, 76, qpos_d, s->qpos_d, (float*)nullptr, mask);
    hipLaunchKernelGGL(k_copy_rows, dim3((n * 75 + 255) / 256), dim3(256), 0, s->stream, n, 75, qvel_d, s->qvel_d, (float*)nullptr, mask);
    HIP_OK(hipGetLastError());
    return launch_step(s, nullptr, 0, mask, false);
}

int kp_sim_set_obj_state(kp_sim* s, const float* obj_qpos, const float* obj_qvel, const uint8_t* mask) {
    if (!s || !obj_qpos || !obj_qvel) return fail("kp_sim_set_obj_state: null argument");
    if (!s->has_objects) return fail("kp_sim_set_obj_state: call kp_sim_set_objects first (it decides which objects are simulated)");
    HIP_OK(hipSetDevice(s->device));
    int n = s->n;
    hipLaunchKernelGGL(k_copy_rows, dim3((n * 35 + 255) / 256), dim3(256), 0, s->stream, n, 35, obj_qpos, s->obj_qpos, (float*)nullptr, mask);
    hipLaunchKernelGGL(k_copy_rows, dim3((n * 30 + 255) / 256), dim3(256), 0, s->stream, n, 30, obj_qvel, s->obj_qvel, (float*)nullptr, mask);
    HIP_OK(hipGetLastError());
    return 0;
}

int kp_sim_diag(kp_sim* s, int32_t* out_host) {
    if (!s || !out_host) return fail("kp_sim_diag: null argument");
    HIP_OK(hipSetDevice(s->device));
    HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipMemcpy(out_host, s->diag, sizeof(int) * 4 * (size_t)s->n, hipMemcpyDeviceToHost));
    unsigned ctr[4] = {0, 0, 0, 0};
    HIP_OK(hipMemcpy(ctr, s->jobctr, sizeof(ctr), hipMemcpyDeviceToHost));
    if (ctr[2]) {
        hipMemset(s->jobctr + 2, 0, sizeof(unsigned));     // reported once
        return fail("kp_step_queue_kernel: a wavefront gave up waiting for a job to be published (job queue stalled); states are incomplete");
    }
    return 0;
}

int kp_job_schedule(int n_substeps, int substeps_per_job, int taper, int* sizes16) {
    if (!sizes16 || n_substeps <= 0 || substeps_per_job <= 0 || n_substeps > 255) return fail("kp_job_schedule: need sizes16, 0 < n_substeps <= 255, substeps_per_job > 0");
    return job_schedule(n_substeps, substeps_per_job, taper, sizes16);
}

int kp_sim_lean_state(kp_sim* s, int32_t* out3) {
    if (!s || !out3) return fail("kp_sim_lean_state: null argument");
    const bool lean = !s->has_objects && s->model->lean_queue && s->model->threads == 64;
    out3[0] = lean && !(s->model->lean_adaptive && s->launch_index < s->lean_off_until);
    out3[1] = s->lean_fallbacks; out3[2] = (int32_t)s->launch_index;
    return 0;
}

int kp_sim_launch_cost(kp_sim* s, uint32_t* out_host) {
    if (!s || !out_host) return fail("kp_sim_launch_cost: null argument");
    HIP_OK(hipSetDevice(s->device));
    HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipMemcpy(out_host, s->cost, sizeof(unsigned) * (size_t)s->n, hipMemcpyDeviceToHost));
    return 0;
}

double kp_sim_last_step_seconds(kp_sim* s) {
    if (!s || !s->timed) return -1.0;
    if (hipEventSynchronize(s->last1) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->last0, s->last1) != hipSuccess) return -1.0;
    return ms * 1e-3;
}

int kp_sim_phase_cycles(kp_sim* s, double* out) {
    if (!s || !out) return fail("kp_sim_phase_cycles: null argument");
    if (!s->prof) return fail("kp_sim_phase_cycles: create the simulator with KP_PROFILE=1");
    HIP_OK(hipSetDevice(s->device));
    HIP_OK(hipStreamSynchronize(s->stream));
    std::vector<unsigned long long> h((size_t)s->n * 8);
    HIP_OK(hipMemcpy(h.data(), s->prof, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; k++) { double acc = 0; for (int e = 0; e < s->n; e++) acc += (double)h[(size_t)e * 8 + k]; out[k] = acc / s->n; }
    return 0;
}

int kp_sim_phase_cycles_env(kp_sim* s, double* out) {
    if (!s || !out) return fail("kp_sim_phase_cycles_env: null argument");
    if (!s->prof) return fail("kp_sim_phase_cycles_env: create the simulator with KP_PROFILE=1");
    HIP_OK(hipSetDevice(s->device));
    HIP_OK(hipStreamSynchronize(s->stream));
    std::vector<unsigned long long> h((size_t)s->n * 8);
    HIP_OK(hipMemcpy(h.data(), s->prof, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); i++) out[i] = (double)h[i];
    return 0;
}

int kp_sim_timing_reset(kp_sim* s) {
    if (!s) return fail("kp_sim_timing_reset: null argument");
    s->ring_used = 0; s->ring_on = true;
    return 0;
}

double kp_sim_timing_mean_seconds(kp_sim* s, int* n_launches) {
    if (n_launches) *n_launches = 0;
    if (!s || s->ring_used == 0) return -1.0;
    if (hipStreamSynchronize(s->stream) != hipSuccess) return -1.0;
    double tot = 0.0;
    for (int i = 0; i < s->ring_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s->ring[2 * i], s->ring[2 * i + 1]) != hipSuccess) return -1.0;
        tot += ms;
    }
    if (n_launches) *n_launches = s->ring_used;
    return tot * 1e-3 / s->ring_used;
}

}  // extern "C"

// ---- model compiler behind the ABI (host only).  Included last: its `#pragma clang fp contract(off)` (bit-reproducible double arithmetic,
// shared rule for rule with kinpoly_amd/model_compiler.py) must not reach the kernels above.
#include "kp_compile.hpp"

extern "C" {

int kp_model_compile(const char* xml_path, const char* uhc_yml_path, const char* out_kpm_path) {
    if (!xml_path || !out_kpm_path) return fail("kp_model_compile: null argument");
    std::string err;
    if (kpc::compile(xml_path, uhc_yml_path, out_kpm_path, err) != 0) return fail("kp_model_compile: " + err);
    return 0;
}

kp_model* kp_model_load_xml(const char* xml_path, const char* uhc_yml_path) {
    if (!xml_path) { fail("kp_model_load_xml: null argument"); return nullptr; }
    char tmpl[] = "/tmp/kp_model_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd < 0) { fail("kp_model_load_xml: cannot create a temporary file"); return nullptr; }
    close(fd);
    kp_model* m = nullptr;
    if (kp_model_compile(xml_path, uhc_yml_path, tmpl) == 0) m = kp_model_load(tmpl);
    std::remove(tmpl);
    return m;
}

}  // extern "C"

