// kp_rollout_kernels.hpp -- per-step bookkeeping of the kinematic-policy env and the rollout driver:
//   k_obs_ar       HumanoidAREnv.get_ar_obs_v1        kin_poly/envs/humanoid_ar_v1.py:133-214 (kin_poly.yml flags)
//   k_term_reward  calc_body_diff / calc_body_gt_diff  kin_poly/envs/humanoid_ar_v1.py:435-458
//                  dynamic_supervision_v1              kin_poly/core/reward_function.py:931-995
//   k_snapshot     prev_bquat / prev_hpos records      kin_poly/envs/humanoid_ar_v1.py:246-249
//   k_gae          estimate_advantages (un-normalised) uhc/khrylib/rl/core/common.py:5-25
// One thread per environment: these are O(100)-flop gathers; the physics kernel dominates the step.
#pragma once
#include "kp_obs_kernels.hpp"

namespace kp {

struct CtxDev {  // mirrors kp_ctx in include/kinpoly_sim.h
    int T;
    const float *head_pose, *head_vels, *obj_rel, *action_one_hot, *gt_bquat, *gt_wbpos, *obj_qpos;
    const int* cur_t;
    const int* row;      // optional: env e reads context row row[e] of the [R, T, .] arrays (null: row e)
    __device__ __forceinline__ size_t r(int e) const { return row ? (size_t)row[e] : (size_t)e; }
};

// action_index_map of HumanoidAREnv (humanoid_ar_v1.py:37-39): first column of the action's object(s) inside data.qpos[76:111]
// (sit -> chair 0, push -> box 7 (+ table 14), avoid -> Can 21, step -> step 28); -1 when the one-hot is all zero (no object, :465-466)
__device__ __forceinline__ int obj_action_start(const float* one_hot) {
    if (!one_hot) return -1;
    int a = -1;
    for (int k = 0; k < 4; k++) if (a < 0 && one_hot[k] != 0.f) a = k;
    return a < 0 ? -1 : (a == 0 ? 0 : (a == 1 ? 7 : (a == 2 ? 21 : 28)));
}

__device__ __forceinline__ V3 tv_heading(V3 v, Q4 q) { return q_tmul_vec(q_heading(q), v); }  // transform_vec(v, q, 'heading')

__global__ void k_obs_ar(int n, CtxDev C, const float* __restrict__ qpos, const float* __restrict__ xpos, const float* __restrict__ xquat,
                         float* __restrict__ out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* q = qpos + (size_t)e * D_NQ;
    float* o = out + (size_t)e * 105;
    int t = C.cur_t[e];
    t = t < 0 ? 0 : (t >= C.T ? C.T - 1 : t);
    const int hb = 13;
    Q4 rq = Q4{q[3], q[4], q[5], q[6]};
    Q4 dh = qmul(q_inverse(q_heading(rq)), rq);  // de_heading(qpos[3:7]) -- no base_rot removal here (:140-141)
    o[0] = q[2]; o[1] = dh.w; o[2] = dh.x; o[3] = dh.y; o[4] = dh.z;
    for (int j = 0; j < D_NU; j++) o[5 + j] = q[7 + j];
    V3 hpos = ld3(xpos + (size_t)e * 72 + 3 * hb);
    const float* hq4 = xquat + (size_t)e * 96 + 4 * hb;
    Q4 hrot = Q4{hq4[0], hq4[1], hq4[2], hq4[3]};
    const float* hp = C.head_pose + (C.r(e) * C.T + t) * 7;
    const float* hv = C.head_vels + (C.r(e) * C.T + t) * 6;
    const float* orl = C.obj_rel + (C.r(e) * C.T + t) * 7;
    const float* oh = C.action_one_hot + C.r(e) * 4;
    st3(o + 74, tv_heading(ld3(hp) - hpos, hrot));
    Q4 dr = qmul(q_inverse(Q4{hp[3], hp[4], hp[5], hp[6]}), hrot);
    o[77] = dr.w; o[78] = dr.x; o[79] = dr.y; o[80] = dr.z;
    float ohs = oh[0] + oh[1] + oh[2] + oh[3];
    V3 opos = v3(0.f, 0.f, 0.f); Q4 orot = Q4{1.f, 0.f, 0.f, 0.f};   // get_obj_qpos: [0,0,0,1,0,0,0] when no action (:465-466)
    if (ohs != 0.f && C.obj_qpos) { const float* ob = C.obj_qpos + (size_t)e * 7; opos = ld3(ob); orot = Q4{ob[3], ob[4], ob[5], ob[6]}; }
    st3(o + 81, tv_heading(opos - hpos, hrot));
    Q4 ol = qmul(q_inverse(q_heading(hrot)), orot);
    o[84] = ol.w; o[85] = ol.x; o[86] = ol.y; o[87] = ol.z;
    o[88] = hv[3]; o[89] = hv[4]; o[90] = hv[5];
    o[91] = hv[0]; o[92] = hv[1]; o[93] = hv[2];
    for (int k = 0; k < 7; k++) o[94 + k] = orl[k];
    for (int k = 0; k < 4; k++) o[101 + k] = oh[k];
}

__global__ void k_snapshot(int n, const float* __restrict__ qpos, const float* __restrict__ xpos, const float* __restrict__ xquat,
                           float* __restrict__ prev_bquat, float* __restrict__ prev_hpos) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * D_NB) return;
    int e = t / D_NB, b = t - e * D_NB;
    const float* q = qpos + (size_t)e * D_NQ;
    Q4 r = b == 0 ? Q4{q[3], q[4], q[5], q[6]} : q_euler_sxyz(q[7 + 3 * (b - 1)], q[8 + 3 * (b - 1)], q[9 + 3 * (b - 1)]);
    float* o = prev_bquat + (size_t)t * 4;
    o[0] = r.w; o[1] = r.x; o[2] = r.y; o[3] = r.z;
    if (b == 13) {
        for (int k = 0; k < 3; k++) prev_hpos[(size_t)e * 7 + k] = xpos[(size_t)e * 72 + 39 + k];
        for (int k = 0; k < 4; k++) prev_hpos[(size_t)e * 7 + 3 + k] = xquat[(size_t)e * 96 + 52 + k];
    }
}

__device__ __forceinline__ float quat_norm_v2(Q4 q) {  // multi_quat_norm_v2 of one quaternion
    float a = fabsf(q.w) - 1.0f;
    return sqrtf(a * a + q.x * q.x + q.y * q.y + q.z * q.z);
}
__device__ __forceinline__ V3 rot_from_quat(Q4 q) {  // rotation_from_quaternion (transformation.py:348-356)
    // the reference (fp64): zero if 1 - |w| < 1e-8, else xyz / sqrt(1 - w^2) * 2 acos(w).  In fp32 1 - w^2 of a small rotation (a body turning at
    // 0.1 rad/s between two control steps: 2e-6) is known to 3 %; for the unit quaternion q it equals |xyz|^2, and 1 - |w| = |xyz|^2 / (1 + |w|)
    const float s2 = q.x * q.x + q.y * q.y + q.z * q.z;
    if (s2 < 1e-8f * (1.0f + fabsf(q.w))) return v3(0.f, 0.f, 0.f);
    const float s = sqrtf(s2), ang = 2.0f * atan2f(s, q.w);
    return (ang / s) * v3(q.x, q.y, q.z);
}

struct RewardW { float w_hp, w_hq, w_p, w_jp, w_act_p, w_act_v, k_hp, k_hq, k_p, k_jp, k_act_p, k_act_v, dt, thresh, gt_thresh; int use_gt; };

// inputs are AFTER do_simulation and cur_t += 1: qpos fresh, xpos/xquat stale (as the reference reads them).
// 32 lanes per env (lane = body, 24 active), 8 envs per 256-thread block; the per-body terms are summed by an xor butterfly.
// POST = true is the fused tail of HumanoidAREnv.step (humanoid_ar_v1.py:288-316): cur_t += 1 first (written back by the env's lane 0
// after every lane has read it), then the same termination / reward, then end = cur_t >= min(env_episode_len, ar_context['len']),
// done = fail or end, percent = cur_t / ar_context['len'] -- one launch instead of the add, the compare, the or, the division and two copies.
struct PostStep {
    int* cur_t;                 // [N] in / out (the same buffer CtxDev::cur_t points to)
    const int* row_len;         // [R] ar_context['len'] of every context row
    int episode_len;            // cc_cfg.env_episode_len
    uint8_t *done, *end;        // [N]
    float* percent;             // [N]
    int* done_count;            // optional: += number of done envs (a rollout loop's episode counter, no extra reduction launch)
    float* obj7;                // optional [N,7]: get_obj_qpos(action_one_hot) after the step = the simulated pose of the action's (first) object,
    const float* sim_obj_qpos;  //   read from the simulator's data.qpos[76:111] rows [N,35]; envs whose clip has no action keep their obj7
};
template <bool POST>
__global__ __launch_bounds__(256) void k_term_reward(int n, CtxDev C, RewardW W, const float* __restrict__ qpos, const float* __restrict__ xpos,
                              const float* __restrict__ xquat, const float* __restrict__ t_wbpos, const float* __restrict__ t_bquat,
                              const float* __restrict__ prev_bquat, const float* __restrict__ prev_hpos, const float* __restrict__ diffw,
                              float* __restrict__ reward, float* __restrict__ info, uint8_t* __restrict__ fail, float* __restrict__ diffs, PostStep PS) {
    const int b = threadIdx.x & 31, e_raw = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool valid = e_raw < n;
    const int e = valid ? e_raw : 0;
    const int t_now = C.cur_t[e] + (POST ? 1 : 0);
    int t = t_now;
    t = t < 1 ? 1 : (t >= C.T ? C.T - 1 : t);
    const float* q = qpos + (size_t)e * D_NQ;
    const float* xp = xpos + (size_t)e * 72;
    const float* gtb = C.gt_bquat + (C.r(e) * C.T + t) * 96;
    const float* gtp = C.gt_bquat + (C.r(e) * C.T + t - 1) * 96;
    const float* gtw = C.gt_wbpos + (C.r(e) * C.T + t) * 72;
    float pq = 0.f, pp = 0.f, pg = 0.f, vel2 = 0.f, bd = 0.f, bgd = 0.f;
    if (b < D_NB) {
        const Q4 cb = b == 0 ? Q4{q[3], q[4], q[5], q[6]} : q_euler_sxyz(q[7 + 3 * (b - 1)], q[8 + 3 * (b - 1)], q[9 + 3 * (b - 1)]);
        const float* tb = t_bquat + (size_t)e * 96 + 4 * b;
        pq = quat_norm_v2(qmul(cb, q_inverse(Q4{tb[0], tb[1], tb[2], tb[3]})));
        const V3 x = ld3(xp + 3 * b);
        const V3 dpos = x - ld3(t_wbpos + (size_t)e * 72 + 3 * b);
        pp = sqrtf(dot(dpos, dpos));
        const float wgt = diffw ? diffw[b] : 1.0f;
        bd = wgt * pp;
        const V3 dg = x - ld3(gtw + 3 * b);
        bgd = wgt * sqrtf(dot(dg, dg));
        const Q4 g = Q4{gtb[4 * b], gtb[4 * b + 1], gtb[4 * b + 2], gtb[4 * b + 3]};
        const Q4 gpv = Q4{gtp[4 * b], gtp[4 * b + 1], gtp[4 * b + 2], gtp[4 * b + 3]};
        pg = quat_norm_v2(qmul(g, q_inverse(cb)));
        const float* pb = prev_bquat + (size_t)e * 96 + 4 * b;
        const V3 cv = (1.0f / W.dt) * rot_from_quat(qmul(cb, q_inverse(Q4{pb[0], pb[1], pb[2], pb[3]})));
        const V3 gv = (1.0f / W.dt) * rot_from_quat(qmul(g, q_inverse(gpv)));
        const V3 dv = cv - gv;
        vel2 = dot(dv, dv);
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        pq += __shfl_xor(pq, m, 32); pp += __shfl_xor(pp, m, 32); pg += __shfl_xor(pg, m, 32);
        vel2 += __shfl_xor(vel2, m, 32); bd += __shfl_xor(bd, m, 32); bgd += __shfl_xor(bgd, m, 32);
    }
    if (b != 0 || !valid) return;
    const float* xq = xquat + (size_t)e * 96;
    const float* hp = C.head_pose + (C.r(e) * C.T + t) * 7;
    const V3 hd = ld3(xp + 39) - ld3(hp);
    const float hp_r = expf(-W.k_hp * dot(hd, hd));
    const Q4 hq = qmul(Q4{xq[52], xq[53], xq[54], xq[55]}, q_inverse(Q4{hp[3], hp[4], hp[5], hp[6]}));
    const float hqd = quat_norm_v2(hq);
    const float hq_r = expf(-W.k_hq * hqd * hqd);
    pq *= (1.0f / 24.0f); pp *= (1.0f / 24.0f); pg *= (1.0f / 24.0f);
    const float p_r = expf(-W.k_p * pq * pq), jp_r = expf(-W.k_jp * pp * pp);
    const float gt_r = expf(-W.k_act_p * pg), av_r = expf(-W.k_act_v * vel2);
    reward[e] = W.w_hp * hp_r + W.w_hq * hq_r + W.w_p * p_r + W.w_jp * jp_r + W.w_act_p * gt_r + W.w_act_v * av_r;
    float* inf = info + (size_t)e * 6;
    inf[0] = hp_r; inf[1] = hq_r; inf[2] = p_r; inf[3] = jp_r; inf[4] = gt_r; inf[5] = av_r;
    const bool failed = (bd > W.thresh) || (W.use_gt && bgd > W.gt_thresh) || !(bd == bd);
    fail[e] = failed;
    diffs[2 * e] = bd; diffs[2 * e + 1] = bgd;
    if (POST) {
        const int clen = PS.row_len[C.r(e)];
        const bool ended = t_now >= (clen < PS.episode_len ? clen : PS.episode_len);
        PS.cur_t[e] = t_now;
        PS.end[e] = ended; PS.done[e] = failed || ended;
        PS.percent[e] = (float)t_now / (float)clen;
        if (PS.done_count && (failed || ended)) atomicAdd(PS.done_count, 1);
        if (PS.obj7) {
            const int st = obj_action_start(C.action_one_hot + C.r(e) * 4);
            if (st >= 0) for (int k = 0; k < 7; k++) PS.obj7[(size_t)e * 7 + k] = PS.sim_obj_qpos[(size_t)e * 35 + st + k];
        }
    }
}

// Masked reset of HumanoidAREnv (mujoco_env.py:86-103 + humanoid_ar_v1.py:334-387) for the envs with mask != 0: cur_t = 0 and
// qpos / qvel (and the derived-state copies) <- ar_context['init_qpos' / 'init_qvel'] of the env's context row, warm start zeroed.  One
// launch instead of a masked fill, an index conversion, two gathers and three masked row copies; sim.forward() and the target FK follow.
__global__ void k_reset_rows(int n, const float* __restrict__ init_qpos, const float* __restrict__ init_qvel, const int* __restrict__ row,
                             const uint8_t* __restrict__ mask, int* __restrict__ cur_t, float* __restrict__ qpos, float* __restrict__ qvel,
                             float* __restrict__ qpos_d, float* __restrict__ qvel_d, float* __restrict__ warm, float* __restrict__ aux_rows, int aux_cols) {
    const int e = blockIdx.x, i = threadIdx.x;          // one 128-thread block per env
    if (e >= n || (mask && !mask[e])) return;
    for (int c = i; c < aux_cols; c += 128) aux_rows[(size_t)e * aux_cols + c] = 0.f;      // the caller's per-env rows that die with the episode (a recurrent policy's hidden state)
    const size_t r = row ? (size_t)row[e] : (size_t)e;
    if (i < D_NQ) { const float v = init_qpos[r * D_NQ + i]; qpos[(size_t)e * D_NQ + i] = v; qpos_d[(size_t)e * D_NQ + i] = v; }
    if (i < D_NV) { const float v = init_qvel[r * D_NV + i]; qvel[(size_t)e * D_NV + i] = v; qvel_d[(size_t)e * D_NV + i] = v; warm[(size_t)e * D_NV + i] = 0.f; }
    if (i == 0 && cur_t) cur_t[e] = 0;
}

// Episode turnover of a sampler that keeps every env's NEXT clips resident as a ring of n_slots context rows per env (row = slot * n + env):
// a finished env moves to the next slot of its ring and has one queued clip less (sample_seq -> init_context -> load_context of the next
// episode, agent_ar.py:518-535, made ahead of time).  One launch instead of an add, a remainder, a subtract, a multiply-add, a cast and two masked copies.
__global__ void k_pool_advance(int n, int n_slots, const uint8_t* __restrict__ done, int* __restrict__ head, int* __restrict__ ahead, int* __restrict__ row) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || !done[e]) return;
    // an env with nothing queued behind it stays on the row it is on (it replays its clip) instead of moving onto a slot the host may be rewriting: by
    // construction this cannot happen (one episode per step at most, the ring is sized for the periods between its refills); if it ever did, `ahead` still
    // goes negative and the sampler's -- possibly one period late -- host check raises, but no env has read a stale or half-written row by then
    const int a = ahead[e];
    ahead[e] = a - 1;
    if (a <= 0) return;
    int h = head[e] + 1;
    if (h >= n_slots) h = 0;
    head[e] = h;
    row[e] = h * n + e;
}

// The sampler's per-step record (Memory.push of sample_worker, kin_poly/core/agent_ar.py:582-597) for all envs at once: every row of the env-major
// [N, T, .] rollout buffers at time t in ONE launch per half-step instead of a dozen strided copies.  `pre` (before env.step): states <- obs,
// episode_start <- fresh, curr_qpos <- the simulator's qpos, gt_target_qpos <- ar_context['qpos'][cur_t + 1] of the env's context row (clamped to the
// clip's last frame), meta <- (take_ind, fr_start) of that row.  `post` (after env.step, before the episode turnover): actions, reward, fail, done,
// percent, custom_info and -- for the full 12-field record -- next_states, res_qpos, cc_action, cc_state, v_metas.  One workgroup per env, lanes over
// the row's floats; every destination pointer may be null (field not recorded).
struct RecordPre {
    int n, T, t, ctx_T;
    const float* obs; const uint8_t* fresh; const float* qpos; const float* ctx_qpos; const int* row; const int* cur_t; const int* row_len; const float* row_meta;
    float* states; uint8_t* episode_start; float* curr_qpos; float* gt_target_qpos; float* meta;
};
struct RecordPost {
    int n, T, t; float fr_num;
    const float* action; const float* reward; const uint8_t* fail; const uint8_t* done; const float* percent; const float* c_info;
    const float* obs; const float* qpos; const float* cc_action; const float* cc_state; const float* meta_t;      // meta_t: this step's [n, 2] rows of `meta` (pre wrote them)
    float* actions; float* rewards; uint8_t* fails; uint8_t* dones; float* percents; float* c_infos;
    float* next_states; float* res_qpos; float* cc_actions; float* cc_states; float* v_metas;
};

__device__ __forceinline__ void copy_row(float* __restrict__ dst, const float* __restrict__ src, int dim) {
    for (int i = threadIdx.x; i < dim; i += blockDim.x) dst[i] = src[i];
}

__global__ void k_record_pre(RecordPre R) {
    const int e = blockIdx.x;
    if (e >= R.n) return;
    const size_t at = (size_t)e * R.T + R.t;
    if (R.states) copy_row(R.states + at * 105, R.obs + (size_t)e * 105, 105);
    if (R.curr_qpos) copy_row(R.curr_qpos + at * 76, R.qpos + (size_t)e * 76, 76);
    const int r = R.row ? R.row[e] : e;
    if (R.gt_target_qpos) {
        int f = R.cur_t[e] + 1;
        const int last = R.row_len[r];
        if (f > last) f = last;
        copy_row(R.gt_target_qpos + at * 76, R.ctx_qpos + ((size_t)r * R.ctx_T + f) * 76, 76);
    }
    if (threadIdx.x == 0) {
        if (R.episode_start) R.episode_start[at] = R.fresh[e];
        if (R.meta) { R.meta[at * 2] = R.row_meta[(size_t)r * 2]; R.meta[at * 2 + 1] = R.row_meta[(size_t)r * 2 + 1]; }
    }
}

__global__ void k_record_post(RecordPost R) {
    const int e = blockIdx.x;
    if (e >= R.n) return;
    const size_t at = (size_t)e * R.T + R.t;
    if (R.actions) copy_row(R.actions + at * 80, R.action + (size_t)e * 80, 80);
    if (R.next_states) copy_row(R.next_states + at * 105, R.obs + (size_t)e * 105, 105);
    if (R.res_qpos) copy_row(R.res_qpos + at * 76, R.qpos + (size_t)e * 76, 76);
    if (R.cc_actions) copy_row(R.cc_actions + at * 75, R.cc_action + (size_t)e * 75, 75);
    if (R.cc_states) copy_row(R.cc_states + at * 784, R.cc_state + (size_t)e * 784, 784);
    if (R.c_infos && threadIdx.x < 6) R.c_infos[at * 6 + threadIdx.x] = R.c_info[(size_t)e * 6 + threadIdx.x];
    if (threadIdx.x == 0) {
        if (R.rewards) R.rewards[at] = R.reward[e];
        if (R.fails) R.fails[at] = R.fail[e];
        if (R.dones) R.dones[at] = R.done[e];
        if (R.percents) R.percents[at] = R.percent[e];
        if (R.v_metas) { R.v_metas[at * 3] = R.meta_t[at * 2]; R.v_metas[at * 3 + 1] = R.meta_t[at * 2 + 1]; R.v_metas[at * 3 + 2] = R.fr_num; }
    }
}

// reverse scan per env over an env-major [N, T] layout (time contiguous per env), masks cut episodes
// last_values (optional, [n]): V(s_T) of the state after each env's last row -- the bootstrap of an episode the horizon cut (the row's
// mask is 1 there); null = 0, the reference's flat-batch recursion whose last row always ends an episode
__global__ void k_gae(int n, int T, const float* __restrict__ rewards, const float* __restrict__ masks, const float* __restrict__ values,
                      const float* __restrict__ last_values, float gamma, float tau, float* __restrict__ adv, float* __restrict__ ret) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float prev_v = last_values ? last_values[e] : 0.f, prev_a = 0.f;
    for (int t = T - 1; t >= 0; t--) {
        size_t i = (size_t)e * T + t;
        float m = masks[i], v = values[i];
        float delta = rewards[i] + gamma * prev_v * m - v;
        float a = delta + gamma * tau * prev_a * m;
        adv[i] = a; ret[i] = v + a;
        prev_v = v; prev_a = a;
    }
}

// ---------------------------------------------------------------- GRU re-unroll of the updates
// The PPO epochs and the supervised step update re-run the kinematic policy's GRU over the whole batch (policy_ar.py:104-122, 216-240:
// padded [T_max, n_episodes] re-pack, hidden state zero at every episode start).  Here the batch is env-major / time-stepped:
//     gi_t = x_t W_ih^T + b_ih  for all t in ONE GEMM,   gh_t = hm_{t-1} W_hh^T + b_hh  one GEMM per step (MFMA, library),
// and these two kernels do everything else of a step, forward and backward, in one pass over [N, H]:
//     r = sigma(gi_r + gh_r), z = sigma(gi_z + gh_z), n = tanh(gi_n + r gh_n), h = (1 - z) n + z hm_prev     (torch.nn.GRUCell)
// hm = h masked by the NEXT step's episode-start flag, ready to be the next step's GEMM input.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// PolicyMCP's mixing stage (uhc/core/policy_mcp.py:30-38) in one pass: weight = softmax(composer output) over the K primitives,
// mean = sum_k weight_k * prim_k, optionally action = mean + std * noise (select_action, policy.py:12-15).  prim is the third batched GEMM's
// output in its own layout [K, N, A] (no transpose pass); one thread per (env, action dim), the K <= 16 logits of an env are re-read by its A threads (L1).
__global__ void k_mcp_compose(int n, int K, int A, const float* __restrict__ logits, const float* __restrict__ prim, const float* __restrict__ noise,
                              int noise_stride, const float* __restrict__ stdv, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * A) return;
    const int e = (int)(i / A), a = (int)(i - (size_t)e * A);
    const float* lg = logits + (size_t)e * K;
    float mx = lg[0];
    for (int k = 1; k < K; k++) mx = fmaxf(mx, lg[k]);
    float den = 0.f, acc = 0.f;
    for (int k = 0; k < K; k++) {
        const float w = expf(lg[k] - mx);
        den += w;
        acc += w * prim[((size_t)k * n + e) * A + a];
    }
    float v = acc / den;
    if (noise) v += stdv[a] * noise[(size_t)e * noise_stride + a];
    out[i] = v;
}

__global__ void k_gru_gates_fwd(int n, int H, const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ hm_prev,
                                const float* __restrict__ next_keep, float* __restrict__ h_out, float* __restrict__ hm_next) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * H) return;
    const int e = (int)(i / H), j = (int)(i - (size_t)e * H);
    const float* a = gi + (size_t)e * 3 * H; const float* b = gh + (size_t)e * 3 * H;
    const float r = sigmoidf_(a[j] + b[j]), z = sigmoidf_(a[H + j] + b[H + j]), nn = tanhf(a[2 * H + j] + r * b[2 * H + j]);
    const float h = (1.0f - z) * nn + z * hm_prev[i];
    h_out[i] = h;
    if (hm_next) hm_next[i] = next_keep ? h * next_keep[e] : h;
}

// dh = dh_out (gradient reaching h_t from its consumers outside the recurrence) + carry * carry_keep (from step t + 1 through hm_t).
// Outputs: dgi (3H: gradient of gi_t), dgh (3H: gradient of gh_t, the GEMM operand), dhz = dh * z (the direct path into hm_{t-1}).
__global__ void k_gru_gates_bwd(int n, int H, const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ hm_prev,
                                const float* __restrict__ dh_out, const float* __restrict__ carry, const float* __restrict__ carry_keep,
                                float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dhz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * H) return;
    const int e = (int)(i / H), j = (int)(i - (size_t)e * H);
    const float* a = gi + (size_t)e * 3 * H; const float* b = gh + (size_t)e * 3 * H;
    const float ghn = b[2 * H + j];
    const float r = sigmoidf_(a[j] + b[j]), z = sigmoidf_(a[H + j] + b[H + j]), nn = tanhf(a[2 * H + j] + r * ghn);
    float dh = dh_out ? dh_out[i] : 0.f;
    if (carry) dh += carry[i] * (carry_keep ? carry_keep[e] : 1.0f);
    const float dn = dh * (1.0f - z), dz = dh * (hm_prev[i] - nn);
    const float dan = dn * (1.0f - nn * nn);           // d(gi_n)
    const float dr = dan * ghn;
    const float dar = dr * r * (1.0f - r), daz = dz * z * (1.0f - z);
    float* gi_o = dgi + (size_t)e * 3 * H; float* gh_o = dgh + (size_t)e * 3 * H;
    gi_o[j] = dar; gi_o[H + j] = daz; gi_o[2 * H + j] = dan;
    gh_o[j] = dar; gh_o[H + j] = daz; gh_o[2 * H + j] = dan * r;
    dhz[i] = dh * z;
}

}  // namespace kp
