/* kinpoly_sim.h -- C ABI of libkinpoly_sim.so: the batched MI355X-native replacement of the
 * MuJoCo rollout path of KinPoly (one call = N environments).
 *
 * The reference has no FFI for this path; the two Python surfaces it replaces are
 *   B1  HumanoidAREnv            kin_poly/envs/humanoid_ar_v1.py:28-516  (+ uhc/envs/humanoid_im.py)
 *   B2  the mujoco-py object API uhc/khrylib/rl/envs/common/mujoco_env.py:23-24,87,99-103,
 *                                 uhc/envs/humanoid_im.py:193,203,217,423,426,504,516,527
 * Each entry point cites the reference interface it stands in for.  INTEGRATION.md shows the ctypes
 * binding a reference maintainer adds.
 *
 * Conventions
 *   - every array argument is a DEVICE pointer (hipMalloc / torch tensor .data_ptr()) to a dense
 *     row-major float32 array [n_envs, dim]; `env_mask` is uint8 [n_envs] (1 = apply) or NULL = all;
 *   - all work is enqueued on the HIP stream given to kp_sim_create (no implicit sync), except
 *     kp_sim_diag() which synchronises that stream;
 *   - return value 0 = ok, negative = error (kp_last_error() gives the text, thread-local);
 *   - handles are opaque; one kp_sim is used by one host thread at a time.
 *
 * Several handles in one process
 *   Any number of kp_sim handles (of the same or of different models) may live in a process and run at the same time on different
 *   HIP streams (kp_sim_create's stream / kp_sim_set_stream): a handle owns every array its kernels write -- state, job queue and its
 *   counters, status words, timing events -- and the library keeps no process-wide device state (no __constant__ / __device__ symbols),
 *   so launches of different handles do not interact except by sharing the machine.  The persistent job-queue kernel of the control step
 *   (kp_step_queue_kernel) waits only for jobs that a RUNNING wavefront of the same launch is about to publish, never for a wavefront that
 *   has yet to become resident, so co-scheduled launches cannot starve each other into a deadlock; a wait that exceeds 2 s raises the
 *   handle's stall flag (kp_sim_status_device word 2 -> the next host-synchronising call fails) instead of hanging.
 *   Measured on MI355X (tests/test_gpu_round5.py::test_concurrent_handles_...; tools/micro/concurrent_handles.py, profiles/r05): two and
 *   three handles of 4096 envs each -- floor scenes and scenes with free objects, in any mix -- stepped 50 control steps on their own
 *   streams with no host synchronisation end in the same states, bit for bit, as when they run one after the other, with clean status
 *   words; the round of launches is 5 ... 20 % shorter than the serial one (the launches' tails overlap).  (The device hang that rounds
 *   3 / 4 saw with THREE sub-batched env-steps on three streams is reproduced by the policies' library GEMMs ALONE at 1365 = 4096 / 3 rows on
 *   three streams -- no kernel of this library in the loop -- and not at 1024 rows: profiles/r05/gemm_streams_probe.log)
 *   What is NOT supported: two host threads in one handle at once; one handle on two streams at once (kp_sim_set_stream moves it, the
 *   caller orders the old stream's work before the new stream's).
 */
#ifndef KINPOLY_SIM_H
#define KINPOLY_SIM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct kp_model kp_model;
typedef struct kp_sim kp_sim;

/* layout constants of the SMPL humanoid (reference humanoid_im.py:30-49 qpos_lim/qvel_lim/body_lim) */
#define KP_NQ 76
#define KP_NV 75
#define KP_NU 69
#define KP_NBODY 24
#define KP_CC_OBS_DIM 784   /* get_full_obs_v1, humanoid_im.py:144-233 */
#define KP_AR_OBS_DIM 105   /* get_ar_obs_v1, humanoid_ar_v1.py:133-214 (kin_poly.yml flags) */
#define KP_KIN_ACTION_DIM 80
#define KP_CC_ACTION_DIM 75

/* ---- model --------------------------------------------------------------------------------------
 * replaces mujoco_py.load_model_from_path(xml) (mujoco_env.py:23): `kpm_path` is the blob
 * kinpoly_amd/model_compiler.py compiles from the same XML + STL hulls + uhc.yml gains. */
kp_model* kp_model_load(const char* kpm_path);
/* mujoco_py.load_model_from_path(xml) itself (mujoco_env.py:23), no Python: compile the reference's scene XML -- with the binary STL meshes it
 * names (resolved relative to the XML), `coordinate="global"`, mesh-hull humanoid bodies with a free root + z / y / x hinges, box / cylinder
 * free objects -- and the PD-gain table of config/uhc/uhc.yml (may be NULL: zero gains) into the blob, then load it.  kp_model_compile writes
 * the blob to a file (same bytes as kinpoly_amd/model_compiler.py up to floating-point rounding of the fields that go through a matrix
 * inverse / eigenvectors; integer tables identical). */
int kp_model_compile(const char* xml_path, const char* uhc_yml_path, const char* out_kpm_path);
kp_model* kp_model_load_xml(const char* xml_path, const char* uhc_yml_path);
void kp_model_free(kp_model*);
/* options: "contact" (0/1), "limits" (0/1), "gravity_z" ("gravity_x", "gravity_y": mjOption.gravity, 0 in the reference's XML; tilted-plane tests),
 * "actuation" (0/1, default 1; 0: ctrl = qfrc_applied = 0, i.e. do_simulation without compute_torque / rfc: torque-free motion for tests), "stale_kinematics" (0/1, default 1: SPD and
 * read-outs see the one-substep-stale derived quantities mujoco-py exposes), "solver_iter" (default: the blob's
 * mjOption.iterations = 100; cap hits are counted in kp_sim_diag), "solver_tol", "threads_per_env" (64/128/256), "dynamic_objects" (0/1);
 * scheduling only (results do not depend on them): "substeps_per_job" (default 4; 0 = one workgroup per env and control step):
 * with more envs than resident wavefront slots a control step is cut into jobs of that many substeps (the last job; each earlier job is
 * "job_taper" substeps longer, default 1: 15 = 6 + 5 + 4; 0 = uniform) which resident waves pull from a FIFO, so that the
 * launch does not end on the tail of its longest envs; "queue_slots" (0 = CUs x 8, x 6 with objects); "queue_fence" (0/1, default 1: the job hand-over is an agent-scope
 * release / acquire fence pair around relaxed write-through accesses, correct by the HIP memory model; 0 = without the fences, +0.5 %);
 * "queue_heavy" (default 160; 0 = off): a wave whose job ran at more than that percentage of the launch's mean time per substep keeps its env and runs the
 * env's next job itself instead of queueing it -- the costliest envs are the ones a launch ends on, and with two envs per slot every trip through the
 * FIFO costs them about one job's length of waiting (objects workload 6.49 -> 5.78 ms per launch; bit-identical results);
 * "lean_queue" (0/1, default 1): floor scenes' queue launches run on the lean LDS layout (12 784 B per env: 12 envs per CU = three waves per SIMD instead of 8 = two;
 * same results bit for bit).  Its defaults, unless the caller sets the option: jobs of 5 + 5 + 5 substeps, "queue_late" (default -1 = automatic: on with the lean
 * layout): an env whose first job had to wait for a slot is never queued again, its wave runs its later jobs itself; "queue_prio" (default -1 = automatic: 3 with
 * the lean layout, else 0): a wave's issue priority (s_setprio) -- 1 envs known to be heavy, 2 by the env's remaining jobs, 3 by its remaining substeps, re-set at
 * every substep.  "lean_max_contacts" (<= 24): a lean job that finds more contacts in a substep hands the env (before it has stored anything) to a second kernel
 * on the full layout; "lean_adaptive" (0/1, default 1): when more than 1 / 64 of the envs did so, the next 64 control steps run on the full layout (kp_sim_lean_state).
 * "lds_pad" (bytes): allocate at least that much LDS per env (experiments: fewer envs per CU with the same binary);
 * "lpt_order" (1 / 0 / -1 = default: on when free objects are simulated): longest-env-first order of the workgroups (plain launch) or of the
 * envs' first jobs in the FIFO, from the previous control step's per-env cycles;
 * "warm_extrap" (beta; default -1 = automatic: 0.75 when the scene's free objects are simulated, 0 otherwise): starting point of the constraint solve.  0 is
 * MuJoCo's: the previous substep's solution a_{k-1} (mjData.qacc_warmstart).  beta != 0 starts from a_{k-1} + beta (a_{k-1} - a_{k-2}) from the second substep
 * of a kp_sim_step_ctrl call on (every call begins with the plain warm start).  Same strictly convex problem, same minimiser, same termination tests: only the
 * iteration path changes (fewer Newton iterations where accelerations change smoothly -- falls, impacts; more on quiet standing states); results do not
 * depend on how a control step is cut into jobs. */
int kp_model_set_option(kp_model*, const char* name, double value);
double kp_model_get_option(const kp_model*, const char* name);

/* ---- simulator ------------------------------------------------------------------------------------
 * replaces MjSim(model) x N (mujoco_env.py:24) */
kp_sim* kp_sim_create(const kp_model*, int n_envs, int device_id, void* hip_stream);
void kp_sim_destroy(kp_sim*);
int kp_sim_n_envs(const kp_sim*);

/* Rebind the HIP stream all later calls enqueue on (kp_sim_create's stream otherwise).  The caller orders the two streams
 * (e.g. an event) if work is still in flight on the old one. */
int kp_sim_set_stream(kp_sim*, void* hip_stream);

/* Device pointer to uint32[4] the control-step launches maintain: [2] != 0 means a kp_step_queue_kernel launch stalled (a wave gave
 * up waiting for a job; sticky until the next kp_sim_diag reports it) -- a rollout loop can fold it into its own device-side
 * checks without a host sync.  [0], [1] are the queue's head / tail counters of the last launch. */
const uint32_t* kp_sim_status_device(kp_sim*);

/* Test / debugging read-out of data.contact: the contact set of the LAST collision pass of the last kp_sim_step_ctrl launch
 * (i.e. of the state before its final substep).  The first call arms the recording (costs a few KB of stores per env and launch
 * afterwards), later calls copy float32 [N, 1 + 64 * 9] to the HOST pointer: [0] = number of contacts, then per contact
 * {entity carrying the first geom's counterpart (0..23 hull body, 24 + k object slot), other entity (-1 world), dist, pos[3],
 * normal[3] (pointing into the first entity)}.  Synchronises. */
int kp_sim_contacts(kp_sim*, float* out_host);

/* mjf.mj_fullM(model, M, data.qM) and data.qfrc_bias as compute_desired_accel reads them (uhc/envs/humanoid_im.py:422-426):
 * M [N,75,75] (dense, symmetric, armature on the diagonal), bias [N,75]; either may be NULL.  The simulator never forms them
 * on the hot path (matrix-free solves); this read-out is for callers of the reference surface.  Same data as KP_M / KP_BIAS. */
int kp_sim_mass_matrix(kp_sim*, float* M, float* bias);

/* sim.reset() + set_state(qpos, qvel) + sim.forward()   (mujoco_env.py:86-103) for the masked envs.
 * qpos [N,76], qvel [N,75]. */
int kp_sim_set_state(kp_sim*, const float* qpos, const float* qvel, const uint8_t* env_mask);

/* self.target = smpl_humanoid.qpos_fk(target_qpos)   (humanoid_ar_v1.py:256, numpy_smpl_humanoid.py:180)
 * target_qpos [N,76] is copied; the target dict {qpos, wbpos, wbquat, bquat, body_com} is kept on device. */
int kp_sim_set_target(kp_sim*, const float* target_qpos, const uint8_t* env_mask);

/* object block of set_state: obj_qpos [N,35] = data.qpos[76:111] as reset_model builds it with convert_obj_qpos
 * (humanoid_ar_v1.py:377-381, 479-496: inactive objects parked at [(i+1)*100, 100, 0]), object velocities zero (:382).
 * Objects within 50 m of the origin (at most two: push = box + table) become free rigid bodies of the env: their
 * boxes / cylinders (XML :190-214) collide with the humanoid hulls, the floor and each other, and they are integrated
 * by the same soft-constraint solve as the humanoid.  Model option "dynamic_objects" = 0 freezes them as static obstacles.
 * Parked objects are not simulated (in the reference they only bounce on the floor 100 m away). */
int kp_sim_set_objects(kp_sim*, const float* obj_qpos, const uint8_t* env_mask);

/* Humanoid.qpos_fk_batch(qpos) (numpy_smpl_humanoid.py:124-178) on n_rows arbitrary rows, used by
 * load_context for the GT clip (humanoid_ar_v1.py:87): qpos [n_rows,76] -> qpos_out [n_rows,76] (root quat
 * normalised), wbpos [n_rows,72], wbquat [n_rows,96], bquat [n_rows,96], body_com [n_rows,72]; outputs may be NULL. */
int kp_sim_fk(kp_sim*, int n_rows, const float* qpos, float* qpos_out, float* wbpos, float* wbquat, float* bquat, float* body_com);

/* backward of qpos -> wbpos of the same rows (the autograd path through Humanoid.qpos_fk that TrajARNet.compute_loss_lite's
 * end-effector term takes, traj_ar_smpl_net.py:459-497, torch_smpl_humanoid.py:125-202): grad_qpos [n_rows,76] =
 * (d wbpos / d qpos)^T grad_wbpos [n_rows,72]; wbpos / wbquat are the outputs of the kp_sim_fk call on those rows. */
int kp_sim_fk_backward(kp_sim*, int n_rows, const float* qpos, const float* wbpos, const float* wbquat, const float* grad_wbpos, float* grad_qpos);

/* HumanoidEnv.do_simulation(cc_action, n_substeps)   (humanoid_im.py:506-533): per substep stable-PD
 * torque (compute_torque :433-480), clip, rfc_implicit (:497-504), sim.step() (:527).  cc_action [N,75]. */
int kp_sim_step_ctrl(kp_sim*, const float* cc_action, int n_substeps, const uint8_t* env_mask);

/* HumanoidAREnv.step_ar(a)   (humanoid_ar_v1.py:216-241): kin_action [N,80] -> next_qpos [N,76] */
int kp_sim_step_kin(kp_sim*, const float* kin_action, float* next_qpos);

/* The head of HumanoidAREnv.step in one launch (humanoid_ar_v1.py:246-256): kp_sim_step_begin + kp_sim_step_kin + kp_sim_set_target with the
 * kinematic step's result -- prev_bquat / prev_hpos recorded, target = qpos_fk(step_ar(kin_action)).  kin_action [N,80]. */
int kp_sim_step_head(kp_sim*, const float* kin_action);

/* get_full_obs_v1() [N,784]   (humanoid_im.py:144-233), optional ZFilter(update=False) + clip
 * (zfilter.py:58-67): pass mean/std [784] device pointers or NULL, clip <= 0 disables clipping. */
int kp_sim_obs_cc(kp_sim*, float* out, const float* zf_mean, const float* zf_std, float clip);

/* per-episode context rows the env reads each step (ar_context, humanoid_ar_v1.py:84-88).
 * Arrays are [R, T, dim] device pointers (action_one_hot [R, 4]), R >= N context rows; env e reads row `row[e]` (int32 [N], or
 * NULL: row e, R = N).  The indirection lets a sampler keep the NEXT episodes' contexts resident next to the current ones and
 * switch an env to its new clip on `done` without copying (agent_ar.py:519-537 draws a new clip per episode).  cur_t is
 * int32 [N] (env.cur_t).  obj_qpos [N,7] (per env, not per row) may be NULL (then get_obj_qpos() == [0,0,0,1,0,0,0], :465-466). */
typedef struct {
    int T;
    const float* head_pose;               /* [N,T,7]  ar_context['head_pose'] */
    const float* head_vels;               /* [N,T,6]  ar_context['head_vels'] */
    const float* obj_head_relative_poses; /* [N,T,7]  */
    const float* action_one_hot;          /* [N,4]    ar_context['action_one_hot'][0] */
    const float* gt_bquat;                /* [N,T,96] ar_context['bquat'] (GT clip) */
    const float* gt_wbpos;                /* [N,T,72] gt_targets['wbpos'] */
    const float* obj_qpos;                /* [N,7] or NULL */
    const int32_t* cur_t;                 /* [N] */
    const int32_t* row;                   /* [N] context row of every env, or NULL */
} kp_ctx;

/* records prev_bquat / prev_hpos at the top of HumanoidAREnv.step (humanoid_ar_v1.py:246-249) */
int kp_sim_step_begin(kp_sim*);

/* get_ar_obs_v1() [N,105]   (humanoid_ar_v1.py:133-214; use_head, use_action, use_obj on; use_vel/of/context off) */
int kp_sim_obs_ar(kp_sim*, const kp_ctx* ctx, float* out);

/* termination (calc_body_diff / calc_body_gt_diff, :435-458, thresholds :53-54) and the reward
 * dynamic_supervision_v1 (kin_poly/core/reward_function.py:931-995) for the state after do_simulation,
 * with ctx->cur_t already incremented.  reward [N], info [N,6], fail uint8 [N], diffs [N,2]. */
typedef struct {
    float w_hp, w_hq, w_p, w_jp, w_act_p, w_act_v;
    float k_hp, k_hq, k_p, k_jp, k_act_p, k_act_v;
    float dt;                  /* env.dt = 1/30 */
    float body_diff_thresh;    /* 10  */
    float body_diff_gt_thresh; /* 12  */
    int use_gt_term;           /* mode == "train" and not wild (:303-306) */
} kp_reward_cfg;
int kp_sim_term_reward(kp_sim*, const kp_ctx* ctx, const kp_reward_cfg* cfg, float* reward, float* info, uint8_t* fail, float* diffs);

/* The tail of HumanoidAREnv.step in one launch (humanoid_ar_v1.py:288-316): cur_t += 1 (in place; cur_t must be ctx->cur_t), termination and
 * reward as kp_sim_term_reward with the incremented cur_t, then end = cur_t >= min(env_episode_len, ar_context['len']), done = fail || end,
 * percent = cur_t / ar_context['len'].  row_len: int32 [R] = ar_context['len'] of every context row (env e reads row_len[ctx->row[e]]).
 * done / end: uint8 [N]; percent: float [N]; done_count (optional, may be NULL): int32 device counter incremented by the number of done envs.
 * obj7 (optional, may be NULL): float [N,7] <- get_obj_qpos(ar_context['action_one_hot'][0]) of the state after the step (humanoid_ar_v1.py:171-172,
 * 466-477): the simulated pose of the action's first object, data.qpos[76 + action_index_map[a] : +7]; rows whose clip has no action are left
 * alone.  It is the buffer kp_ctx.obj_qpos points to, so the next kp_sim_obs_ar reads the objects where the physics left them. */
int kp_sim_post_step(kp_sim*, const kp_ctx* ctx, const kp_reward_cfg* cfg, int32_t* cur_t, const int32_t* row_len, int env_episode_len,
                     float* reward, float* info, uint8_t* fail, float* diffs, uint8_t* done, uint8_t* end, float* percent, int32_t* done_count, float* obj7);

/* Masked reset in one gather + sim.forward() (mujoco_env.py:86-103, humanoid_ar_v1.py:334-387): for the envs with env_mask != 0 (NULL: all)
 * qpos / qvel <- init_qpos / init_qvel [R, 76] / [R, 75] of context row row[e] (NULL: row e), cur_t[e] = 0 (cur_t may be NULL), warm start
 * zeroed, derived quantities recomputed; set_target != 0: target = qpos_fk(init_qpos) for those envs as reset_model does (:384-386).
 * aux_rows (optional, may be NULL): caller-owned float [N, aux_cols] device rows zeroed for the same envs -- per-episode state that lives
 * outside the simulator, i.e. the kinematic policy's GRU hidden state (PolicyAR.reset / action_rnn.initialize at every episode start,
 * kin_poly/models/policy_ar.py:124-131).
 * row_obj_qpos (optional, may be NULL): float [R,35] = convert_obj_qpos(action_one_hot, obj_pose[0]) of every context row (humanoid_ar_v1.py:377,
 * 479-496); the masked envs' object block is set from their row exactly as kp_sim_set_objects sets it (velocities zero, :382) before sim.forward().
 * row_action_one_hot [R,4] + obj7 [N,7] (optional, both or neither): obj7 <- get_obj_qpos(action_one_hot) of the fresh block for those envs
 * ([0,0,0,1,0,0,0] for a clip without action, :465-466). */
int kp_sim_reset_rows(kp_sim*, const float* init_qpos, const float* init_qvel, const int32_t* row, const uint8_t* env_mask, int32_t* cur_t, int set_target,
                      float* aux_rows, int aux_cols, const float* row_obj_qpos, const float* row_action_one_hot, float* obj7);

/* Episode turnover of a sampler that keeps the NEXT clips of every env resident (the per-episode sample_seq -> init_context -> load_context of
 * sample_worker, kin_poly/core/agent_ar.py:518-535, made ahead of time and batched): the context table holds n_slots rows per env, row = slot * n + env;
 * for every env with done != 0: head <- (head + 1) mod n_slots, ahead <- ahead - 1 (clips still queued behind the current one), row <- head * n + env
 * (the int32 [n] buffer kp_ctx.row points to).  A pure function of its device arrays (no simulator handle); follow with kp_sim_reset_rows(done). */
int kp_pool_advance(int n, int n_slots, const uint8_t* done, int32_t* head, int32_t* ahead, int32_t* row, void* hip_stream);

/* The sampler's per-step record: Memory.push of sample_worker (kin_poly/core/agent_ar.py:582-597; TrajBatchEgo's twelve fields,
 * kin_poly/core/trajbatch_ego.py:5-14) for all n envs in one launch per half-step, into env-major [n, T, .] device buffers at time index t.
 * Every destination may be NULL (field not recorded); a destination needs its source.  bool-like arrays are uint8.
 *   pre  (before the env step): states <- obs [n,105]; episode_start <- fresh [n]; curr_qpos <- qpos [n,76] (get_humanoid_qpos, :571);
 *        gt_target_qpos <- ctx_qpos[row[e], min(cur_t[e] + 1, row_len[row[e]])] (ar_context['qpos'][cur_t + 1], :572; ctx_qpos is [R, ctx_T, 76]);
 *        meta [n,T,2] <- row_meta[row[e]] = (take_ind, fr_start) of the clip the env is on (:627-631).  row may be NULL (row = env).
 *   post (after the env step, before the episode turnover): actions, rewards, fails, dones, percents, c_infos [n,T,6] and, for the full record,
 *        next_states <- obs, res_qpos <- qpos, cc_actions [75], cc_states [784], v_metas [n,T,3] <- (meta[.., t, :], fr_num). */
typedef struct {
    int n, T, t, ctx_T;
    const float* obs; const uint8_t* fresh; const float* qpos; const float* ctx_qpos; const int32_t* row; const int32_t* cur_t; const int32_t* row_len; const float* row_meta;
    float* states; uint8_t* episode_start; float* curr_qpos; float* gt_target_qpos; float* meta;
} kp_record_pre;
typedef struct {
    int n, T, t; float fr_num;
    const float* action; const float* reward; const uint8_t* fail; const uint8_t* done; const float* percent; const float* c_info;
    const float* obs; const float* qpos; const float* cc_action; const float* cc_state; const float* meta;     /* meta: the [n,T,2] buffer `pre` wrote */
    float* actions; float* rewards; uint8_t* fails; uint8_t* dones; float* percents; float* c_infos;
    float* next_states; float* res_qpos; float* cc_actions; float* cc_states; float* v_metas;
} kp_record_post;
int kp_rollout_record_pre(const kp_record_pre*, void* hip_stream);
int kp_rollout_record_post(const kp_record_post*, void* hip_stream);

/* estimate_advantages before normalisation (uhc/khrylib/rl/core/common.py:5-20) on an env-major
 * [N,T] layout (each env's T rows contiguous, time increasing).  All pointers device, float32. */
int kp_gae(int n_envs, int T, const float* rewards, const float* masks, const float* values, float gamma, float tau,
           float* advantages, float* returns, void* hip_stream);
/* same with a bootstrap: last_values [N] = V(state after env e's last row), used where that row's mask is 1 (an episode the fixed
 * horizon of the lock-step sampler cut; the reference's workers always finish their episodes, so its recursion starts from 0).
 * last_values may be NULL (= kp_gae). */
int kp_gae_bootstrap(int n_envs, int T, const float* rewards, const float* masks, const float* values, const float* last_values,
                     float gamma, float tau, float* advantages, float* returns, void* hip_stream);

/* PolicyMCP.forward / select_action after the primitives' and the composer's GEMMs (uhc/core/policy_mcp.py:30-38, uhc/khrylib/rl/core/policy.py:12-15):
 * out [n, A] = sum_k softmax(logits [n, K])_k * prim [K, n, A]  (+ stdv [A] * noise [n, A], rows noise_stride floats apart, when noise != NULL).
 * All device float32; logits are the composer MLP's output BEFORE its softmax. */
int kp_mcp_compose(int n, int K, int A, const float* logits, const float* prim, const float* noise, int noise_stride, const float* stdv, float* out, void* hip_stream);

/* One frame of the kinematic roll-out behind PolicyAR.init_context (TrajARNet.step, kin_poly/models/traj_ar_smpl_net.py:292-330): next_qpos [n,76] =
 * step_ar(qpos, kin_action [n,80]) with the root quaternion normalised, qvel_fd [n,75] = get_qvel_fd_batch(qpos, next_qpos, dt)
 * (kin_poly/utils/torch_utils.py:315-331).  A pure function of its device rows (no simulator handle). */
int kp_kin_advance(int n, const float* qpos, const float* kin_action, float dt, float* next_qpos, float* qvel_fd, void* hip_stream);

/* PolicyMCP's last layer AND its mixing stage in one fp32 MFMA kernel (same reference lines): with h2 [K, n, J] the raw output of the second
 * batched GEMM (no bias, no activation), b2 [K, J], w3 [K, J, ldw >= A] (= nets[k][1].weight^T stacked, rows ldw floats apart; with ldw >= 80,
 * ldw % 4 == 0 and a 16-byte aligned base the kernel reads it with 16-byte loads: pad the rows to 80), b3 [K, A]:
 *   out [n, A] = sum_k softmax(logits [n, K])_k * (b3[k] + relu(h2[k] + b2[k]) w3[k])   (+ stdv [A] * noise [n, A] as above)
 * K <= 16, J a multiple of 64, A <= 80.  All device float32, contiguous except noise (row stride noise_stride). */
int kp_mcp_tail(int n, int K, int J, int A, const float* h2, const float* b2, const float* w3, int ldw, const float* b3, const float* logits, const float* noise,
                int noise_stride, const float* stdv, float* out, void* hip_stream);

/* One roll-out step of torch.nn.GRUCell after its two gate GEMMs (TrajARNet.get_action, kin_poly/models/traj_ar_smpl_net.py:333-343;
 * RNN.forward in step mode, uhc/khrylib/models/rnn.py:24-36): gi [n, 3H] = x W_ih^T and gh [n, 3H] = h_in W_hh^T WITHOUT biases, b_ih / b_hh [3H]
 * -> h_out [n, H] (may alias h_in); xcat (optional) [n, D + H] <- [state | h_out], the row `torch.cat((state, hx), dim=1)` builds for
 * the action MLP (state [n, D], D <= H). */
int kp_gru_cell_step(int n, int H, int D, const float* gi, const float* gh, const float* b_ih, const float* b_hh, const float* h_in, const float* state,
                     float* h_out, float* xcat, void* hip_stream);

/* GRU re-unroll of the PPO / supervised updates (policy_ar.py:104-122, 216-240), one time step of torch.nn.GRUCell
 * semantics over n rows, all arrays contiguous float32 device pointers:
 *   forward : gi [n,3H] = x_t W_ih^T + b_ih, gh [n,3H] = hm_prev W_hh^T + b_hh (the caller's GEMMs), hm_prev [n,H] = previous hidden state
 *             already zeroed at episode starts -> h_out [n,H]; hm_next [n,H] (optional) = h_out * next_keep[row] (next_keep [n] = 1 - episode
 *             start flag of step t + 1, or NULL), the next step's GEMM input.
 *   backward: dh_out [n,H] (gradient reaching h_t from outside the recurrence, may be NULL), carry [n,H] * carry_keep [n] (gradient from
 *             step t + 1, may be NULL) -> dgi [n,3H], dgh [n,3H] (the caller forms carry' = dgh W_hh + dhz and dW_hh += dgh^T hm_prev), dhz [n,H]. */
int kp_gru_gates_forward(int n, int H, const float* gi, const float* gh, const float* hm_prev, const float* next_keep, float* h_out, float* hm_next,
                         void* hip_stream);
int kp_gru_gates_backward(int n, int H, const float* gi, const float* gh, const float* hm_prev, const float* dh_out, const float* carry,
                          const float* carry_keep, float* dgi, float* dgh, float* dhz, void* hip_stream);

/* restore a complete simulator state (what MjSimState + the derived arrays would hold): qpos/qvel and the
 * state (qpos_d/qvel_d) the stale derived quantities belong to; runs the forward pass on the latter. */
int kp_sim_set_full_state(kp_sim*, const float* qpos, const float* qvel, const float* qpos_d, const float* qvel_d, const uint8_t* env_mask);

/* object block of MjSimState save / restore: overwrite data.qpos[76:111] / data.qvel[75:105] of the simulated objects
 * (read them back with KP_OBJ_QPOS / KP_OBJ_QVEL).  Which objects are simulated stays as kp_sim_set_objects decided. */
int kp_sim_set_obj_state(kp_sim*, const float* obj_qpos, const float* obj_qvel, const uint8_t* env_mask);

/* read-outs of the mujoco-py data fields the env uses (humanoid_im.py:342-416, humanoid_ar_v1.py:460-512) */
typedef enum {
    KP_QPOS = 0,        /* data.qpos[:76]                    [N,76]  */
    KP_QVEL = 1,        /* data.qvel[:75]                    [N,75]  */
    KP_XPOS = 2,        /* data.body_xpos[1:25]              [N,72]  (stale by one substep, like mujoco-py) */
    KP_XQUAT = 3,       /* data.body_xquat[1:25]             [N,96]  */
    KP_XIPOS = 4,       /* data.xipos[1:25]                  [N,72]  */
    KP_BQUAT = 5,       /* get_body_quat()                   [N,96]  (humanoid_im.py:342-354, from fresh qpos) */
    KP_HEAD = 6,        /* get_head()                        [N,7]   */
    KP_TARGET_QPOS = 7, /* target['qpos']                    [N,76]  */
    KP_TARGET_WBPOS = 8,   /* target['wbpos']                [N,72]  */
    KP_TARGET_WBQUAT = 9,  /* target['wbquat']               [N,96]  */
    KP_TARGET_BQUAT = 10,  /* target['bquat']                [N,96]  */
    KP_TARGET_COM = 11,    /* target['body_com']             [N,72]  */
    KP_QPOS_D = 12,     /* state the derived quantities were computed at (x_14 after a control step) */
    KP_QVEL_D = 13,
    KP_PREV_BQUAT = 14, /* env.prev_bquat                    [N,96] */
    KP_PREV_HPOS = 15,  /* env.prev_hpos                     [N,7]  */
    KP_OBJ_QPOS = 16,   /* get_obj_qpos() = data.qpos[76:111] [N,35]  (simulated poses of the active objects) */
    KP_OBJ_QVEL = 17,   /* get_obj_qvel() = data.qvel[75:105] [N,30] */
    KP_M = 18,          /* mj_fullM(model, M, data.qM)[:75,:75]   [N,75*75] row-major, armature included (humanoid_im.py:422-425);
                           of the state the derived quantities belong to (KP_QPOS_D), like mujoco-py's data.qM between steps */
    KP_BIAS = 19        /* data.qfrc_bias[:75]               [N,75]  (humanoid_im.py:426), same state */
} kp_field;
int kp_field_dim(int field);
int kp_sim_get(kp_sim*, int field, float* out);
/* Device address of a STORED per-env field ([N, kp_field_dim(field)] float32 rows: KP_QPOS, KP_QVEL, KP_XPOS, KP_XQUAT, KP_XIPOS, KP_TARGET_QPOS,
 * KP_QPOS_D, KP_QVEL_D, KP_OBJ_QPOS, KP_OBJ_QVEL), NULL for derived read-outs.  The zero-copy counterpart of kp_sim_get (`data.qpos` as a view,
 * humanoid_im.py:193): valid for the life of the handle, contents ordered by the handle's stream, overwritten by the next step. */
const float* kp_sim_field_device(kp_sim*, int field);

/* per-env diagnostics of the last kp_sim_step_ctrl: int32 [N,4] = {contacts in last substep,
 * Newton iterations (sum over substeps), flags (bit 0 = non-finite state) | number of substeps whose Newton solve stopped at the
 * iteration cap ("solver_iter", default mjOption.iterations = 100) instead of a termination test << 8,
 * max contacts | Hessian factorisations << 8}.  HOST pointer;
 * synchronises the stream.  Fails if the job queue of the last launch stalled (never observed; see kp_step_queue_kernel). */
int kp_sim_diag(kp_sim*, int32_t* out_host);

/* shader-clock cycles >> 10 every env took inside the last kp_sim_step_ctrl launch, uint32 [N], HOST pointer; synchronises.
 * With model option "lpt_order" on (default: when free objects are simulated) the next launch starts the envs longest-first from these
 * (launch / queue order only: results do not depend on it). */
int kp_sim_launch_cost(kp_sim*, uint32_t* out_host);

/* Floor scenes' job queue: which LDS layout the control-step launches run on.  out3[0] = 1 when the next queue launch would use the lean layout (12 envs per CU,
 * 24 contact slots; model option "lean_queue"), out3[1] = how many times so far the handle has fallen back to the full layout for 64 launches because more than
 * 1 / 64 of the envs needed more contact slots than the lean layout has (their jobs are re-run by a second kernel: cheaper to run everything on the full layout
 * then; model option "lean_adaptive" = 0 keeps the lean layout regardless), out3[2] = control-step launches so far.  Host arithmetic only; no reference
 * counterpart (launch policy: results do not depend on the layout). */
int kp_sim_lean_state(kp_sim*, int32_t* out3);

/* the job sizes kp_sim_step_ctrl uses for a control step of n_substeps when it schedules through the job queue (host arithmetic, no
 * device): sizes16[0 .. return value) sum to n_substeps, the last is substeps_per_job (or absorbs a smaller remainder), every earlier
 * one is `taper` substeps longer than the one after it (the first takes what is left).  Returns the number of jobs (<= 16) or a negative error. */
int kp_job_schedule(int n_substeps, int substeps_per_job, int taper, int* sizes16);

/* seconds the last kp_sim_step_ctrl launch took, measured with HIP events on the sim's stream
 * (synchronises); -1 if none recorded. */
double kp_sim_last_step_seconds(kp_sim*);

/* Launch-duration statistics of kp_sim_step_ctrl: every launch after kp_sim_timing_reset() is bracketed
 * by a HIP event pair on the sim's stream (up to 4096 launches, no host sync while recording).
 * kp_sim_timing_mean_seconds() synchronises once and returns the mean duration (count in *n_launches). */
int kp_sim_timing_reset(kp_sim*);
double kp_sim_timing_mean_seconds(kp_sim*, int* n_launches);

/* Diagnostic: mean shader-clock cycles per environment the last kp_sim_step_ctrl spent in each phase
 * {stable-PD, kinematics+bias, collision, constraint set-up, smooth solve, contact solve, integrate, total}.
 * Collected only if the environment variable KP_PROFILE=1 was set at kp_sim_create.  Synchronises. */
int kp_sim_phase_cycles(kp_sim*, double* out8_host);
/* the same per environment: out_host [N, 8] (the tail of a launch is a few environments, not the mean). */
int kp_sim_phase_cycles_env(kp_sim*, double* out_host);

const char* kp_last_error(void);
const char* kp_version(void);

#ifdef __cplusplus
}
#endif
#endif
