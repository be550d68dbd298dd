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
void kp_model_free(kp_model*);
/* options: "contact" (0/1), "limits" (0/1), "gravity_z", "stale_kinematics" (0/1, default 1: SPD and
 * read-outs see the one-substep-stale derived quantities mujoco-py exposes), "solver_iter",
 * "solver_tol", "threads_per_env" (64/128/256). */
int kp_model_set_option(kp_model*, const char* name, double value);
double kp_model_get_option(const kp_model*, const char* name);

/* ---- simulator ------------------------------------------------------------------------------------
 * replaces MjSim(model) x N (mujoco_env.py:24) */
kp_sim* kp_sim_create(const kp_model*, int n_envs, int device_id, void* hip_stream);
void kp_sim_destroy(kp_sim*);
int kp_sim_n_envs(const kp_sim*);

/* sim.reset() + set_state(qpos, qvel) + sim.forward()   (mujoco_env.py:86-103) for the masked envs.
 * qpos [N,76], qvel [N,75]. */
int kp_sim_set_state(kp_sim*, const float* qpos, const float* qvel, const uint8_t* env_mask);

/* self.target = smpl_humanoid.qpos_fk(target_qpos)   (humanoid_ar_v1.py:256, numpy_smpl_humanoid.py:180)
 * target_qpos [N,76] is copied; the target dict {qpos, wbpos, wbquat, bquat, body_com} is kept on device. */
int kp_sim_set_target(kp_sim*, const float* target_qpos, const uint8_t* env_mask);

/* HumanoidEnv.do_simulation(cc_action, n_substeps)   (humanoid_im.py:506-533): per substep stable-PD
 * torque (compute_torque :433-480), clip, rfc_implicit (:497-504), sim.step() (:527).  cc_action [N,75]. */
int kp_sim_step_ctrl(kp_sim*, const float* cc_action, int n_substeps, const uint8_t* env_mask);

/* HumanoidAREnv.step_ar(a)   (humanoid_ar_v1.py:216-241): kin_action [N,80] -> next_qpos [N,76] */
int kp_sim_step_kin(kp_sim*, const float* kin_action, float* next_qpos);

/* get_full_obs_v1() [N,784]   (humanoid_im.py:144-233), optional ZFilter(update=False) + clip
 * (zfilter.py:58-67): pass mean/std [784] device pointers or NULL, clip <= 0 disables clipping. */
int kp_sim_obs_cc(kp_sim*, float* out, const float* zf_mean, const float* zf_std, float clip);

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
    KP_QVEL_D = 13
} kp_field;
int kp_field_dim(int field);
int kp_sim_get(kp_sim*, int field, float* out);

/* per-env diagnostics of the last kp_sim_step_ctrl: int32 [N,4] = {contacts in last substep,
 * Newton iterations (sum over substeps), flags (1 = non-finite state), max contacts}.  HOST pointer;
 * synchronises the stream. */
int kp_sim_diag(kp_sim*, int32_t* out_host);

/* seconds the last kp_sim_step_ctrl launch took, measured with HIP events on the sim's stream
 * (synchronises); -1 if none recorded. */
double kp_sim_last_step_seconds(kp_sim*);

const char* kp_last_error(void);
const char* kp_version(void);

#ifdef __cplusplus
}
#endif
#endif
