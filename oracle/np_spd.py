"""numpy restatement of the reference's stable-PD controller (TEST INFRASTRUCTURE ONLY).

compute_desired_accel / compute_torque   uhc/envs/humanoid_im.py:418-480
rfc_implicit                             uhc/envs/humanoid_im.py:497-504
Pinned by tests/golden/env_funcs.npz; kp_oracle.c::kpo_compute_torque is checked against this.
"""
import numpy as np
from scipy.linalg import cho_factor, cho_solve

from . import np_oracle as O


def compute_torque_np(qpos, qvel, M, C, ctrl, target_qpos, kpm):
    dt = kpm["opt"][0]
    jkp, jkd, a_scale = kpm["kp"], kpm["kd"], kpm["a_scale"]
    base_pos = np.array(target_qpos[7:], float)
    while np.any(base_pos - qpos[7:] > np.pi):
        base_pos[base_pos - qpos[7:] > np.pi] -= 2 * np.pi
    while np.any(base_pos - qpos[7:] < -np.pi):
        base_pos[base_pos - qpos[7:] < -np.pi] += 2 * np.pi
    target_pos = base_pos + ctrl[:69] * a_scale
    k_p = np.zeros(75); k_d = np.zeros(75)
    k_p[6:] = jkp; k_d[6:] = jkd
    qpos_err = np.concatenate((np.zeros(6), qpos[7:] + qvel[6:] * dt - target_pos))
    qvel_err = np.array(qvel, float)
    q_accel = cho_solve(cho_factor(M + np.diag(k_d) * dt), -C - k_p * qpos_err - k_d * qvel_err)
    qvel_err = qvel_err + q_accel * dt
    return -jkp * qpos_err[6:] - jkd * qvel_err[6:]


def rfc_implicit_np(qpos, vf, kpm):
    vf = np.array(vf, float) * kpm["opt"][17]
    hq = O.get_heading_q(O.remove_base_rot(qpos[3:7]))
    vf[:3] = O.quat_mul_vec(hq, vf[:3])
    return np.clip(vf, -kpm["opt"][18], kpm["opt"][18])
