"""Generate tests/golden/*.npz by IMPORTING the reference's Python (this container only).

The reference tree (/root/reference) is imported with stub modules for its GUI / MuJoCo / IO
dependencies; its pure-Python functions on the rollout path are then run
on seeded inputs and (inputs, outputs) are written as small .npz fixtures.  Only data is
committed -- no reference source.  MuJoCo-side inputs (body_xpos, qM, qfrc_bias, ...) are supplied
from this repo's fp64 oracle so that the fixtures are physically plausible; for the functions under
test they are just inputs.

Run from a scratch cwd (reference Config classes mkdir under cwd):
    cd /tmp && python /root/repo/tools/make_golden.py
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True


class Pkg(MagicMock):
    __path__ = []


for m in ['cv2', 'OpenGL', 'OpenGL.GL', 'gym', 'gym.envs', 'gym.envs.mujoco', 'gym.envs.mujoco.mujoco_env', 'gym.utils',
          'gym.spaces', 'glfw', 'torchvision', 'torchvision.models', 'torchvision.transforms', 'skimage', 'skimage.util',
          'skimage.util.shape', 'mujoco_py', 'mujoco_py.builder', 'mujoco_py.generated', 'mujoco_py.generated.const',
          'mujoco_py.utils', 'mujoco_py.functions', 'wandb', 'lxml', 'lxml.etree', 'ipdb', 'torchgeometry', 'smplx', 'imageio',
          'kin_poly.envs.humanoid_v2', 'kin_poly.data_loaders.statereg_dataset', 'kin_poly.utils.torch_humanoid']:   # modules the reference imports but does not ship
    sys.modules[m] = Pkg()
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import torch  # noqa: E402

torch.set_default_dtype(torch.float64)

from uhc.khrylib.utils.math import (de_heading, get_heading, get_heading_q, quat_from_expmap, quat_mul_vec,  # noqa: E402
                                     transform_vec, transform_vec_batch, multi_quat_diff, get_angvel_fd)
from uhc.khrylib.utils.transformation import (quaternion_from_euler, quaternion_inverse, quaternion_matrix,  # noqa: E402
                                               quaternion_multiply, rotation_from_quaternion)
from uhc.khrylib.utils.zfilter import ZFilter  # noqa: E402
from uhc.khrylib.rl.core.common import estimate_advantages  # noqa: E402
import uhc.envs.humanoid_im as him  # noqa: E402
import kin_poly.envs.humanoid_ar_v1 as har  # noqa: E402
import kin_poly.utils.numpy_smpl_humanoid as nsh  # noqa: E402
from kin_poly.utils.math_utils import multi_quat_norm_v2  # noqa: E402
from kin_poly.core.reward_function import dynamic_supervision_v1  # noqa: E402
from uhc.core.policy_mcp import PolicyMCP  # noqa: E402
from uhc.khrylib.rl.core.critic import Value  # noqa: E402
from uhc.khrylib.models.mlp import MLP  # noqa: E402

from kinpoly_amd.model_compiler import read_kpm  # noqa: E402
from oracle.kpo import OracleSim  # noqa: E402

KPM = read_kpm(os.path.join(REPO, "kinpoly_amd", "assets", "smpl_humanoid.kpm"))
STD = np.load(os.path.join(OUT, "standing_neutral.npz"))
NB = 24
NAMES = ["Pelvis", "L_Hip", "L_Knee", "L_Ankle", "L_Toe", "R_Hip", "R_Knee", "R_Ankle", "R_Toe", "Torso", "Spine", "Chest",
         "Neck", "Head", "L_Thorax", "L_Shoulder", "L_Elbow", "L_Wrist", "L_Hand", "R_Thorax", "R_Shoulder", "R_Elbow",
         "R_Wrist", "R_Hand"]


def fake_mj_model():
    m = types.SimpleNamespace()
    m.body_pos = np.vstack([np.zeros((1, 3)), KPM["body_pos"].reshape(NB, 3)])
    m.body_ipos = np.vstack([np.zeros((1, 3)), KPM["body_ipos"].reshape(NB, 3)])
    m.body_parentid = np.concatenate([[0], KPM["body_parent"] + 1])
    m.body_names = ["world"] + NAMES
    m._body_name2id = {n: i for i, n in enumerate(m.body_names)}
    m.nv = 75
    m.opt = types.SimpleNamespace(timestep=KPM["opt"][0])
    return m


def rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def rand_qpos(rng, scale=0.4):
    q = STD["qpos"].copy()
    q[:2] += rng.normal(size=2) * 0.5
    q[2] += rng.normal() * 0.1
    q[3:7] = rand_quat(rng) if rng.random() < 0.5 else quaternion_multiply(q[3:7], quat_from_expmap(rng.normal(size=3) * 0.3))
    q[7:] += rng.normal(size=69) * scale
    return q


def gen_quat_utils():
    rng = np.random.default_rng(100)
    n = 64
    q0 = np.stack([rand_quat(rng) for _ in range(n)]); q1 = np.stack([rand_quat(rng) * rng.uniform(0.5, 2.0) for _ in range(n)])
    v = rng.normal(size=(n, 3)); e = rng.normal(size=(n, 3)); eul = rng.uniform(-np.pi, np.pi, size=(n, 3))
    # literal inputs of the reference's own self-check (kin_poly/utils/torch_utils.py:446-460)
    q1[0] = [1.522, 0.560, 0.161, 0.623]; v[0] = [1.2363, 4.41412, 7.2432]
    out = dict(q0=q0, q1=q1, v=v, e=e, eul=eul)
    out["mul"] = np.stack([quaternion_multiply(a, b) for a, b in zip(q1, q0)])
    out["inv"] = np.stack([quaternion_inverse(a) for a in q1])
    out["mat"] = np.stack([quaternion_matrix(a)[:3, :3] for a in q1])
    out["qmulvec"] = np.stack([quat_mul_vec(a, b) for a, b in zip(q1, v)])
    out["tv_root"] = np.stack([transform_vec(b, a, 'root') for a, b in zip(q1, v)])
    out["tv_heading"] = np.stack([transform_vec(b, a, 'heading') for a, b in zip(q1, v)])
    out["heading_q"] = np.stack([get_heading_q(a) for a in q1])
    out["heading"] = np.array([get_heading(a) for a in q1])
    out["de_heading"] = np.stack([de_heading(a) for a in q1])
    out["expmap"] = np.stack([quat_from_expmap(a) for a in e])
    out["euler_rzyx"] = np.stack([quaternion_from_euler(a[0], a[1], a[2], 'rzyx') for a in eul])
    out["euler_sxyz"] = np.stack([quaternion_from_euler(a[0], a[1], a[2]) for a in eul])
    out["rot_from_quat"] = np.stack([rotation_from_quaternion(a) for a in q0])
    out["quat_norm_v2"] = multi_quat_norm_v2(q0.reshape(-1))
    np.savez(os.path.join(OUT, "quat_utils.npz"), **out)


def make_humanoid():
    nsh.load_model_from_path = lambda f: fake_mj_model()
    return nsh.Humanoid(model_file="unused.xml")


def gen_fk(hum):
    rng = np.random.default_rng(101)
    qs = [STD["qpos"].copy()] + [rand_qpos(rng, 0.6) for _ in range(32)]
    qs[3][3:7] *= 1.7  # un-normalised root quaternion (step_ar emits those)
    res = [hum.qpos_fk(q.copy()) for q in qs]
    np.savez(os.path.join(OUT, "fk.npz"), qpos_in=np.stack(qs), qpos=np.stack([r["qpos"] for r in res]),
             wbpos=np.stack([r["wbpos"] for r in res]), wbquat=np.stack([r["wbquat"] for r in res]),
             bquat=np.stack([r["bquat"] for r in res]), body_com=np.stack([r["body_com"] for r in res]))


class FakeData:
    pass


def oracle_data(qpos, qvel, stale_steps=1):
    """data.* as mujoco-py exposes them after a control step: derived arrays one substep stale."""
    o = OracleSim()
    o.reset(qpos, qvel)
    for _ in range(stale_steps):
        o.step()
    d = FakeData()
    d.qpos = np.concatenate([o.get("qpos"), np.zeros(35)])
    d.qvel = np.concatenate([o.get("qvel"), np.zeros(30)])
    d.body_xpos = np.vstack([np.zeros((1, 3)), o.get("xpos").reshape(NB, 3), np.zeros((5, 3))])
    d.body_xquat = np.vstack([[[1, 0, 0, 0]], o.get("xquat").reshape(NB, 4), np.tile([1., 0, 0, 0], (5, 1))])
    d.xipos = np.vstack([np.zeros((1, 3)), o.get("xipos").reshape(NB, 3), np.zeros((5, 3))])
    d.qfrc_bias = np.concatenate([o.get("qfrc_bias"), np.zeros(30)])
    d.qfrc_applied = np.zeros(105)
    d.ctrl = np.zeros(69)
    d.qM = None
    d._M = o.fullM()
    d.get_body_xipos = lambda name: d.xipos[["world"] + NAMES == name] if False else d.xipos[(["world"] + NAMES).index(name)]
    return d


def make_env(cls):
    env = cls.__new__(cls)
    env.model = fake_mj_model()
    env.qpos_lim, env.qvel_lim, env.body_lim = 76, 75, 25
    env.base_rot = [0.7071, 0.7071, 0.0, 0.0]
    env.no_root = False
    env.sim_iter = 15
    env.rfc_rate = 1
    env.ndof, env.vf_dim, env.meta_pd_dim = 69, 6, 0
    cc = types.SimpleNamespace(obs_coord='root', obs_vel='full', action_v=1, meta_pd=False, meta_pd_joint=False,
                               a_scale=KPM["a_scale"].copy(), jkp=KPM["kp"].copy(), jkd=KPM["kd"].copy(),
                               torque_lim=KPM["torque_lim"].copy(), residual_force_scale=100.0, residual_force_lim=100.0,
                               residual_force=True, residual_force_mode='implicit', action_type='position', obs_v=1,
                               env_term_body='body', env_episode_len=100000)
    env.cc_cfg = cc
    env.body_qposaddr = {n: (7 + 3 * (i - 1), 10 + 3 * (i - 1)) for i, n in enumerate(NAMES) if i > 0}
    env.jpos_diffw = np.ones((24, 1))      # reward_weights.get("jpos_diffw", ones): no config sets it (humanoid_im.py:28, humanoid_ar_v1.py:59)
    return env


def patch_fullM():
    def mj_fullM(model, M, qM_unused):
        M[:] = 0
        full = np.zeros((model.nv, model.nv))
        full[:75, :75] = patch_fullM.M
        M[:] = full.ravel()
    him.mjf.mj_fullM = mj_fullM


def gen_env_fixtures(hum):
    rng = np.random.default_rng(102)
    n = 24
    rec = {k: [] for k in ["qpos", "qvel", "xpos", "xquat", "xipos", "target_qpos", "obs_cc", "bquat", "M", "bias", "ctrl",
                           "torque", "rfc", "kin_action", "next_qpos", "body_diff", "head"]}
    env = make_env(har.HumanoidAREnv)
    env.smpl_humanoid = hum
    env.pose_delta = False
    patch_fullM()
    for i in range(n):
        q0 = rand_qpos(rng, 0.3); v0 = rng.normal(size=75) * 0.5
        d = oracle_data(q0, v0)
        if i == 5:
            d.body_xquat[1, 0] = 0.0  # exercises the cur_quat[0,0]==0 fallback (humanoid_im.py:219-220)
        env.data = d
        env.model.nv = 75
        tq = rand_qpos(rng, 0.3)
        if i % 3 == 0:
            tq[7:] += 2 * np.pi * rng.integers(-1, 2, size=69)  # exercises the 2*pi unwrap loops (:442-445)
        env.target = hum.qpos_fk(tq.copy())
        rec["qpos"].append(d.qpos[:76].copy()); rec["qvel"].append(d.qvel[:75].copy())
        rec["xpos"].append(d.body_xpos[1:25].copy()); rec["xquat"].append(d.body_xquat[1:25].copy()); rec["xipos"].append(d.xipos[1:25].copy())
        rec["target_qpos"].append(env.target["qpos"].copy())
        rec["obs_cc"].append(env.get_full_obs_v1())
        rec["bquat"].append(env.get_body_quat())
        rec["head"].append(env.get_head())
        rec["body_diff"].append(env.calc_body_diff())
        # SPD torque + RFC with this M / bias
        patch_fullM.M = d._M
        ctrl = rng.normal(size=75) * 0.5
        env.model.nv = 75
        d.qfrc_bias = d.qfrc_bias[:75]
        torque = env.compute_torque(ctrl.copy(), i_iter=0)
        d.qfrc_applied = np.zeros(75)
        env.rfc_implicit(ctrl[69:75].copy())
        rec["M"].append(d._M); rec["bias"].append(d.qfrc_bias[:75].copy()); rec["ctrl"].append(ctrl)
        rec["torque"].append(torque); rec["rfc"].append(d.qfrc_applied[:6].copy())
        # step_ar
        a = rng.normal(size=80) * 0.5
        a[1:5] = rand_quat(rng)
        rec["kin_action"].append(a); rec["next_qpos"].append(env.step_ar(a.copy()))
    np.savez(os.path.join(OUT, "env_funcs.npz"), **{k: np.stack(v) for k, v in rec.items()})


def gen_ar_obs_reward(hum):
    rng = np.random.default_rng(103)
    n, T = 12, 6
    env = make_env(har.HumanoidAREnv)
    env.smpl_humanoid = hum
    env.kin_cfg = types.SimpleNamespace(use_context=False, use_of=False, use_head=True, use_vel=False, use_obj=True, use_action=True,
                                        policy_specs={'reward_weights': dict(w_hp=0.15, w_hq=0.15, w_p=0.2, w_jp=0.2, w_act_p=0.2, w_act_v=0.1,
                                                                            k_hp=45, k_hq=45, k_p=50, k_jp=50, k_act_p=5, k_act_v=0.005)})
    env.ar_model_v = 1
    env.policy_v = 1
    env.action_index_map = [0, 7, 21, 28]; env.action_len = [7, 14, 7, 7]
    env.frame_skip = 15  # dt is a property: model.opt.timestep * frame_skip
    env.end_reward = 0.0
    rec = {k: [] for k in ["qpos", "qvel", "xpos", "xquat", "xipos", "t", "head_pose", "head_vels", "obj_rel", "action_one_hot",
                           "obs_ar", "target_qpos", "prev_bquat", "prev_hpos", "gt_bquat", "gt_prev_bquat", "gt_wbpos", "reward",
                           "reward_info", "body_diff", "body_gt_diff", "obj_qpos"]}
    for i in range(n):
        q0 = rand_qpos(rng, 0.2); v0 = rng.normal(size=75) * 0.3
        d = oracle_data(q0, v0)
        one_hot = np.zeros(4)
        if i % 4 == 1:
            one_hot[0] = 1.0
            d.qpos[76:83] = np.concatenate([rng.normal(size=3), rand_quat(rng)])
        env.data = d
        t = int(rng.integers(1, T - 1))
        env.cur_t = t
        gt_qpos = np.stack([rand_qpos(rng, 0.2) for _ in range(T)])
        gt = hum.qpos_fk_batch(gt_qpos)
        ctx = dict(action_one_hot=np.tile(one_hot, (T, 1)), head_pose=np.concatenate([rng.normal(size=(T, 3)), np.stack([rand_quat(rng) for _ in range(T)])], 1),
                   head_vels=rng.normal(size=(T, 6)), obj_head_relative_poses=rng.normal(size=(T, 7)), qpos=gt_qpos, bquat=gt["bquat"].reshape(T, -1))
        env.ar_context = ctx
        env.gt_targets = gt
        tq = rand_qpos(rng, 0.2)
        env.target = hum.qpos_fk(tq.copy())
        env.prev_bquat = hum.qpos_fk(rand_qpos(rng, 0.2))["bquat"].reshape(-1)
        env.prev_hpos = np.concatenate([d.body_xpos[14] + rng.normal(size=3) * 0.01, rand_quat(rng)])
        rec["qpos"].append(d.qpos[:76].copy()); rec["qvel"].append(d.qvel[:75].copy()); rec["obj_qpos"].append(d.qpos[76:111].copy())
        rec["xpos"].append(d.body_xpos[1:25].copy()); rec["xquat"].append(d.body_xquat[1:25].copy()); rec["xipos"].append(d.xipos[1:25].copy())
        rec["t"].append(t); rec["head_pose"].append(ctx["head_pose"][t]); rec["head_vels"].append(ctx["head_vels"][t])
        rec["obj_rel"].append(ctx["obj_head_relative_poses"][t]); rec["action_one_hot"].append(one_hot)
        rec["obs_ar"].append(env.get_ar_obs_v1())
        rec["target_qpos"].append(env.target["qpos"].copy()); rec["prev_bquat"].append(env.prev_bquat.copy()); rec["prev_hpos"].append(env.prev_hpos.copy())
        rec["gt_bquat"].append(ctx["bquat"][t]); rec["gt_prev_bquat"].append(ctx["bquat"][t - 1]); rec["gt_wbpos"].append(gt["wbpos"][t])
        r, info = dynamic_supervision_v1(env, None, None, {"end": False})
        rec["reward"].append(r); rec["reward_info"].append(info)
        rec["body_diff"].append(env.calc_body_diff()); rec["body_gt_diff"].append(env.calc_body_gt_diff())
    np.savez(os.path.join(OUT, "ar_obs_reward.npz"), **{k: np.stack(v) for k, v in rec.items()})


def gen_gae_zfilter():
    rng = np.random.default_rng(104)
    B = 257
    rewards = rng.uniform(0, 1, size=(B, 1)); values = rng.normal(size=(B, 1))
    masks = np.ones((B, 1)); masks[rng.choice(B, 9, replace=False)] = 0; masks[-1] = 0
    adv, ret = estimate_advantages(torch.tensor(rewards), torch.tensor(masks), torch.tensor(values), 0.95, 0.95)
    zf = ZFilter((784,), clip=5)
    xs = rng.normal(size=(50, 784)) * rng.uniform(0.1, 3, size=784) + rng.normal(size=784)
    for x in xs:
        zf(x)
    x = rng.normal(size=784) * 4
    np.savez(os.path.join(OUT, "gae_zfilter.npz"), rewards=rewards, values=values, masks=masks, adv=adv.numpy(), ret=ret.numpy(),
             zf_mean=zf.rs.mean, zf_std=zf.rs.std, zf_x=x, zf_y=zf(x, update=False))


def seeded_state_dict(module, seed):
    rng = np.random.default_rng(seed)
    sd = {}
    for k, v in module.state_dict().items():
        fan_in = v.shape[-1] if v.dim() > 1 else v.shape[0]
        sd[k] = torch.tensor(rng.standard_normal(tuple(v.shape)) / np.sqrt(max(fan_in, 1)))
    return sd


def gen_policies():
    cfg = types.SimpleNamespace(policy_hsize=[512, 256], policy_htype='relu', fix_std=True, log_std=-2.3, num_primitive=8)
    cfg.get = lambda k, dflt=None: getattr(cfg, k, dflt)
    pol = PolicyMCP(cfg, action_dim=75, state_dim=784)
    pol.load_state_dict(seeded_state_dict(pol, 7))
    val = Value(MLP(105, [512, 256], 'relu'))
    val.load_state_dict(seeded_state_dict(val, 8))
    rng = np.random.default_rng(105)
    x = np.clip(rng.normal(size=(16, 784)) * 1.5, -5, 5); s = rng.normal(size=(16, 105))
    with torch.no_grad():
        mean = pol.forward(torch.tensor(x)).loc.numpy()
        w = pol.composer(torch.tensor(x)).numpy()
        v = val(torch.tensor(s)).numpy()
    np.savez(os.path.join(OUT, "policies.npz"), mcp_seed=7, value_seed=8, x=x, mcp_mean=mean, mcp_weights=w, s=s, value=v,
             mcp_keys=np.array(list(pol.state_dict().keys())), value_keys=np.array(list(val.state_dict().keys())))


def gen_traj_ar_net(hum):
    """TrajARNet.init_states / forward (kin_poly/models/traj_ar_smpl_net.py:180-201, 346-383): context GRU -> init state,
    then the full kinematic roll-out of a short clip with seeded weights (weights are regenerated from the seed in the test)."""
    import kin_poly.models.traj_ar_smpl_net as tn
    import kin_poly.utils.torch_smpl_humanoid as tsh
    tsh.load_model_from_path = lambda f: fake_mj_model()
    cfg = types.SimpleNamespace(model_specs=dict(model_v=1, rnn_hdim=1024, mlp_hsize=[1024, 512, 256], mlp_htype="relu", rnn_type="gru"),
                                mujoco_model_file="unused.xml", use_of=False, use_head=True, use_action=True, use_vel=False, use_context=False,
                                add_noise=False, noise_std=0.01, has_z=True, data_dir=os.path.join(REF, "sample_data"))
    rng = np.random.default_rng(106)
    B, T = 3, 5
    qpos = np.stack([[rand_qpos(rng, 0.15) for _ in range(T)] for _ in range(B)])
    data = dict(qpos=qpos, qvel=rng.normal(size=(B, T, 75)) * 0.1, target=rng.normal(size=(B, T, 80)) * 0.1,
                head_pose=np.concatenate([rng.normal(size=(B, T, 3)), np.stack([[rand_quat(rng) for _ in range(T)] for _ in range(B)])], 2),
                head_vels=rng.normal(size=(B, T, 6)) * 0.3, obj_head_relative_poses=rng.normal(size=(B, T, 7)) * 0.3,
                obj_pose=np.tile(np.array([0.3, 0.2, 0.1, 1.0, 0, 0, 0]), (B, T, 1)), action_one_hot=np.tile(np.array([0, 0, 1.0, 0]), (B, T, 1)))
    data_t = {k: torch.tensor(v) for k, v in data.items()}
    net = tn.TrajARNet(cfg, data_sample=data_t, device=torch.device("cpu"), dtype=torch.float64, mode="test", as_policy=True)
    sd = seeded_state_dict(net, 9)
    # small output layers keep the roll-out in a sane pose range
    for k in sd:
        if k.startswith(("action_fc", "context_fc")):
            sd[k] = sd[k] * 0.05
    net.load_state_dict(sd)
    net.set_schedule_sampling(0.0)
    with torch.no_grad():
        d1 = net.init_states({k: v.clone() for k, v in data_t.items()})
        init_qpos, init_qvel, ctx_feat = d1["init_qpos"].numpy(), d1["init_qvel"].numpy(), d1["context_feat_rnn"].numpy()
        fp = net.forward({k: v.clone() for k, v in data_t.items()})
    np.savez(os.path.join(OUT, "traj_ar_net.npz"), seed=9, keys=np.array(list(net.state_dict().keys())),
             shapes=np.array([list(v.shape) + [0] * (2 - v.dim()) for v in net.state_dict().values()]),
             **{"in_" + k: v for k, v in data.items()}, init_qpos=init_qpos, init_qvel=init_qvel, context_feat_rnn=ctx_feat,
             ar_qpos=fp["qpos"].numpy(), ar_qvel=fp["qvel"].numpy(), action=fp["action"].numpy(), state_dim=net.state_dim, context_dim=net.context_dim)
    # PolicyAR.initialize_rnn + forward(mode="train") (policy_ar.py:104-122, 216-240): the padded [T_max, n_episodes] re-unroll of the
    # GRU over a flat batch whose episodes are cut by masks == 0, run on a duck-typed `self` that holds the same seeded TrajARNet
    import kin_poly.models.policy_ar as par
    nb = 14
    masks = torch.ones(nb); masks[[2, 6, 7, 13]] = 0          # episodes of 3, 4, 1 and 6 rows
    v_metas = torch.zeros((nb, 3))
    states = torch.tensor(rng.normal(size=(nb, net.state_dim)) * 0.5)
    stub = types.SimpleNamespace(traj_ar_net=net, state_dim=net.state_dim, action_dim=80, policy_v=1, mode="train",
                                 action_log_std=torch.ones(1, 80) * -3.2)
    stub.get_action = lambda st: par.PolicyAR.get_action(stub, st)
    par.PolicyAR.initialize_rnn(stub, (masks, v_metas))
    with torch.no_grad():
        _, mean, _ = par.PolicyAR.forward(stub, states)
    np.savez(os.path.join(OUT, "unroll.npz"), seed=9, states=states.numpy(), masks=masks.numpy(), action_mean=mean.numpy(),
             num_episode=stub.num_episode, max_episode_len=stub.max_episode_len)
    from scipy.ndimage import gaussian_filter1d
    x = rng.normal(size=(12, 69))
    np.savez(os.path.join(OUT, "smooth.npz"), x=x, y=gaussian_filter1d(x, 1, axis=0))
    # the statement PolicyAR.init_context really executes (policy_ar.py:150-152) on its [1, T, 76] roll-out tensor
    ar_qpos = torch.tensor(rng.normal(size=(1, 20, 76)))
    x_eff = ar_qpos.numpy().copy()
    ar_qpos[:, 7:] = torch.from_numpy(gaussian_filter1d(ar_qpos[:, 7:].cpu(), 1, axis=0))
    np.savez(os.path.join(OUT, "smooth_effective.npz"), x=x_eff, y=ar_qpos.numpy())


def gen_pretrain(hum):
    """The warm start of AgentAR.train_init (agent_ar.py:366-385): TrajARNet.forward in train form (traj_ar_smpl_net.py:346-383; the differentiable
    kinematic roll-out incl. scheduled sampling), compute_loss (:390-457), compute_loss_init (:499-527) and the gradients both losses leave on a few
    small parameters, with seeded weights and kin_poly.yml's model_specs weights.  add_noise is off here (the observation noise is random)."""
    import kin_poly.models.traj_ar_smpl_net as tn
    import kin_poly.utils.torch_smpl_humanoid as tsh
    tsh.load_model_from_path = lambda f: fake_mj_model()
    cfg = types.SimpleNamespace(model_specs=dict(model_v=1, rnn_hdim=1024, mlp_hsize=[1024, 512, 256], mlp_htype="relu", rnn_type="gru",
                                                 w_rp=50.0, w_rr=50.0, w_p=1.0, w_v=1.0, w_ee=10.0, w_op=1.0, w_or=10.0),
                                mujoco_model_file="unused.xml", use_of=False, use_head=True, use_action=True, use_vel=False, use_context=False,
                                add_noise=False, noise_std=0.01, has_z=True, data_dir=os.path.join(REF, "sample_data"))
    rng = np.random.default_rng(206)
    B, T = 3, 6
    qpos = np.stack([[rand_qpos(rng, 0.15) for _ in range(T)] for _ in range(B)])
    data = dict(qpos=qpos, qvel=rng.normal(size=(B, T, 75)) * 0.1, target=rng.normal(size=(B, T, 80)) * 0.1,
                head_pose=np.concatenate([rng.normal(size=(B, T, 3)), np.stack([[rand_quat(rng) for _ in range(T)] for _ in range(B)])], 2),
                head_vels=rng.normal(size=(B, T, 6)) * 0.3, obj_head_relative_poses=rng.normal(size=(B, T, 7)) * 0.3,
                obj_pose=np.concatenate([rng.normal(size=(B, T, 3)), np.stack([[rand_quat(rng) for _ in range(T)] for _ in range(B)])], 2),
                action_one_hot=np.tile(np.array([0, 1.0, 0, 0]), (B, T, 1)), wbpos=rng.normal(size=(B, T, 72)))
    data_t = {k: torch.tensor(v) for k, v in data.items()}
    net = tn.TrajARNet(cfg, data_sample=data_t, device=torch.device("cpu"), dtype=torch.float64, mode="train", as_policy=True)
    sd = seeded_state_dict(net, 19)
    for k in sd:
        if k.startswith(("action_fc", "context_fc")):
            sd[k] = sd[k] * 0.05
    net.load_state_dict(sd)
    out = {}
    watch = ("action_fc.bias", "context_fc.bias", "action_mlp.affine_layers.2.bias", "context_mlp.affine_layers.0.bias")
    params = dict(net.named_parameters())
    coins = [0, 0, 1, 0, 1, 0]           # scheduled sampling's draws (np.random.binomial(1, gt_rate) at the initial state and after every step), scripted
    out["coins"] = np.array(coins)
    for tag, rate in (("", 0.0), ("_gt", 0.3)):
        net.set_schedule_sampling(rate)
        net.zero_grad()
        seq, orig = iter(coins), np.random.binomial
        np.random.binomial = lambda n, p: next(seq)
        try:
            fp = net.forward({k: v.clone() for k, v in data_t.items()})
        finally:
            np.random.binomial = orig
        loss, idv = net.compute_loss(fp, data_t)
        loss.backward()
        out.update({f"qpos{tag}": fp["qpos"].detach().numpy(), f"qvel{tag}": fp["qvel"].detach().numpy(), f"action{tag}": fp["action"].detach().numpy(),
                    f"obj_2_head{tag}": fp["obj_2_head"].detach().numpy(), f"pred_wbpos{tag}": fp["pred_wbpos"].detach().numpy(),
                    f"loss{tag}": float(loss), f"loss_idv{tag}": np.array(idv)})
        for w in watch:
            out[f"grad{tag}:{w}"] = params[w].grad.numpy().copy()
    net.set_schedule_sampling(0.0)
    net.zero_grad()
    d1 = net.init_states({k: v.clone() for k, v in data_t.items()})
    loss_i, idv_i = net.compute_loss_init(d1["init_qpos"], data_t["qpos"][:, 0], d1["init_qvel"], data_t["qvel"][:, 0])
    loss_i.backward()
    out.update(loss_init=float(loss_i), loss_init_idv=np.array(idv_i))
    for w in ("context_fc.bias", "context_mlp.affine_layers.0.bias"):
        out[f"grad_init:{w}"] = params[w].grad.numpy().copy()
    np.savez(os.path.join(OUT, "pretrain.npz"), seed=19, keys=np.array(list(net.state_dict().keys())),
             shapes=np.array([list(v.shape) + [0] * (2 - v.dim()) for v in net.state_dict().values()]),
             state_dim=net.state_dim, context_dim=net.context_dim, **{"in_" + k: v for k, v in data.items()}, **out)


def gen_metrics():
    """The kinematic paper metrics (kin_poly/utils/metrics.py + compute_error_accel / the mpjpe lines of scripts/eval_pose_all.py:45-74, 140-172) on a
    seeded pair of pose sequences; joint positions are inputs (MuJoCo's body_xpos in the reference)."""
    import importlib
    import kin_poly.utils.metrics as M
    rng = np.random.default_rng(321)
    T = 12
    def seq():
        q = np.stack([rand_qpos(rng, 0.2) for _ in range(T)])
        q[1:, :3] = q[0, :3] + np.cumsum(rng.normal(size=(T - 1, 3)) * 0.01, 0)
        base = q[0, 3:7].copy()
        for t in range(1, T):                      # small frame-to-frame root rotations (incl. one exactly repeated frame: the 1e-6 branch)
            d = quaternion_from_euler(*(rng.normal(size=3) * 0.05))
            q[t, 3:7] = quaternion_multiply(d, q[t - 1, 3:7]) if t != 5 else q[t - 1, 3:7]
        return q
    pred, gt = seq(), seq()
    jp, jg = rng.normal(size=(T, 24, 3)), rng.normal(size=(T, 24, 3))
    hp = np.concatenate([rng.normal(size=(T, 3)), np.stack([rand_quat(rng) for _ in range(T)])], 1)
    hg = np.concatenate([rng.normal(size=(T, 3)), np.stack([rand_quat(rng) for _ in range(T)])], 1)
    dt = 1 / 30
    vp, vg = M.get_joint_vels(pred, dt), M.get_joint_vels(gt, dt)
    src = open(os.path.join(REF, "scripts", "eval_pose_all.py")).read()
    ns = {"np": np}
    exec(src[src.index("def compute_error_accel"):src.index("def compute_vel")], ns)          # the function's own text, run here, nothing of it is stored
    accel = np.mean(ns["compute_error_accel"](jp.copy(), jg.copy())) * 1000
    a, b = jp - jp[:, 0:1], jg - jg[:, 0:1]
    np.savez(os.path.join(OUT, "metrics.npz"), pred=pred, gt=gt, jpos_pred=jp, jpos_gt=jg, head_pred=hp, head_gt=hg, dt=dt, vels_pred=vp, vels_gt=vg,
             root_dist=M.get_frobenious_norm(M.get_root_matrix(pred), M.get_root_matrix(gt)), head_dist=M.get_frobenious_norm(M.get_root_matrix(hp), M.get_root_matrix(hg)),
             vel_dist=M.get_mean_dist(vp, vg), accel_dist=accel, mpjpe=np.linalg.norm(a - b, axis=2).mean() * 1000,
             accels_abs=M.get_mean_abs(M.get_joint_accels(vp, dt)))


def gen_update_params(hum):
    """AgentAR.optimize_policy's update half, run by the reference itself (kin_poly/core/agent_ar.py:264-269 per_epoch_update, :682-752 update_params,
    :756-772 update_policy, :852-870 ppo_loss / update_value; uhc/khrylib/rl/agents/agent_ppo.py:53-56 clip_policy_grad;
    kin_poly/models/policy_ar.py:72-89 optimiser + step_lr, :104-122 initialize_rnn, :216-240 forward(train), :277-287 update_supervised_step;
    uhc/khrylib/rl/core/common.py:5-25 estimate_advantages) for TWO consecutive iterations in fp64 on a recorded batch, with kin_poly.yml's
    update switches and rates (rl_update + step_update, 10 PPO epochs, 20 step updates, policy_lr 1e-5, value_lr 3e-4, lr 5e-4, clip 40, eps 0.2).

    A real PolicyAR (two TrajARNets, its own Adam + LambdaLR) and Value are built; AgentAR is built by __new__ with exactly the attributes
    update_params reads (its __init__ needs the data set and MuJoCo).  The LambdaLR horizon is shortened (num_epoch_fix 0, num_epoch 4) so
    that two iterations see the decay.  The fixture holds the batch, the seeds of the weights, and -- per iteration -- the advantages / returns,
    every epoch's surrogate, value loss and step loss, the gradient norms clip_grad_norm_ reported (the generator-consumed clip: only the
    very first call sees parameters), the learning rates, and a handful of parameter tensors after each iteration."""
    import kin_poly.models.traj_ar_smpl_net as tn
    import kin_poly.models.policy_ar as par
    import kin_poly.core.agent_ar as aar
    import kin_poly.utils.torch_smpl_humanoid as tsh
    tsh.load_model_from_path = lambda f: fake_mj_model()
    policy_specs = dict(policy_v=1, log_std=-3.2, fix_std=True, gamma=0.95, tau=0.95, policy_lr=1e-5, value_lr=3e-4, clip_epsilon=0.2,
                        rl_update=True, init_update=False, step_update=True, full_update=False, num_step_update=20, num_optim_epoch=10)
    cfg = types.SimpleNamespace(model_specs=dict(model_v=1, rnn_hdim=1024, mlp_hsize=[1024, 512, 256], mlp_htype="relu", rnn_type="gru",
                                                 w_rp=50.0, w_rr=50.0, w_p=1.0, w_v=1.0, w_ee=10.0, w_op=1.0, w_or=10.0),
                                policy_specs=policy_specs, mujoco_model_file="unused.xml", use_of=False, use_head=True, use_action=True, use_vel=False,
                                use_context=False, add_noise=False, noise_std=0.01, has_z=True, data_dir=os.path.join(REF, "sample_data"),
                                lr=5e-4, num_epoch_fix=0, num_epoch=4, smooth=True, model_dir="unused", joint_controller=False)
    cfg.get = lambda k, dflt=None: getattr(cfg, k, dflt)
    rng = np.random.default_rng(306)
    Bd, Td = 2, 4
    data = dict(qpos=np.stack([[rand_qpos(rng, 0.15) for _ in range(Td)] for _ in range(Bd)]), qvel=rng.normal(size=(Bd, Td, 75)) * 0.1,
                target=rng.normal(size=(Bd, Td, 80)) * 0.1,
                head_pose=np.concatenate([rng.normal(size=(Bd, Td, 3)), np.stack([[rand_quat(rng) for _ in range(Td)] for _ in range(Bd)])], 2),
                head_vels=rng.normal(size=(Bd, Td, 6)) * 0.3, obj_head_relative_poses=rng.normal(size=(Bd, Td, 7)) * 0.3,
                obj_pose=np.tile(np.array([0.3, 0.2, 0.1, 1.0, 0, 0, 0]), (Bd, Td, 1)), action_one_hot=np.tile(np.array([0, 0, 1.0, 0]), (Bd, Td, 1)))
    data_t = {k: torch.tensor(v) for k, v in data.items()}
    dev, dt = torch.device("cpu"), torch.float64
    pol = par.PolicyAR(cfg, data_t, dev, dt, mode="train")
    sd = seeded_state_dict(pol.traj_ar_net, 29)
    for k in sd:
        if k.startswith(("action_fc", "context_fc")):
            sd[k] = sd[k] * 0.05
    pol.traj_ar_net.load_state_dict(sd)
    from uhc.khrylib.rl.core.critic import Value as RefValue
    val = RefValue(MLP(pol.state_dim, [512, 256], "relu"))
    val.load_state_dict(seeded_state_dict(val, 30))

    ag = aar.AgentAR.__new__(aar.AgentAR)
    ag.cfg, ag.dtype, ag.device, ag.policy_net, ag.value_net = cfg, dt, dev, pol, val
    ag.update_modules = [pol, val]
    ag.gamma, ag.tau, ag.clip_epsilon, ag.opt_num_epochs, ag.value_opt_niter = 0.95, 0.95, 0.2, 10, 1
    ag.optimizer_policy = torch.optim.Adam(pol.parameters(), lr=policy_specs["policy_lr"], weight_decay=0.0)            # setup_optimizer :184-199
    ag.optimizer_value = torch.optim.Adam(val.parameters(), lr=policy_specs["value_lr"], weight_decay=0.0)
    from kin_poly.utils.torch_ext import get_scheduler
    ag.scheduler_policy = get_scheduler(ag.optimizer_policy, policy="lambda", nepoch_fix=cfg.num_epoch_fix, nepoch=cfg.num_epoch)
    ag.scheduler_value = get_scheduler(ag.optimizer_value, policy="lambda", nepoch_fix=cfg.num_epoch_fix, nepoch=cfg.num_epoch)
    ag.policy_grad_clip = [(pol.parameters(), 40)]                                                                      # :92-93: a generator
    ag.epoch = 0

    # the recorded batch: 8 workers' rows back to back, 12 rows each, every worker's rows made of whole episodes (masks == 0 on their last rows)
    N, T = 8, 12
    ep_lens = [[12], [5, 7], [3, 4, 5], [12], [1, 11], [6, 6], [2, 2, 8], [9, 3]]
    B = N * T
    masks = np.ones(B)
    off = 0
    for lens in ep_lens:
        assert sum(lens) == T
        for L in lens:
            off += L
            masks[off - 1] = 0
    out = dict(seed_policy=29, seed_value=30, N=N, T=T, masks=masks, policy_lr=1e-5, value_lr=3e-4, sup_lr=5e-4, num_epoch_fix=0, num_epoch=4,
               keys=np.array(list(pol.traj_ar_net.state_dict().keys())),
               shapes=np.array([list(v.shape) + [0] * (2 - v.dim()) for v in pol.traj_ar_net.state_dict().values()]),
               value_keys=np.array(list(val.state_dict().keys())),
               value_shapes=np.array([list(v.shape) + [0] * (2 - v.dim()) for v in val.state_dict().values()]),
               state_dim=pol.state_dim, context_dim=pol.traj_ar_net.context_dim)
    watch_p = ("action_fc.bias", "action_mlp.affine_layers.2.bias", "action_rnn.rnn_f.bias_hh", "action_fc.weight")
    watch_v = ("value_head.weight", "net.affine_layers.1.bias")
    params_p, params_v = dict(pol.traj_ar_net.named_parameters()), dict(val.named_parameters())

    rec = {}
    orig_ppo, orig_uv, orig_lite, orig_clip, orig_est = aar.AgentAR.ppo_loss, aar.AgentAR.update_value, pol.traj_ar_net.compute_loss_lite, torch.nn.utils.clip_grad_norm_, aar.estimate_advantages

    def ppo_loss(self, log_probs, advantages, fixed_log_probs, ind):
        loss, ratio = orig_ppo(self, log_probs, advantages, fixed_log_probs, ind)
        rec["surr"].append(float(loss)); rec["ratio_mean"].append(float(ratio.mean()))
        return loss, ratio

    def update_value(self, states, returns):
        with torch.no_grad():
            rec["vloss"].append(float((self.value_net(states) - returns).pow(2).mean()))
        return orig_uv(self, states, returns)

    def lite(pred, gt, return_mean=True):
        loss, idv = orig_lite(pred, gt, return_mean)
        rec["step"].append(float(loss))
        return loss, idv

    def clip(params, max_norm, *a, **k):
        r = orig_clip(params, max_norm, *a, **k)
        rec["clip_norm"].append(float(r))
        return r

    def est(rewards, masks_, values, gamma, tau):
        adv, ret = orig_est(rewards, masks_, values, gamma, tau)
        rec["adv"], rec["ret"] = adv.numpy().copy(), ret.numpy().copy()
        return adv, ret

    aar.AgentAR.ppo_loss, aar.AgentAR.update_value, pol.traj_ar_net.compute_loss_lite, aar.estimate_advantages = ppo_loss, update_value, lite, est
    torch.nn.utils.clip_grad_norm_ = clip
    try:
        for it in range(2):
            states = rng.normal(size=(B, pol.state_dim)) * 0.5
            # actions: the behaviour policy's own samples at the parameters this iteration starts from (epoch 0's ratio is then 1)
            pol.set_mode("train")
            pol.initialize_rnn((torch.tensor(masks), torch.zeros((B, 3))))
            with torch.no_grad():
                _, mean0, _ = pol.forward(torch.tensor(states))
            actions = mean0.numpy() + np.exp(-3.2) * rng.normal(size=(B, 80))
            curr = np.stack([rand_qpos(rng, 0.2) for _ in range(B)])
            gt = curr.copy(); gt[:, :3] += rng.normal(size=(B, 3)) * 0.02; gt[:, 7:] += rng.normal(size=(B, 69)) * 0.05
            gt[:, 3:7] = np.stack([quaternion_multiply(q, quat_from_expmap(rng.normal(size=3) * 0.05)) for q in curr[:, 3:7]])
            batch = types.SimpleNamespace(states=states, actions=actions, rewards=rng.uniform(0.2, 0.9, size=B), masks=masks.copy(), exps=np.ones(B),
                                          v_metas=np.zeros((B, 3)), gt_target_qpos=gt, curr_qpos=curr, res_qpos=curr.copy(),
                                          cc_action=np.zeros((B, 75)), cc_state=np.zeros((B, 4)))
            rec.update(surr=[], ratio_mean=[], vloss=[], step=[], clip_norm=[])
            ag.epoch = it
            ag.per_epoch_update(it)                                      # optimize_policy :264-269: the schedulers step BEFORE the iteration's update
            lr_p, lr_v, lr_s = ag.optimizer_policy.param_groups[0]["lr"], ag.optimizer_value.param_groups[0]["lr"], pol.optimizer.param_groups[0]["lr"]
            ag.update_params(batch)
            tag = f"it{it}_"
            out.update({tag + "states": states, tag + "actions": actions, tag + "rewards": batch.rewards, tag + "curr_qpos": curr, tag + "gt_target_qpos": gt,
                        tag + "mean0": mean0.numpy(), tag + "adv": rec["adv"], tag + "ret": rec["ret"], tag + "surr": np.array(rec["surr"]),
                        tag + "ratio_mean": np.array(rec["ratio_mean"]), tag + "vloss": np.array(rec["vloss"]), tag + "step": np.array(rec["step"]),
                        tag + "clip_norm": np.array(rec["clip_norm"]), tag + "lr": np.array([lr_p, lr_v, lr_s])})
            for w in watch_p:
                out[tag + "p:" + w] = params_p[w].detach().numpy().copy()
            for w in watch_v:
                out[tag + "v:" + w] = params_v[w].detach().numpy().copy()
    finally:
        aar.AgentAR.ppo_loss, aar.AgentAR.update_value, pol.traj_ar_net.compute_loss_lite, aar.estimate_advantages = orig_ppo, orig_uv, orig_lite, orig_est
        torch.nn.utils.clip_grad_norm_ = orig_clip
    np.savez(os.path.join(OUT, "update_params.npz"), **out)
    print("update_params: clip norms it0", out["it0_clip_norm"][:3], "... it1", out["it1_clip_norm"][:3], "surr it0", out["it0_surr"], "lr", out["it0_lr"], out["it1_lr"])


def gen_loss_and_checkpoint(hum):
    """TrajARNet.step + compute_loss_lite (traj_ar_smpl_net.py:292-330, 459-497) on seeded poses, and a small pickle in
    the reference's checkpoint layout written with the reference's own ZFilter class (agent_ar.py:341-364)."""
    import pickle
    import kin_poly.models.traj_ar_smpl_net as tn
    import kin_poly.utils.torch_smpl_humanoid as tsh
    tsh.load_model_from_path = lambda f: fake_mj_model()
    cfg = types.SimpleNamespace(model_specs=dict(model_v=1, rnn_hdim=8, mlp_hsize=[8, 8], mlp_htype="relu", rnn_type="gru", w_rp=50.0, w_rr=50.0, w_p=1.0, w_ee=10.0),
                                mujoco_model_file="unused.xml", use_of=False, use_head=True, use_action=True, use_vel=False, use_context=False,
                                add_noise=False, noise_std=0.01, has_z=True, data_dir=os.path.join(REF, "sample_data"))
    rng = np.random.default_rng(107)
    B = 6
    cur = np.stack([rand_qpos(rng, 0.2) for _ in range(B)]); gt = np.stack([rand_qpos(rng, 0.2) for _ in range(B)])
    act = rng.normal(size=(B, 80)) * 0.3; act[:, 1:5] = np.stack([rand_quat(rng) for _ in range(B)])
    data_t = dict(qpos=torch.tensor(cur)[:, None], qvel=torch.zeros(B, 1, 75), target=torch.zeros(B, 1, 80), head_pose=torch.zeros(B, 1, 7),
                  head_vels=torch.zeros(B, 1, 6), obj_head_relative_poses=torch.zeros(B, 1, 7), obj_pose=torch.zeros(B, 1, 7), action_one_hot=torch.zeros(B, 1, 4))
    data_t["head_pose"][:, :, 3] = 1; data_t["obj_pose"][:, :, 3] = 1
    net = tn.TrajARNet(cfg, data_sample=data_t, device=torch.device("cpu"), dtype=torch.float64, mode="test", as_policy=True)
    with torch.no_grad():
        net.set_sim(torch.tensor(cur))
        nxt, nqvel = net.step(torch.tensor(act))
        loss, idv = net.compute_loss_lite(nxt, torch.tensor(gt))
    np.savez(os.path.join(OUT, "step_loss.npz"), cur=cur, gt=gt, act=act, next_qpos=nxt.numpy(), next_qvel=nqvel.numpy(), loss=loss.item(), loss_idv=np.array(idv))
    zf = ZFilter((5,), clip=5)
    for x in rng.normal(size=(20, 5)):
        zf(x)
    cp = {"policy_dict": {"traj_ar_net.action_fc.weight": torch.tensor(rng.normal(size=(2, 3))), "action_log_std": torch.ones(1, 2) * -3.2},
          "value_dict": {"value_head.bias": torch.zeros(1)}, "running_state": zf}
    with open(os.path.join(OUT, "ref_checkpoint_small.p"), "wb") as f:
        pickle.dump(cp, f)
    np.savez(os.path.join(OUT, "ref_checkpoint_small_expect.npz"), mean=zf.rs.mean, std=zf.rs.std, w=cp["policy_dict"]["traj_ar_net.action_fc.weight"].numpy())


def gen_ppo_loss():
    """AgentPPO.ppo_loss (uhc/khrylib/rl/agents/agent_ppo.py:58-65) with DiagGaussian log-probabilities
    (uhc/khrylib/rl/core/distributions.py:6-23) on seeded means / actions / advantages, exps mask included."""
    from uhc.khrylib.rl.agents.agent_ppo import AgentPPO
    from uhc.khrylib.rl.core.distributions import DiagGaussian
    rng = np.random.default_rng(31)
    B, A = 96, 80
    log_std = -3.2
    mean_old = rng.normal(size=(B, A)) * 0.3
    mean_new = mean_old + rng.normal(size=(B, A)) * 0.5 * np.exp(log_std)
    actions = mean_old + rng.normal(size=(B, A)) * np.exp(log_std)
    adv = rng.normal(size=(B, 1))
    exps = (rng.uniform(size=B) < 0.8).astype(np.float64)

    class Pol:                                   # policy_net.get_log_prob(x, a) = forward(x).log_prob(a)  (policy.py)
        def __init__(self, mean):
            self.mean = torch.tensor(mean)

        def get_log_prob(self, x, a):
            idx = x.long().view(-1)
            return DiagGaussian(self.mean[idx], torch.full((idx.numel(), A), float(np.exp(log_std)))).log_prob(a)

    states = torch.arange(B, dtype=torch.float64).view(B, 1)        # "states" index the stored means
    fixed = Pol(mean_old).get_log_prob(states, torch.tensor(actions))
    stub = types.SimpleNamespace(policy_net=Pol(mean_new), trans_policy=lambda x: x, clip_epsilon=0.2)
    ind = torch.tensor(exps).nonzero(as_tuple=False).squeeze(1)
    loss = AgentPPO.ppo_loss(stub, states, torch.tensor(actions), torch.tensor(adv), fixed, ind)
    np.savez(os.path.join(OUT, "ppo_loss.npz"), mean_old=mean_old, mean_new=mean_new, actions=actions, adv=adv, exps=exps, log_std=log_std,
             fixed_log_probs=fixed.numpy(), new_log_probs=Pol(mean_new).get_log_prob(states, torch.tensor(actions)).numpy(), loss=float(loss), clip_epsilon=0.2)


def gen_uhc_expert_reward():
    """UHC training env (uhc/envs/humanoid_im.py HumanoidEnv) pieces: get_expert (uhc/utils/tools.py:20-85) over a seeded clip,
    world_rfc_implicit_reward (uhc/core/reward_function.py:4-53), calc_body_diff (mean form, humanoid_im.py:719-726), the
    784-d observation against expert frame t + 1.  sim.forward() is played by this repo's oracle (MuJoCo-side inputs)."""
    from uhc.utils.tools import get_expert
    from uhc.core.reward_function import world_rfc_implicit_reward
    rng = np.random.default_rng(41)
    env = make_env(him.HumanoidEnv)
    env.frame_skip = 15                              # dt = model.opt.timestep * frame_skip (a property)
    env.cur_t, env.start_ind = 0, 0
    o = OracleSim()

    class Sim:
        def get_state(self): return None
        def set_state(self, st): pass
        def forward(self_inner):
            o.reset(env.data.qpos[:76].copy(), np.zeros(75))
            d = env.data
            d.body_xpos = np.vstack([np.zeros((1, 3)), o.get("xpos").reshape(NB, 3)])
            d.body_xquat = np.vstack([[[1, 0, 0, 0]], o.get("xquat").reshape(NB, 4)])
            d.xipos = np.vstack([np.zeros((1, 3)), o.get("xipos").reshape(NB, 3)])
            d.subtree_com = np.vstack([o.get("subtree_com")[None], np.zeros((24, 3))])
    env.sim = Sim()
    d = FakeData(); d.qpos = np.zeros(76); d.qvel = np.zeros(75)
    d.get_body_xipos = lambda name: d.xipos[(["world"] + NAMES).index(name)]
    env.data = d
    T = 12
    clip = np.stack([rand_qpos(rng, 0.15) for _ in range(T)])
    base = rand_qpos(rng, 0.3)
    for t in range(T):                               # a smooth-ish clip around one pose, with a joint crossing +-pi
        clip[t, :3] = base[:3] + 0.02 * t * np.array([1.0, 0.5, 0.1])
        clip[t, 7:] = base[7:] + 0.1 * np.sin(0.5 * t + np.arange(69))
    clip[:, 7 + 20] = np.linspace(3.0, 3.4, T)       # crosses pi: exercises the 2 pi unwrap of get_qvel_fd_new
    expert = get_expert(clip.copy(), {"cyclic": False, "seq_name": "synthetic"}, env)
    out = {"clip": clip}
    for k in ("qvel", "rlinv", "rlinv_local", "rangv", "rq_rmh", "com", "body_com", "head_pose", "ee_pos", "ee_wpos", "bquat", "bangvel", "wbpos", "wbquat"):
        out["e_" + k] = np.asarray(expert[k])
    out["e_height_lb"], out["e_head_height_lb"] = expert["height_lb"], expert["head_height_lb"]
    # reward / termination / observation at a simulated state a few frames into the clip
    env.expert = expert
    env.cfg = types.SimpleNamespace(reward_weights=dict(w_p=0.3, w_v=0.1, w_e=0.45, w_c=0.1, w_vf=0.05, k_p=2.0, k_v=0.005, k_e=5.0, k_c=100.0, k_vf=1.0),
                                    b_diffw=KPM["uhc_b_diffw"][1:].copy())    # uhc.yml body_params (toes / hands 0)
    rec = {k: [] for k in ("r_t", "r_qpos", "r_qvel", "r_prev_bquat", "r_action", "r_reward", "r_info", "r_body_diff", "r_obs", "r_xpos", "r_xquat", "r_xipos", "r_com")}
    for t in (1, 4, 9):
        q = clip[t] + np.concatenate([rng.normal(size=3) * 0.02, np.zeros(4), rng.normal(size=69) * 0.05]); q[3:7] /= np.linalg.norm(q[3:7])
        v = rng.normal(size=75) * 0.3
        env.data.qpos = q.copy(); env.data.qvel = v.copy()
        env.sim.forward()
        env.cur_t = t
        env.prev_bquat = out["e_bquat"][t - 1] + 0.0
        action = rng.normal(size=75) * 0.3
        r, info = world_rfc_implicit_reward(env, None, action, None)
        rec["r_t"].append(t); rec["r_qpos"].append(q); rec["r_qvel"].append(v); rec["r_prev_bquat"].append(env.prev_bquat.copy()); rec["r_action"].append(action)
        rec["r_reward"].append(r); rec["r_info"].append(info); rec["r_body_diff"].append(env.calc_body_diff())
        rec["r_obs"].append(env.get_full_obs_v1())
        rec["r_xpos"].append(env.data.body_xpos[1:25].copy()); rec["r_xquat"].append(env.data.body_xquat[1:25].copy()); rec["r_xipos"].append(env.data.xipos[1:25].copy())
        rec["r_com"].append(env.data.subtree_com[0].copy())
    out.update({k: np.stack(v) for k, v in rec.items()})
    np.savez(os.path.join(OUT, "uhc_expert_reward.npz"), **out)


def gen_dataset_features(hum):
    """Data formats either side of the path: per-take feature construction (kin_poly/data_process/process_smpl.py:30-135:
    get_head_vel, get_obj_relative_pose) and the dataset's derived trajectory (statear_smpl_dataset.py:153-214:
    get_traj_de_heading with has_z, get_root_vel), plus the adaptive take-sampling probabilities (:281-290, ewma)."""
    import kin_poly.data_process.process_smpl as ps
    import kin_poly.data_loaders.statear_smpl_dataset as dsm
    from kin_poly.utils.math_utils import ewma
    rng = np.random.default_rng(51)
    T = 14
    base = rand_qpos(rng, 0.2)
    clip = np.tile(base, (T, 1))
    for t in range(T):
        clip[t, :3] = base[:3] + 0.03 * t * np.array([1.0, -0.4, 0.05])
        ang = 0.25 * t
        clip[t, 3:7] = quaternion_multiply(np.array([np.cos(ang / 2), 0, 0, np.sin(ang / 2)]), base[3:7])      # turning about z: heading changes
        clip[t, 7:] = base[7:] + 0.1 * np.sin(0.4 * t + np.arange(69))
    head_pose = np.stack([np.concatenate([r["wbpos"].reshape(24, 3)[13], r["wbquat"].reshape(24, 4)[13]]) for r in (hum.qpos_fk(q.copy()) for q in clip)])
    obj_pose = np.tile(np.concatenate([[0.6, 0.2, 0.4], rand_quat(rng), [1.0, -0.3, 0.7], rand_quat(rng)]), (T, 1))
    obj_pose[:, 0] += 0.01 * np.arange(T)
    out = {"clip": clip, "head_pose": head_pose, "obj_pose": obj_pose,
           "head_vels": ps.get_head_vel(head_pose), "obj_head_relative_poses": ps.get_obj_relative_pose(obj_pose, head_pose, num_objs=2),
           "obj_root_relative_poses": ps.get_obj_relative_pose(obj_pose, clip[:, :7], num_objs=2)}
    ds = dsm.StateARDataset.__new__(dsm.StateARDataset)
    ds.cfg = types.SimpleNamespace(has_z=True); ds.dt = 1 / 30; ds.base_rot = [0.7071, 0.7071, 0.0, 0.0]
    out["traj_pos"] = ds.get_traj_de_heading(clip.copy()); out["traj_root_vel"] = ds.get_root_vel(clip.copy())
    # adaptive sampling probabilities over takes
    freq = {f"take{i}": [[int(rng.uniform() < 0.6), int(rng.integers(0, 50))] for _ in range(int(rng.integers(0, 12)))] for i in range(9)}
    temp = 0.5
    probs = np.exp(-np.array([ewma(np.array(freq[k])[:, 0] == 1) if len(freq[k]) > 0 else 0 for k in freq.keys()]) / temp)
    out["freq_success"] = np.array([np.array([r[0] for r in freq[k]] + [-1] * (12 - len(freq[k]))) for k in freq])   # -1 padded
    out["freq_probs"] = probs / probs.sum(); out["sampling_temp"] = temp
    np.savez(os.path.join(OUT, "dataset_features.npz"), **out)


if __name__ == "__main__" and os.environ.get("KP_GOLDEN_ONLY") == "data":
    gen_dataset_features(make_humanoid())
    print("dataset_features.npz ok")


if __name__ == "__main__" and os.environ.get("KP_GOLDEN_ONLY") == "uhc":
    gen_uhc_expert_reward()
    print("uhc_expert_reward.npz ok")


if __name__ == "__main__" and os.environ.get("KP_GOLDEN_ONLY") == "ppo":
    gen_ppo_loss()
    print("ppo_loss.npz ok")


if __name__ == "__main__" and os.environ.get("KP_GOLDEN_ONLY") == "loss":
    gen_loss_and_checkpoint(make_humanoid())
    print("step_loss.npz ok")


if __name__ == "__main__" and os.environ.get("KP_GOLDEN_ONLY") is None:
    np.savez(os.path.join(OUT, "standing_neutral.npz"), **{k: v for k, v in
             __import__("joblib").load(os.path.join(REF, "sample_data/standing_neutral.pkl")).items() if k in ("qpos", "qvel")})
    hum = make_humanoid()
    gen_quat_utils()
    gen_fk(hum)
    gen_env_fixtures(hum)
    gen_ar_obs_reward(hum)
    gen_gae_zfilter()
    gen_policies()
    gen_traj_ar_net(hum)
    gen_pretrain(hum)
    gen_update_params(hum)
    gen_metrics()
    gen_loss_and_checkpoint(hum)
    gen_ppo_loss()
    gen_uhc_expert_reward()
    gen_dataset_features(hum)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__" and os.environ.get("KP_GOLDEN_ONLY") == "metrics":
    gen_metrics()
    print("metrics.npz", os.path.getsize(os.path.join(OUT, "metrics.npz")))


if __name__ == "__main__" and os.environ.get("KP_GOLDEN_ONLY") == "pretrain":
    gen_pretrain(make_humanoid())
    print("pretrain.npz", os.path.getsize(os.path.join(OUT, "pretrain.npz")))


if __name__ == "__main__" and os.environ.get("KP_GOLDEN_ONLY") == "traj":
    gen_traj_ar_net(make_humanoid())
    print("traj_ar_net.npz", os.path.getsize(os.path.join(OUT, "traj_ar_net.npz")))


if __name__ == "__main__" and os.environ.get("KP_GOLDEN_ONLY") == "update":
    gen_update_params(make_humanoid())
    print("update_params.npz", os.path.getsize(os.path.join(OUT, "update_params.npz")))
