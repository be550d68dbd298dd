"""where does a top-up of the episode pool (VectorSampler._top_up: 16384 clips at fail rate 1 with pool_depth 4) spend its time?"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd import dataset as D
from kinpoly_amd import sim as kpsim
from kinpoly_amd.agent import AgentAR
from kinpoly_amd.model_compiler import read_kpm

n, m = 4096, 16384
std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), n, 0)
takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=8, T_range=(110, 160), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=4, with_objects=len(sys.argv) > 1)
ds = D.StateARDataset(takes, fr_num=100, seed=4, device=fk_sim.device)
agent = AgentAR(n, dataset=ds, device=0, horizon=24, pool_depth=4)
env, src = agent.env, agent.source


def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps, r


dt, data = t(lambda: ds.sample_batch(m, freq_dict=src.freq_dict, sampling_temp=0.3, sampling_freq=0.5)); print("dataset.sample_batch(%d): %.1f ms" % (m, dt * 1e3))
data = {k: (v.to(env.device) if torch.is_tensor(v) else v) for k, v in data.items()}
dt, out = t(lambda: agent.ctx_builder.init_context(data)); print("init_context (context GRU over 100 frames, no roll-out): %.1f ms" % (dt * 1e3))
rows = torch.arange(m, device=env.device) + n
dt, _ = t(lambda: env.write_context_rows(rows, out)); print("env.write_context_rows: %.1f ms" % (dt * 1e3))
q = out["qpos"].reshape(-1, 76).contiguous()
dt, _ = t(lambda: env.sim.fk(q)); print("   of which FK of the GT clip (%d rows): %.1f ms" % (q.shape[0], dt * 1e3))
dt, _ = t(lambda: [env.ctx[k].index_copy_(0, rows, out[k]) for k in ("qpos", "head_pose", "head_vels", "obj_head_relative_poses")]); print("   of which the four [m, T, .] row copies: %.1f ms" % (dt * 1e3))


# the same inside the sampler: host time of the pieces of _top_up over one sample(24) at fail rate 1 (with / without the init_context memo)
for memo in (False, True):
    ds2 = D.StateARDataset(takes, fr_num=100, seed=4, device=fk_sim.device)
    ag = AgentAR(n, dataset=ds2, device=0, horizon=24, pool_depth=4, cache_init_context=memo)
    acc = {"draw": 0.0, "write": 0.0, "plan": 0.0, "n": 0}
    sm, srcm, envm = ag.sampler, ag.source, ag.env
    d0, w0 = srcm.draw, envm.write_context_rows

    def draw(k, dev, d0=d0, acc=acc):
        torch.cuda.synchronize(); t0 = time.time(); r = d0(k, dev); torch.cuda.synchronize(); acc["draw"] += time.time() - t0; acc["n"] += 1; return r

    def write(rows_, data_, w0=w0, acc=acc):
        torch.cuda.synchronize(); t0 = time.time(); w0(rows_, data_); torch.cuda.synchronize(); acc["write"] += time.time() - t0
    srcm.draw, envm.write_context_rows = draw, write
    sm.sample(24); torch.cuda.synchronize()
    for k in acc: acc[k] = 0 if k == "n" else 0.0
    t0 = time.time(); sm.sample(24); torch.cuda.synchronize(); tot = time.time() - t0
    print(f"memo={memo}: sample(24) {tot * 1e3:.0f} ms; {acc['n']} draws: draw {acc['draw'] * 1e3:.0f} ms, write rows {acc['write'] * 1e3:.0f} ms", flush=True)
    r0 = srcm.record
    tr = [0.0]

    def rec(*a, r0=r0, tr=tr, **k):
        t0 = time.time(); r0(*a, **k); tr[0] += time.time() - t0
    srcm.record = rec
    for k in acc: acc[k] = 0 if k == "n" else 0.0
    t0 = time.time(); b = sm.sample(24); torch.cuda.synchronize(); tot = time.time() - t0
    t0 = time.time(); srcm.draw(1, envm.device); tp = time.time() - t0      # the first draw after a record() evaluates take_probs on the new freq_dict
    print(f"   again: sample(24) {tot * 1e3:.0f} ms; draw {acc['draw'] * 1e3:.0f}, write {acc['write'] * 1e3:.0f}, record {tr[0] * 1e3:.0f} ms; first draw after it (take_probs) {tp * 1e3:.0f} ms; episodes {len(b.episodes['percent'])}", flush=True)

# the loop alone (no episode source: a finished env restarts on its own clip), with and without the recorded qpos rows
from kinpoly_amd.rollout import VectorSampler
for rq in (False, True):
    vs = VectorSampler(ag.env, ag.policy_net, record_qpos=rq)
    vs.obs, vs.hx, vs.fresh = sm.obs, sm.hx, sm.fresh
    vs.sample(24); torch.cuda.synchronize()
    t0 = time.time(); vs.sample(24); torch.cuda.synchronize()
    print(f"loop only, record_qpos={rq}: sample(24) {(time.time() - t0) * 1e3:.0f} ms", flush=True)
