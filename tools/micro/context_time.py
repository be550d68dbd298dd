"""where does VectorSampler._refill's time go (dataset batch, context GRU, kinematic roll-out, FK)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from kinpoly_amd import sim as kpsim, dataset as D
from kinpoly_amd.agent import AgentAR
from kinpoly_amd.model_compiler import read_kpm
n = 4096
std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), n, 0)
takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=4, T_range=(110, 160), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=4)
ds = D.StateARDataset(takes, fr_num=100, seed=4, device=fk_sim.device)
agent = AgentAR(n, dataset=ds, device=0, horizon=24, sampling_temp=0.3, sampling_freq=0.5)
src, cb = agent.source, agent.ctx_builder
def t(f, reps=3):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.time() - t0) / reps, r
dt, d = t(lambda: src.draw(n, agent.device)); print("source.draw (dataset batch + init_context): %.3f s" % dt)
dt, b = t(lambda: ds.sample_batch(n) if hasattr(ds, "sample_batch") else None); print("dataset.sample_batch: %.3f s" % dt)
data = {k: v for k, v in d.items()}
net = agent.policy_net
with torch.no_grad():
    dt, _ = t(lambda: net.get_context_feat(data)); print("context GRU over the clip: %.3f s" % dt)
    dt, st = t(lambda: net.init_states(data)); print("init_states (incl. context GRU): %.3f s" % dt)
    dt, _ = t(lambda: net.rollout(data, agent.kin_sim, st[0], st[1])); print("kinematic roll-out of the clip: %.3f s" % dt)
    dt, _ = t(lambda: cb.init_context(data)); print("init_context total: %.3f s" % dt)
dt, _ = t(lambda: agent.sampler._refill(), 2); print("_refill (2 levels): %.3f s" % dt)
