"""A closer look at one state dumped by tools/substep_parity.py (KP_DUMP=substep,scene): contact geometry and the one-substep result on both sides.
    python tools/micro/substep_state.py tools/micro/states/<file>.npz"""
import os
import sys

import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.model_compiler import read_kpm
from kinpoly_amd.sim import STEP_KPM, KpModel, KpSim
from oracle.kpo import OracleSim
d = np.load(sys.argv[1])
kpm = read_kpm(STEP_KPM)
objs = [int(x) for x in d["objects"]]
o = OracleSim(kpm=STEP_KPM)
for slot, oi in enumerate(objs):
    o.set_object(slot, kpm, oi, d["blk"][7 * oi: 7 * oi + 7], d["bv"][6 * oi: 6 * oi + 6])
o.reset(d["qpos"], d["qvel"])
dev = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=torch.float32, device="cuda")  # noqa: E731
sim = KpSim(KpModel(STEP_KPM), 1)
sim.record_contacts()
sim.set_objects(dev(d["blk"][None])); sim.set_obj_state(dev(d["blk"][None]), dev(d["bv"][None])); sim.set_state(dev(d["qpos"][None]), dev(d["qvel"][None])); sim.set_target(dev(d["target"][None]))
sim.step_ctrl(dev(d["action"][None]), 1)
o.do_simulation(d["action"], d["target"], 1)
c = o.contacts_full(); h = sim.contacts()[0]
np.set_printoptions(precision=7, suppress=True, linewidth=200)
for tag, x in (("oracle", c), ("hip", h)):
    for i in range(len(x["body"])):
        print(f"{tag:6s} contact ({x['body'][i]}, {x['b2'][i]}): dist {x['dist'][i]:.8f} pos {x['pos'][i]} normal {x['normal'][i]}")
hv = sim.get("qvel").double().cpu().numpy()[0]; wv = o.get("qvel")
dv = hv - wv
print("efc force (oracle)", o.efc()[0])
print(f"|dqvel| max {np.abs(dv).max():.2e} at dof {int(np.abs(dv).argmax())}; qacc of the constraint (oracle) there {((o.qacc_full() - o.qacc_smooth_full())[int(np.abs(dv).argmax())]):.3f}; Newton iterations oracle {o.niter} hip {sim.diag()[0, 1]}")
if len(c["body"]) == len(h["body"]) and len(c["body"]):
    ang = [float(np.arctan2(np.linalg.norm(np.cross(c["normal"][i], h["normal"][i])), np.dot(c["normal"][i], h["normal"][i]))) for i in range(len(c["body"]))]
    print("angle between the normals (rad):", np.round(ang, 6), "  |ddist|:", np.abs(c["dist"] - h["dist"]))
