"""Per-substep HIP-vs-oracle trace of one scene of tools/obj_fuzz.py:  python tools/micro/obj_scene_trace.py <n> <seed> <scene>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
n, seed, scene = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nsub_total = int(sys.argv[4]) if len(sys.argv) > 4 else 45
sys.argv = ["obj_fuzz.py", str(n), "0", str(seed)]            # nstep = 0: build the scenes only
src = open(os.path.join(ROOT, "tools", "obj_fuzz.py")).read()
src = src[:src.index("sim = KpSim(KpModel(STEP_KPM), n)")].replace("ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", "")
exec(src)
e = scene
print("scene", e, "objects", scenes[e])
sim = KpSim(KpModel(STEP_KPM), 1)
sim.set_objects(dev(blk[e:e + 1])); sim.set_state(dev(qpos[e:e + 1]), dev(qvel[e:e + 1])); sim.set_target(dev(qpos[e:e + 1]))
a_t = dev(action[e:e + 1])
o = OracleSim(kpm=STEP_KPM)
for slot, oi in enumerate(sorted(scenes[e])):
    o.set_object(slot, kpm, oi, scenes[e][oi])
o.reset(qpos[e], qvel[e])
for k in range(nsub_total):
    sim.step_ctrl(a_t, 1)
    o.do_simulation(action[e], qpos[e], 1)
    got = sim.get("qpos").double().cpu().numpy()[0]; gv = sim.get("qvel").double().cpu().numpy()[0]
    gobj = sim.get("obj_qpos").double().cpu().numpy()[0]
    dg = sim.diag()[0]
    eo = max(np.abs(o.get_object(slot)[0] - gobj[7 * oi: 7 * oi + 7]).max() for slot, oi in enumerate(sorted(scenes[e])))
    dq = np.abs(o.get("qpos") - got); dv = np.abs(o.get("qvel") - gv)
    print(f"substep {k:2d}: |dqpos| {dq.max():.2e} (dof {int(dq.argmax())}) |dqvel| {dv.max():.2e} (dof {int(dv.argmax())}) obj {eo:.2e}  HIP ncon {dg[0]} it {dg[1]}  oracle ncon {len(o.contact_pairs()[0])}")

# ---- first-substep detail: object velocities after one substep, HIP vs oracle, from a fresh start
sim2 = KpSim(KpModel(STEP_KPM), 1)
sim2.set_objects(dev(blk[e:e + 1])); sim2.set_state(dev(qpos[e:e + 1]), dev(qvel[e:e + 1])); sim2.set_target(dev(qpos[e:e + 1]))
sim2.step_ctrl(a_t, 1)
o2 = OracleSim(kpm=STEP_KPM)
for slot, oi in enumerate(sorted(scenes[e])):
    o2.set_object(slot, kpm, oi, scenes[e][oi])
o2.reset(qpos[e], qvel[e]); o2.do_simulation(action[e], qpos[e], 1)
gv = sim2.get("obj_qvel").double().cpu().numpy()[0]
b1, b2 = o2.contact_pairs()
print("oracle contact pairs (entity a, entity b):", sorted(set(zip(b1.tolist(), b2.tolist()))), "counts", {k: int(((b1 == k[0]) & (b2 == k[1])).sum()) for k in set(zip(b1.tolist(), b2.tolist()))})
for slot, oi in enumerate(sorted(scenes[e])):
    print(f"object {oi} (slot {slot}) qvel after 1 substep: HIP {np.round(gv[6 * oi: 6 * oi + 6], 5)}  oracle {np.round(o2.get_object(slot)[1], 5)}")
