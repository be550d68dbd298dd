"""Does the first substep see the object-floor contacts of a table hovering `gap` above the floor (gap < margin = 1 mm)?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.model_compiler import read_kpm
from kinpoly_amd.sim import STEP_KPM, KpModel, KpSim
from oracle.kpo import OracleSim
kpm = read_kpm(STEP_KPM)
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
x0, y0 = std["qpos"][0], std["qpos"][1]
dev = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=torch.float32, device="cuda")
for objs in ({2: [x0 + 2.0, y0, 0.7905]}, {1: [x0 + 2.0, y0, 0.921], 2: [x0 + 2.0, y0, 0.7905]}, {4: [x0 + 2.0, y0, 0.3705]}):
    for gap in (0.0, 0.0002, 0.0005, 0.0009):
        blk = np.zeros((1, 35))
        for i in range(5):
            blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
        sc = {}
        for oi, p in objs.items():
            sc[oi] = [p[0], p[1], p[2] + gap, 1, 0, 0, 0]; blk[0, 7 * oi: 7 * oi + 7] = sc[oi]
        q = std["qpos"][None].copy(); v = np.zeros((1, 75))
        sim = KpSim(KpModel(STEP_KPM), 1)
        sim.set_objects(dev(blk)); sim.set_state(dev(q), dev(v)); sim.set_target(dev(q))
        a = dev(np.zeros((1, 75)))
        out = []
        for k in range(3):
            sim.step_ctrl(a, 1)
            out.append((int(sim.diag()[0, 0]), [round(float(sim.get("obj_qvel")[0, 6 * oi + 2]), 5) for oi in sorted(objs)]))
        o = OracleSim(kpm=STEP_KPM)
        for slot, oi in enumerate(sorted(sc)):
            o.set_object(slot, kpm, oi, sc[oi])
        o.reset(q[0], v[0]); oo = []
        for k in range(3):
            o.do_simulation(np.zeros(75), q[0], 1)
            oo.append((len(o.contact_pairs()[0]), [round(float(o.get_object(slot)[1][2]), 5) for slot in range(len(sc))]))
        print(f"objects {sorted(objs)} gap {gap * 1e3:.1f} mm: HIP (ncon, vz) {out}   oracle {oo}")
