"""Where does an outlier of tools/obj_fuzz.py come from?  For one scene of that generator, walk the fp64 oracle's trajectory substep by substep and, at
every substep, start BOTH sides from the same fp32-rounded state and advance them one substep: the one-substep error isolates a real disagreement
of the arithmetic from the growth of an earlier one, and the contact sets of the two sides are compared at the same state.

    python tools/obj_fuzz_trace.py <seed> <scene> [n_scenes=64] [substeps=45]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _scenes  # noqa: E402
from kinpoly_amd.model_compiler import read_kpm  # noqa: E402
from kinpoly_amd.sim import STEP_KPM, KpModel, KpSim  # noqa: E402
from oracle.kpo import OracleSim  # noqa: E402

seed = int(sys.argv[1]); scene = int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
nsub = int(sys.argv[4]) if len(sys.argv) > 4 else 45
kpm = read_kpm(STEP_KPM)
S = _scenes.object_scenes(n, seed)                  # the generator of obj_fuzz.py, draw for draw
blk, qpos, qvel, action, scenes = S["blk"], S["qpos"], S["qvel"], S["action"], S["objects"]
r32 = _scenes.r32
dev = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")  # noqa: E731
e = scene
ois = sorted(scenes[e])
print(f"seed {seed} scene {e}: objects {ois}", flush=True)


def oracle_at(q, v, oq, ov):
    o = OracleSim(kpm=STEP_KPM)
    for slot, oi in enumerate(ois):
        o.set_object(slot, kpm, oi, oq[slot], ov[slot])
    o.reset(q, v)
    return o


sim = KpSim(KpModel(STEP_KPM), 1)
sim.record_contacts()
sim.set_objects(dev(blk[e:e + 1])); sim.set_state(dev(qpos[e:e + 1]), dev(qvel[e:e + 1])); sim.set_target(dev(qpos[e:e + 1]))
free = oracle_at(qpos[e], qvel[e], [blk[e, 7 * oi: 7 * oi + 7] for oi in ois], [np.zeros(6)] * len(ois))
a1 = action[e:e + 1]
worst = 0.0
for k in range(nsub):
    q, v = r32(free.get("qpos")), r32(free.get("qvel"))
    ob = [free.get_object(s) for s in range(len(ois))]
    oq, ov = [r32(x[0]) for x in ob], [r32(x[1]) for x in ob]
    o = oracle_at(q, v, oq, ov)
    b = blk[e:e + 1].copy(); bv = np.zeros((1, 30))
    for slot, oi in enumerate(ois):
        b[0, 7 * oi: 7 * oi + 7] = oq[slot]; bv[0, 6 * oi: 6 * oi + 6] = ov[slot]
    sim.set_obj_state(dev(b), dev(bv)); sim.set_state(dev(q[None]), dev(v[None]))
    sim.step_ctrl(dev(a1), 1)
    o.do_simulation(action[e], qpos[e], 1)
    hq = sim.get("qpos").double().cpu().numpy()[0]; hv = sim.get("qvel").double().cpu().numpy()[0]
    hob = sim.get("obj_qpos").double().cpu().numpy()[0]
    eq, ev = np.abs(o.get("qpos") - hq).max(), np.abs(o.get("qvel") - hv).max()
    eo = max(np.abs(o.get_object(s)[0] - hob[7 * oi: 7 * oi + 7]).max() for s, oi in enumerate(ois))
    c = o.contacts_full(); h = sim.contacts()[0]
    same = len(c["body"]) == len(h["body"]) and np.array_equal(c["body"], h["body"]) and np.array_equal(c["b2"], h["b2"])
    dg = sim.diag()[0]
    flag = "" if same else "   CONTACT SETS DIFFER"
    worst = max(worst, eq, eo)
    if not same or eq > 2e-6 or eo > 2e-6 or ev > 1e-3:
        print(f"substep {k:3d} (control step {k // 15}): one-substep |dqpos| {eq:.1e} |dqvel| {ev:.1e} object {eo:.1e}; contacts oracle {len(c['body'])} hip {len(h['body'])}; "
              f"newton it oracle {o.niter} hip {dg[1]}{flag}", flush=True)
        if same and len(c["body"]):
            # inside one (entity, entity) pair the order of the contacts carries no meaning: match every oracle contact with the nearest hip contact of its pair
            order = []
            used = set()
            for i in range(len(c["body"])):
                cand = [j for j in range(len(h["body"])) if j not in used and h["body"][j] == c["body"][i] and h["b2"][j] == c["b2"][i]]
                j = min(cand, key=lambda jj: np.abs(h["pos"][jj] - c["pos"][i]).max())
                used.add(j); order.append(j)
            cc = c; hh = {kk: vv[order] for kk, vv in h.items()}
            dd, dp, dn = np.abs(cc["dist"] - hh["dist"]), np.abs(cc["pos"] - hh["pos"]).max(1), np.abs(cc["normal"] - hh["normal"]).max(1)
            i = int(np.argmax(dd + dp + dn))
            print(f"      contact geometry: max |ddist| {dd.max():.1e} |dpos| {dp.max():.1e} |dnormal| {dn.max():.1e}; worst contact entities ({cc['body'][i]}, {cc['b2'][i]}) "
                  f"dist oracle {cc['dist'][i]:.6f} hip {hh['dist'][i]:.6f} normal oracle {np.round(cc['normal'][i], 5)} hip {np.round(hh['normal'][i], 5)}", flush=True)
            for tag, x in (("oracle", cc), ("hip   ", hh)):
                sel = [i2 for i2 in range(len(x["body"])) if x["body"][i2] == cc["body"][i] and x["b2"][i2] == cc["b2"][i]]
                print(f"      {tag} contacts of that pair: " + "; ".join(f"dist {x['dist'][i2]:.7f} pos {np.round(x['pos'][i2], 6)}" for i2 in sel), flush=True)
            print(f"      object state set: q {oq[0]} v {ov[0]}", flush=True)
            print(f"      pairs {sorted(set(zip(cc['body'].tolist(), cc['b2'].tolist())))}; min dist {cc['dist'].min():.5f}", flush=True)
        if not same:
            so, sh = set(zip(c["body"].tolist(), c["b2"].tolist())), set(zip(h["body"].tolist(), h["b2"].tolist()))
            print(f"      pairs only in oracle {sorted(so - sh)} only in hip {sorted(sh - so)}; oracle {list(zip(c['body'], c['b2'], np.round(c['dist'], 6)))}\n"
                  f"      hip {list(zip(h['body'], h['b2'], np.round(h['dist'], 6)))}", flush=True)
    free.do_simulation(action[e], qpos[e], 1)
print(f"worst one-substep error from a common state over {nsub} substeps: {worst:.2e}")

# ---- the same scene free-running on both sides (what obj_fuzz.py measures): where does the distance between the two trajectories start to grow?
sim.set_objects(dev(blk[e:e + 1])); sim.set_state(dev(qpos[e:e + 1]), dev(qvel[e:e + 1])); sim.set_target(dev(qpos[e:e + 1]))
free = oracle_at(qpos[e], qvel[e], [blk[e, 7 * oi: 7 * oi + 7] for oi in ois], [np.zeros(6)] * len(ois))
prev = 0.0
print("free-running trajectories, substep: |dqpos| (contacts oracle / hip)")
for k in range(nsub):
    sim.step_ctrl(dev(a1), 1); free.do_simulation(action[e], qpos[e], 1)
    err = np.abs(free.get("qpos") - sim.get("qpos").double().cpu().numpy()[0]).max()
    c = free.contacts_full(); h = sim.contacts()[0]
    so, sh = sorted(zip(c["body"].tolist(), c["b2"].tolist())), sorted(zip(h["body"].tolist(), h["b2"].tolist()))
    note = ""
    if so != sh:
        only_o = [p for p in set(so) if so.count(p) > sh.count(p)]; only_h = [p for p in set(sh) if sh.count(p) > so.count(p)]
        note = f"   contact sets differ: more in oracle {sorted(only_o)} more in hip {sorted(only_h)}"
        for p in sorted(set(only_o + only_h)):
            do = c["dist"][(c["body"] == p[0]) & (c["b2"] == p[1])]; dh = h["dist"][(h["body"] == p[0]) & (h["b2"] == p[1])]
            note += f"; pair {p} dist oracle {np.round(do, 6).tolist()} hip {np.round(dh, 6).tolist()}"
    if err > 10 * prev and so == sh and len(so):
        # the same pairs on both sides but the states were 1e-6 apart a substep ago: which contact moved?
        used, worst_c = set(), (0.0, None)
        for i in range(len(c["body"])):
            cand = [j for j in range(len(h["body"])) if j not in used and h["body"][j] == c["body"][i] and h["b2"][j] == c["b2"][i]]
            j = min(cand, key=lambda jj: np.abs(h["pos"][jj] - c["pos"][i]).max()); used.add(j)
            d = max(abs(c["dist"][i] - h["dist"][j]), np.abs(c["normal"][i] - h["normal"][j]).max())
            if d > worst_c[0]:
                worst_c = (d, f"pair ({c['body'][i]}, {c['b2'][i]}): dist oracle {c['dist'][i]:.6f} hip {h['dist'][j]:.6f}, normal oracle {np.round(c['normal'][i], 4).tolist()} hip {np.round(h['normal'][j], 4).tolist()}")
        note += f"   contact geometry of the colliding pass that produced this state differs most at {worst_c[1]}"
    if err > 1.5 * prev or note or k % 15 == 14:
        print(f"  {k:3d}: {err:.1e} ({len(so)} / {len(sh)}){note}", flush=True)
    prev = max(err, 1e-7)
