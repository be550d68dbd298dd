"""One-substep parity from common states.  Trajectory sweeps (floor_fuzz.py, obj_fuzz.py) measure the distance between two free-running trajectories,
which mixes what the arithmetic disagrees on with how fast the scene amplifies it.  Here every scene follows the fp64 oracle's trajectory and, at EVERY
substep, both sides start from the same fp32-rounded state (humanoid and objects, positions and velocities) and advance ONE substep: the error is the
disagreement of one substep's arithmetic, and the contact sets are compared at the same state.

    python tools/substep_parity.py floor|objects|bench:tracked|bench:random_init|bench:objects [n_scenes=64] [seed] [substeps=45]"""
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

LS_EXACT = bool(int(os.environ.get("KP_ORACLE_LS_EXACT", "0")))     # 1: the oracle's line search returns the exact minimiser (as the kernel's does) instead of MuJoCo's PrimalSearch

DUMP = tuple(int(x) for x in os.environ["KP_DUMP"].split(",")) if "KP_DUMP" in os.environ else None

MARGIN = 0.001        # geom margin of the compiled models (the blob's `opt`)
_K = read_kpm(STEP_KPM)
RBOUND, PM_TOL = _K["mesh_rbound"], float(_K["planemesh"][1])


def bench_states(workload, n, seed):
    """States of bench.py's own workloads ('tracked', 'random_init', 'objects'): the engine of that workload is built and stepped as the bench does
    (warm-up + 25 env-steps), then n envs' states after the last step, with that step's UHC action and target, become the scenes."""
    import bench
    rec, env, policy, sampler, std = bench.run_workload(workload, 0, 4 if seed is None else seed, 64, 24, 5)
    keep = {}
    orig = env.step

    def recording_step(*a, **k):
        out = orig(*a, **k); keep["info"] = out[3]; return out
    env.step = recording_step
    a_track = bench.tracking_action(env) if workload == "tracked" else None
    if workload == "tracked":
        sampler.start(); bench.stagger_episodes(env, sampler, 4, False)
        bench.rollout_steps(sampler, 10, a_track, False, False)
    else:
        bench.rollout_steps(sampler, 1, None, False, workload == "objects")
    S = states_from_engine(env, keep["info"]["cc_action"], n, workload == "objects")
    del env, sampler, policy
    torch.cuda.empty_cache()
    return S


def states_from_engine(env, cc_action, n, objects=False):
    """n envs' states of a live BatchedHumanoidAREnv (evenly spaced over its envs) after its last step, with that step's UHC action and target, as scenes
    of run(): what bench.py hands over for its `parity_live` block (the driver's own run, the same engine it just timed)."""
    sim = env.sim
    pick = np.linspace(0, env.n - 1, n).astype(int)
    g = lambda k: sim.get(k).double().cpu().numpy()[pick]  # noqa: E731
    qpos, qvel, target = g("qpos"), g("qvel"), g("target_qpos")
    action = cc_action.double().cpu().numpy()[pick]
    workload = "objects" if objects else "floor"
    blk = np.zeros((n, 35)); bv = np.zeros((n, 30))
    for i in range(5):
        blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
    objects = [{} for _ in range(n)]
    if workload == "objects":
        blk, bv = g("obj_qpos"), g("obj_qvel")
        for e in range(n):
            for oi in range(5):
                if abs(blk[e, 7 * oi]) < 50 and abs(blk[e, 7 * oi + 1]) < 50:            # not parked (convert_obj_qpos parks the inactive ones 100+ m away)
                    objects[e][oi] = blk[e, 7 * oi: 7 * oi + 7].copy()
            objects[e] = dict(list(objects[e].items())[:2])
    return dict(qpos=qpos, qvel=qvel, action=action, target=target, kind=np.zeros(n, int), objects=objects, blk=blk, bv=bv)


def run(mode="floor", n=64, seed=None, nsub=45, states=None, follow_flips=True):
    """-> dict(eq, ev, eo [nsub, n], differ [nsub, n] bool, ncon, scenes, flips).  states: scenes handed in (states_from_engine) instead of built here.
    follow_flips: where the two sides' contact sets differ at a common state (a knife edge of MuJoCo's contact rules), BOTH sides continue free-running
    from their own post-substep states for the rest of a control step (14 more substeps, same action and target), and `flips` records how far apart
    they are at its end: (substep, scene, one-substep |dqpos|, |dqpos| after the control step) -- is a flip damped or amplified?"""
    seed = (2024 if mode == "floor" else 0) if seed is None else seed
    obj = mode != "floor"
    if states is not None:
        S = states
        n = len(S["qpos"])
    elif mode.startswith("bench:"):
        S = bench_states(mode.split(":")[1], n, seed)
    else:
        S = _scenes.object_scenes(n, seed) if obj else _scenes.floor_scenes(n, seed)
    kpm = read_kpm(STEP_KPM) if obj else None
    r32 = _scenes.r32
    dev = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=torch.float32, device="cuda")  # noqa: E731
    sim = KpSim(KpModel(STEP_KPM) if obj else KpModel(), n)
    sim.record_contacts()
    if obj:
        sim.set_objects(dev(S["blk"]))
        if "bv" in S:
            sim.set_obj_state(dev(S["blk"]), dev(S["bv"]))
    sim.set_state(dev(S["qpos"]), dev(S["qvel"])); sim.set_target(dev(S["target"]))
    a_t = dev(S["action"])
    oracles = []
    for e in range(n):
        o = OracleSim(kpm=STEP_KPM, ls_exact=LS_EXACT) if obj else OracleSim(ls_exact=LS_EXACT)
        for slot, oi in enumerate(sorted(S["objects"][e])):
            o.set_object(slot, kpm, oi, S["objects"][e][oi], S["bv"][e, 6 * oi: 6 * oi + 6] if "bv" in S else None)
        o.reset(S["qpos"][e], S["qvel"][e])
        oracles.append(o)
    eq = np.zeros((nsub, n)); ev = np.zeros((nsub, n)); eo = np.zeros((nsub, n)); differ = np.zeros((nsub, n), bool); ncon = np.zeros((nsub, n), int)
    nit_o = np.zeros((nsub, n), int); nit_h = np.zeros((nsub, n), int); vertex = np.zeros((nsub, n), bool); klass = np.full((nsub, n), "", dtype=object)
    nang = np.zeros((nsub, n)); ddist = np.zeros((nsub, n))
    flips, side, o2 = [], None, None
    for k in range(nsub):
        q = np.stack([r32(o.get("qpos")) for o in oracles]); v = np.stack([r32(o.get("qvel")) for o in oracles])
        if obj:
            b = S["blk"].copy(); bv = np.zeros((n, 30))
            for e, o in enumerate(oracles):
                for slot, oi in enumerate(sorted(S["objects"][e])):
                    oq, ov = o.get_object(slot)
                    b[e, 7 * oi: 7 * oi + 7] = r32(oq); bv[e, 6 * oi: 6 * oi + 6] = r32(ov)
                    o.set_object(slot, kpm, oi, b[e, 7 * oi: 7 * oi + 7], bv[e, 6 * oi: 6 * oi + 6])
            sim.set_obj_state(dev(b), dev(bv))
        sim.set_state(dev(q), dev(v))
        if DUMP and DUMP[0] == k:               # KP_DUMP=substep,scene: the common state of that substep, for a closer look (tools/micro/substep_state.py)
            e = DUMP[1]
            np.savez(os.path.join(ROOT, "gpurun_out", f"substep_state_{mode}_{seed}_{k}_{e}.npz"), qpos=q[e], qvel=v[e], action=S["action"][e], target=S["target"][e],
                     blk=b[e] if obj else np.zeros(35), bv=bv[e] if obj else np.zeros(30), objects=np.asarray(sorted(S["objects"][e]), int))
        sim.step_ctrl(a_t, 1)
        hq = sim.get("qpos").double().cpu().numpy(); hv = sim.get("qvel").double().cpu().numpy()
        hob = sim.get("obj_qpos").double().cpu().numpy() if obj else None
        hc = sim.contacts()
        dg = sim.diag()
        assert dg[:, 2].max() == 0, "non-finite state"
        nit_h[k] = dg[:, 1]
        for e, o in enumerate(oracles):
            o.reset(q[e], v[e])
            o.do_simulation(S["action"][e], S["target"][e], 1)
            wv = o.get("qvel"); nit_o[k, e] = o.niter
            eq[k, e] = np.abs(o.get("qpos") - hq[e]).max(); ev[k, e] = np.abs(wv - hv[e]).max() / max(1.0, np.abs(wv).max())
            if obj:
                eo[k, e] = max(np.abs(o.get_object(slot)[0] - hob[e, 7 * oi: 7 * oi + 7]).max() for slot, oi in enumerate(sorted(S["objects"][e]))) if S["objects"][e] else 0.0
            if obj:
                c = o.contacts_full()
                so, sh = sorted(zip(c["body"].tolist(), c["b2"].tolist())), sorted(zip(hc[e]["body"].tolist(), hc[e]["b2"].tolist()))
                cb, cp, hk = list(zip(c["body"].tolist(), c["b2"].tolist())), c["pos"], list(zip(hc[e]["body"].tolist(), hc[e]["b2"].tolist()))
            else:
                ob, op, _ = o.contacts()
                so, sh = sorted(ob.tolist()), sorted(hc[e]["body"].tolist())
                cb, cp, hk = ob.tolist(), op, hc[e]["body"].tolist()
            differ[k, e] = so != sh; ncon[k, e] = len(so)
            if so != sh and follow_flips and len(flips) < 64:
                # the rest of the control step from each side's OWN state after the flipped substep (a 1-env simulator / a second oracle, so that the walk
                # along the oracle's trajectory is not disturbed); both sides are restarted the same way (set_state + forward)
                if side is None:
                    side = KpSim(KpModel(STEP_KPM) if obj else KpModel(), 1)
                    o2 = OracleSim(kpm=STEP_KPM, ls_exact=LS_EXACT) if obj else OracleSim(ls_exact=LS_EXACT)
                if obj:
                    b1 = S["blk"][e:e + 1].copy(); bv1 = np.zeros((1, 30))
                    o2.clear_objects()
                    for slot, oi in enumerate(sorted(S["objects"][e])):
                        oq, ov = o.get_object(slot)
                        o2.set_object(slot, kpm, oi, oq, ov)
                        b1[0, 7 * oi: 7 * oi + 7] = hob[e, 7 * oi: 7 * oi + 7]
                    hobv = sim.get("obj_qvel").double().cpu().numpy()
                    for oi in sorted(S["objects"][e]):
                        bv1[0, 6 * oi: 6 * oi + 6] = hobv[e, 6 * oi: 6 * oi + 6]
                    side.set_objects(dev(b1)); side.set_obj_state(dev(b1), dev(bv1))
                side.set_state(dev(hq[e:e + 1]), dev(hv[e:e + 1])); side.set_target(dev(S["target"][e:e + 1]))
                side.step_ctrl(dev(S["action"][e:e + 1]), 14)
                o2.reset(o.get("qpos"), wv)
                o2.do_simulation(S["action"][e], S["target"][e], 14)
                end = float(np.abs(o2.get("qpos") - side.get("qpos")[0].double().cpu().numpy()).max())
                flips.append((int(k), int(e), float(eq[k, e]), end))
            if so != sh:
                # which rule's threshold does the difference sit on?  'margin': a one-sided contact within 1e-6 of dist == margin; 'support': the hull's first
                # contact (mjc_PlaneConvex's support vertex) is another vertex at the same height, which also changes the neighbours that follow it;
                # 'separation': a one-sided neighbour vertex within 1e-5 of the tolplanemesh x rbound distance from the first contact; 'other': none of these
                od = (c["dist"] if obj else o.contacts()[2]); ol = (cb if obj else ob.tolist())
                hd, hp = hc[e]["dist"], hc[e]["pos"]
                kinds = set()
                for key in set(ol) | set(hk):
                    O = [(od[i], cp[i]) for i in range(len(ol)) if ol[i] == key]; H = [(hd[j], hp[j]) for j in range(len(hk)) if hk[j] == key]
                    if len(O) == len(H):
                        continue
                    floor_pair = (not obj) or key[1] == -1
                    body = key[0] if obj else key
                    if floor_pair and body < 24 and O and H and np.abs(O[0][1] - H[0][1]).max() > 1e-4 and abs(O[0][0] - H[0][0]) < 1e-6:
                        kinds.add("support"); continue
                    long_, short_ = (O, H) if len(O) > len(H) else (H, O)
                    lone = [x for x in long_ if not any(np.abs(x[1] - y[1]).max() < 1e-4 for y in short_)] or long_[len(short_):]
                    for d, pp in lone:
                        if abs(d - MARGIN) < 1e-6:
                            kinds.add("margin")
                        elif floor_pair and body < 24 and long_ and abs(np.linalg.norm(np.array([pp[0], pp[1], d]) - long_[0][1]) - PM_TOL * RBOUND[body]) < 1e-5:
                            kinds.add("separation")
                        else:
                            kinds.add("other")
                klass[k, e] = "+".join(sorted(kinds)) or "other"
            if so == sh and len(so):
                # the same entities on both sides: are they the same POINTS?  (two hull vertices level to 1e-8 -- a flat sole -- are one contact with two
                # possible positions centimetres apart; which one is "the support vertex" is decided by the last bit of the kinematics)
                used = set()
                for i in range(len(cb)):
                    cand = [j for j in range(len(hk)) if j not in used and hk[j] == cb[i]]
                    j = min(cand, key=lambda jj: np.abs(hc[e]["pos"][jj] - cp[i]).max()); used.add(j)
                    if np.abs(hc[e]["pos"][j] - cp[i]).max() > 1e-4:
                        vertex[k, e] = True
                    elif obj:            # the same point: how far apart are the two normals (MPR's direction for the hull - primitive pairs) and distances?
                        nang[k, e] = max(nang[k, e], float(np.arctan2(np.linalg.norm(np.cross(c["normal"][i], hc[e]["normal"][j])), np.dot(c["normal"][i], hc[e]["normal"][j]))))     # not acos(dot): the hip normal is unit to 1e-7 only
                        ddist[k, e] = max(ddist[k, e], abs(c["dist"][i] - hc[e]["dist"][j]))
    return dict(eq=eq, ev=ev, eo=eo, differ=differ, ncon=ncon, scenes=S, seed=seed, nit_o=nit_o, nit_h=nit_h, vertex=vertex, klass=klass, nang=nang, ddist=ddist, flips=flips)


def summary(R):
    """the figures bench.py prints as `parity_live` / the committed logs hold, from run()'s arrays"""
    err = np.maximum(R["eq"], R["eo"])
    same = ~R["differ"] & ~R["vertex"]
    fl = R.get("flips", [])
    return {"substeps": int(err.size), "contact_set_diffs": int(R["differ"].sum()), "same_entities_other_hull_vertex": int(R["vertex"].sum()),
            "max_same_set_dqpos": float(err[same].max()) if same.any() else None, "median_dqpos": float(np.median(err)), "p99_dqpos": float(np.quantile(err, .99)),
            "flips_followed": len(fl), "flip_one_substep_dqpos_max": max((f[2] for f in fl), default=None),
            "flip_dqpos_after_one_control_step_max": max((f[3] for f in fl), default=None),
            "flip_dqpos_after_one_control_step_median": float(np.median([f[3] for f in fl])) if fl else None}


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "floor"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else None
    nsub = int(sys.argv[4]) if len(sys.argv) > 4 else 45
    R = run(mode, n, seed, nsub)
    eq, ev, eo, differ, ncon, S, seed = (R[k] for k in ("eq", "ev", "eo", "differ", "ncon", "scenes", "seed"))
    err = np.maximum(eq, eo)
    vertex = R["vertex"]
    same = ~differ & ~vertex
    print(f"{mode}{' [oracle with the exact line search]' if LS_EXACT else ''}: {n} scenes (seed {seed}) x {nsub} substeps, every substep from a common fp32-rounded state: one-substep |dqpos| median {np.median(err):.1e} p99 {np.quantile(err, .99):.1e} "
          f"max {err.max():.1e}; rel |dqvel| max {ev.max():.1e}; contacts mean {ncon.mean():.1f} max {ncon.max()}")
    print(f"   substeps whose contact sets differ between the two sides at the same state: {int(differ.sum())} of {differ.size}"
          + (f" (their one-substep |dqpos| max {err[differ].max():.1e})" if differ.any() else "")
          + (" [" + ", ".join(f"{nm}: {int((R['klass'] == nm).sum())}" for nm in sorted(set(R['klass'][differ].tolist()))) + "]" if differ.any() else "")
          + f"; same entities but another vertex of a hull at the same height (to 1e-7): {int(vertex.sum())}" + (f" (|dqpos| max {err[vertex].max():.1e})" if vertex.any() else "")
          + f"; with the same contact points: max |dqpos| {err[same].max():.1e}, above 1e-6: {int((err[same] > 1e-6).sum())}, above 1e-5: {int((err[same] > 1e-5).sum())}")
    if R["flips"]:
        fl = np.array([(f[2], f[3]) for f in R["flips"]])
        print(f"   knife-edge flips followed for the rest of their control step (14 more substeps, both sides free-running from their own states): {len(fl)}; "
              f"|dqpos| right after the flipped substep median {np.median(fl[:, 0]):.1e} max {fl[:, 0].max():.1e} -> at the end of the control step median {np.median(fl[:, 1]):.1e} max {fl[:, 1].max():.1e}; "
              f"amplified (end > 2 x start) in {int((fl[:, 1] > 2 * fl[:, 0]).sum())}, damped (end < start / 2) in {int((fl[:, 1] < 0.5 * fl[:, 0]).sum())}")
    if R["nang"].any():
        print(f"   matched contacts of the same point: |d dist| max {R['ddist'].max():.1e}; angle between the two normals max {R['nang'].max():.1e} rad, above 1e-4 rad in {int((R['nang'] > 1e-4).sum())} substeps, above 1e-3 in {int((R['nang'] > 1e-3).sum())}")
    order = np.dstack(np.unravel_index(np.argsort(-err, axis=None)[:6], err.shape))[0]
    for k, e in order:
        print(f"   substep {k:2d} scene {e:3d} (kind {S['kind'][e]}, objects {sorted(S['objects'][e])}): |dqpos| {eq[k, e]:.1e} object {eo[k, e]:.1e} rel |dqvel| {ev[k, e]:.1e} contacts {ncon[k, e]} Newton iterations oracle {R['nit_o'][k, e]} hip {R['nit_h'][k, e]}"
              f"{'  contact sets differ' if differ[k, e] else ('  another vertex at the same height' if vertex[k, e] else '')}")
