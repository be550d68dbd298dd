"""Where the object kernel's cycles go.  (1) KP_PROFILE substep phases of the shipped library on the object scenes of tools/obj_bench.py and on
bench.py's `objects` workload, with the per-env cost of the last launch grouped by action class; (2) with KP_OBJ_NEWTON=1 the sub-phases of
solve_constraints_obj from the instrumented build (tools/micro/obj_instr.py)."""
import os, sys
os.environ["KP_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kinpoly_amd import sim as _sim
newton = os.environ.get("KP_OBJ_NEWTON") == "1"
if newton:
    _sim.load_library(os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_objnewton.so"))
import bench
names = ("init+cand", "gradient", "Hhh_factor", "schur_cols", "dense+backsub", "rows+ls", "update+cost", "total") if newton else \
        ("stable-PD", "kinematics", "collision", "constraints", "smooth", "newton", "integrate", "total")
rec, env, policy, sampler, std = bench.run_workload("objects", 0, 4, 64, 8, 4)
pc = env.sim.phase_cycles()
d = np.asarray(rec["diag"])
print("objects workload (one workgroup per env: KP_PROFILE turns the job queue off) launch ms %.3f" % (rec["kern_s"] * 1e3),
      {n: int(v / 15) for n, v in zip(names, pc.values())}, "per substep; newton it/substep %.2f nfact/substep %.2f contacts %.1f" % (d[:, 1].mean() / 15, (d[:, 3] >> 8).mean() / 15, d[:, 0].mean()), flush=True)
cost = env.sim.launch_cost().astype(np.float64) * 1024
cls = env.ctx["action_one_hot"][env.row.long()].argmax(1).cpu().numpy()
for a, nm in enumerate(("sit", "push", "avoid", "step")):
    m = cls == a
    print(f"   class {nm}: {m.sum()} envs, cycles per control step mean {cost[m].mean() / 1e6:.2f} M  p90 {np.percentile(cost[m], 90) / 1e6:.2f} M  max {cost[m].max() / 1e6:.2f} M; "
          f"contacts {d[m, 0].mean():.1f}, newton it/substep {d[m, 1].mean() / 15:.2f}, nfact/substep {(d[m, 3] >> 8).mean() / 15:.2f}", flush=True)
del env, sampler, policy
torch.cuda.empty_cache()
# the scenes of tools/obj_bench.py
from kinpoly_amd.sim import STEP_KPM, KpModel, KpSim
n, rng = 4096, np.random.default_rng(3)
x0, y0 = std["qpos"][0], std["qpos"][1]
for name, active, lift in (("standing on the step box", {4: [x0, y0, 0.3705, 1, 0, 0, 0]}, 0.341),
                           ("push scene (box on table, 0.45 m ahead)", {1: [x0 + 0.75, y0, 0.921, 1, 0, 0, 0], 2: [x0 + 0.75, y0, 0.7905, 1, 0, 0, 0]}, 0.0),
                           ("step box 2 m away (untouched)", {4: [x0 + 2.0, y0, 0.3705, 1, 0, 0, 0]}, 0.0)):
    blk = np.zeros((n, 35))
    for i in range(5):
        blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
    for oi, pose in active.items():
        blk[:, 7 * oi: 7 * oi + 7] = pose
    sim = KpSim(KpModel(STEP_KPM), n)
    qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 2] += lift; qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.05
    dev = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")      # noqa: E731
    q = dev(qpos)
    sim.set_objects(dev(blk)); sim.set_state(q, dev(rng.normal(size=(n, 75)) * 0.2)); sim.set_target(q.clone())
    a = dev(rng.normal(size=(n, 75)) * 0.1)
    ts = []
    for _ in range(6):
        sim.step_ctrl(a, 15); ts.append(sim.last_step_seconds())
    dg = sim.diag()
    print(f"{name}: launch {np.mean(ts[2:]) * 1e3:.3f} ms", {k: int(v / 15) for k, v in zip(names, sim.phase_cycles().values())},
          "per substep; newton it/substep %.2f nfact/substep %.2f contacts %.1f" % (dg[:, 1].mean() / 15, (dg[:, 3] >> 8).mean() / 15, dg[:, 0].mean()), flush=True)
    del sim
    torch.cuda.empty_cache()
