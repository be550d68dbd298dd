"""The tail of the objects workload's launch: the envs that take longest in the last control step -- action class, contacts, Newton iterations,
factorisations, object state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kinpoly_amd import sim as _sim
if os.environ.get("KP_OBJ_NEWTON") == "1":
    _sim.load_library(os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_objnewton.so"))
import bench
steps = int(os.environ.get("STEPS", "40"))
rec, env, policy, sampler, std = bench.run_workload("objects", 0, 4, 64, steps, 10)
d = np.asarray(rec["diag"])
cost = env.sim.launch_cost().astype(np.float64)
cls = env.ctx["action_one_hot"][env.row.long()].argmax(1).cpu().numpy()
names = ("sit", "push", "avoid", "step")
print("launch ms %.3f  cost unit per env: mean %.0f median %.0f p99 %.0f max %.0f" % (rec["kern_s"] * 1e3, cost.mean(), np.median(cost), np.percentile(cost, 99), cost.max()))
oq = env.sim.get("obj_qpos").cpu().numpy(); ov = env.sim.get("obj_qvel").cpu().numpy()
qp = env.sim.get("qpos").cpu().numpy()
order = np.argsort(-cost)
if os.environ.get("KP_PROFILE") == "1":
    pe = env.sim.phase_cycles_env() / 15.0
    newton = os.environ.get("KP_OBJ_NEWTON") == "1"
    names8 = ("init+cand", "gradient", "Hhh_factor", "schur_cols", "dense+backsub", "rows+ls", "update+cost", "total") if newton else ("stable-PD", "kinematics", "collision", "constraints", "smooth", "newton", "integrate", "total")
    top = order[:40]
    print("cycles per substep, mean over the 40 costliest envs:", {n_: int(v) for n_, v in zip(names8, pe[top].mean(0))})
    print("cycles per substep, median env:", {n_: int(v) for n_, v in zip(names8, np.median(pe, 0))})
    for e in np.argsort(-pe[:, 7])[:4]:
        print(f"cycles per substep, env {e}:", {n_: int(v) for n_, v in zip(names8, pe[e])}, "newton it/substep %.1f contacts %d" % (d[e, 1] / 15, d[e, 0]))
for e in order[:16]:
    a = cls[e]
    print(f"env {e} class {names[a]} cost {cost[e] / np.median(cost):.1f}x median  contacts {d[e, 0]} (max {d[e, 3] & 255}) newton it/substep {d[e, 1] / 15:.1f} nfact/substep {(d[e, 3] >> 8) / 15:.1f} cap hits {d[e, 2] >> 8}"
          f" cur_t {int(env.cur_t[e])} root z {qp[e, 2]:.2f} obj speed {np.abs(ov[e]).max():.2f}")
for a in range(4):
    m = cls == a
    print(names[a], "n", m.sum(), "cost/median mean %.2f p99 %.2f max %.2f" % (cost[m].mean() / np.median(cost), np.percentile(cost[m], 99) / np.median(cost), cost[m].max() / np.median(cost)),
          "newton it/substep mean %.2f p99 %.2f max %.2f" % (d[m, 1].mean() / 15, np.percentile(d[m, 1], 99) / 15, d[m, 1].max() / 15))
