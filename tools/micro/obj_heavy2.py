"""Fine sub-phase profile of solve_constraints_obj on the costliest and the median envs of bench.py's `objects` workload (builds: obj_instr2.py A / B).
   KP_FINE=A|B python tools/micro/obj_heavy2.py"""
import os, sys
os.environ["KP_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from kinpoly_amd import sim as _sim
which = os.environ.get("KP_FINE", "A")
_sim.load_library(os.path.join(ROOT, "tools", "micro", "bin", f"libkinpoly_sim_objfine{which}.so"))
import bench
names = {"A": ("con_prepare", "wrench_project", "obj_gradient+|g|", "obj_hessian", "Hhh aba_solve", "coupling rhs+schur", "everything else", "total"),
         "B": ("up to dense", "dense_solve", "coupling wrench+backsub pass", "eval_rows", "quad forms", "line_search", "update+cost+active set", "total")}[which]
rec, env, policy, sampler, std = bench.run_workload("objects", 0, 4, 64, 8, 4)
pe = env.sim.phase_cycles_env()
d = env.sim.diag()
order = np.argsort(-pe[:, 7])
for tag, idx in (("top 32", order[:32]), ("top 33..256", order[32:256]), ("median 256", order[len(order) // 2 - 128: len(order) // 2 + 128])):
    it = d[idx, 1].mean() / 15
    print(f"{which} {tag}: per ITERATION", {n: int(pe[idx, k].mean() / 15 / it) for k, n in enumerate(names[:7])}, "| per substep total %d" % int(pe[idx, 7].mean() / 15),
          "contacts %.1f newton it/substep %.2f nfact/substep %.2f" % (d[idx, 0].mean(), it, (d[idx, 3] >> 8).mean() / 15), flush=True)
