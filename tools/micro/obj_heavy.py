"""Phase profile of the COSTLIEST environments of bench.py's `objects` workload (they bound the launch: DESIGN section 6.6), next to the
median ones.  KP_OBJ_NEWTON=1: sub-phases of solve_constraints_obj from the instrumented build (tools/micro/obj_instr.py)."""
import os, sys
os.environ["KP_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from kinpoly_amd import sim as _sim
newton = os.environ.get("KP_OBJ_NEWTON") == "1"
if newton:
    _sim.load_library(os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_objnewton.so"))
import bench
names = ("init+cand", "gradient", "Hhh_factor", "schur_cols", "dense+backsub", "rows+ls", "update+cost", "total") if newton else \
        ("stable-PD", "kinematics", "collision", "constraints", "smooth", "newton", "integrate", "total")
rec, env, policy, sampler, std = bench.run_workload("objects", 0, 4, 64, 8, 4)
pe = env.sim.phase_cycles_env()            # [N, 8] cycles of the last control step
d = env.sim.diag()
cls = env.ctx["action_one_hot"][env.row.long()].argmax(1).cpu().numpy()
order = np.argsort(-pe[:, 7])
for tag, idx in (("top 32", order[:32]), ("top 33..256", order[32:256]), ("median 256", order[len(order) // 2 - 128: len(order) // 2 + 128])):
    print(f"{tag}: per substep", {n: int(pe[idx, k].mean() / 15) for k, n in enumerate(names)},
          "contacts %.1f newton it/substep %.2f nfact/substep %.2f" % (d[idx, 0].mean(), d[idx, 1].mean() / 15, (d[idx, 3] >> 8).mean() / 15),
          "classes", np.bincount(cls[idx], minlength=4).tolist(), flush=True)
for e in order[:8]:
    print(f"env {e} class {cls[e]}: per substep", {n: int(pe[e, k] / 15) for k, n in enumerate(names)},
          "contacts %d newton it/substep %.2f nfact/substep %.2f cap hits %d" % (d[e, 0], d[e, 1] / 15, (d[e, 3] >> 8) / 15, d[e, 2] >> 8), flush=True)
