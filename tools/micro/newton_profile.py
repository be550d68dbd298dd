"""Newton sub-phase cycles on the bench's tracked workload (needs the temporary NP() instrumentation build)."""
import os, sys
os.environ["KP_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kinpoly_amd import sim as _sim
_sim.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libkinpoly_sim_newton.so"))   # tools/micro/newton_instr.py builds it
import bench
for wl in ("tracked", "random_init"):
    rec, env, policy, sampler, std = bench.run_workload(wl, 0, 4, 64, 12, 6)
    pc = env.sim.phase_cycles()
    d = rec["diag"]
    import numpy as np
    d = np.asarray(d)
    det = os.environ.get("KP_NEWTON_DETAIL")
    names = ("w_stage1", "w_subtree", "w_project", "g2_active_sums", "first_clean", "rest", "-", "total") if det == "gradient" else ("lev_setup", "contact_inertia", "elim3_store", "clean_half", "forward", "rest", "-", "total") if det == "factor" else ("init", "gradient", "factor_solve", "rows_quad", "linesearch", "update_cost", "ls_iters", "total")
    print(wl, "launch ms %.3f" % (rec["kern_s"] * 1e3), {n: int(v) for n, v in zip(names, pc.values())}, "newton it/ctrl-step", d[:, 1].mean(), "nfact", (d[:, 3] >> 8).mean(), flush=True)
