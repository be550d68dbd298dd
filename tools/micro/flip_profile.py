"""How the Newton solve's active set changes between iterations, on bench.py's workloads (instrumented build: tools/micro/flip_instr.py)."""
import os, sys
os.environ["KP_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kinpoly_amd import sim as _sim
_sim.load_library(os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_flips.so"))
import bench
for wl in ("tracked", "random_init", "objects"):
    rec, env, policy, sampler, std = bench.run_workload(wl, 0, 4, 64, 12, 6)
    pe = env.sim.phase_cycles_env()                 # [N, 8] of the last control step
    d = np.asarray(rec["diag"])
    tot = pe.sum(0)
    ref = max(tot[0], 1.0)
    nfact, nit = (d[:, 3] >> 8).sum(), d[:, 1].sum()
    print(f"{wl}: per control step and env: newton iterations {nit / len(d):.2f}, factorisations {nfact / len(d):.2f} of which after a substep's first (active set changed) {tot[0] / len(d):.2f}; "
          f"rows flipped per such re-factorisation: mean {tot[6] / ref:.2f}; share with 1 row {tot[1] / ref:.3f}, 2 rows {tot[2] / ref:.3f}, 3-4 rows {tot[3] / ref:.3f}, >= 5 rows {tot[4] / ref:.3f}; "
          f"only object-side contacts flipped {tot[5] / ref:.3f}", flush=True)
    if wl == "objects":
        cost = pe[:, 7]
        heavy = cost >= np.percentile(cost, 99)
        th = pe[heavy].sum(0); rh = max(th[0], 1.0)
        print(f"   heaviest 1 % of the envs: re-factorisations {th[0] / heavy.sum():.2f} per control step; 1 row {th[1] / rh:.3f}, 2 rows {th[2] / rh:.3f}, 3-4 {th[3] / rh:.3f}, >= 5 {th[4] / rh:.3f}; "
              f"only object-side contacts flipped {th[5] / rh:.3f}; rows per re-factorisation {th[6] / rh:.2f}", flush=True)
    del env, sampler, policy
    torch.cuda.empty_cache()

# what a rank-k update would compete with (instrumented build `cost`: python tools/micro/flip_instr.py tools/micro/bin/libkinpoly_sim_factcost.so cost)
cost_lib = os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_factcost.so")
if os.path.exists(cost_lib) and os.environ.get("KP_FLIP_COST") == "1":
    pass
