"""collide() of the objects workload's costliest envs after 50 env-steps: how many libccd MPR queries, how many hit, what they cost (build: collide_instr.py)."""
import os, sys
os.environ["KP_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from kinpoly_amd import sim as _sim
_sim.load_library(os.environ.get("KP_COLLIDE_LIB", os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_collide.so")))
import bench
rec, env, policy, sampler, std = bench.run_workload("objects", 0, 4, 64, int(os.environ.get("STEPS", "40")), 10)
pe = env.sim.phase_cycles_env() / 15.0
order = np.argsort(-pe[:, 7])
for tag, idx in (("40 costliest", order[:40]), ("top 41..256", order[40:256]), ("median 256", order[len(order) // 2 - 128: len(order) // 2 + 128])):
    m = pe[idx].mean(0)
    print(f"{tag}: per substep: MPR queries {m[0]:.1f} of which hit {m[1]:.1f}, cycles in them {m[2]:.0f} ({m[2] / max(m[0], 1e-9):.0f} per query); object-object / object-floor narrow phases {m[3]:.0f} "
          f"(box-box calls {m[5]:.2f}, {m[6]:.0f} cycles); floor-hull {m[4]:.0f}; substep total {m[7]:.0f}", flush=True)
