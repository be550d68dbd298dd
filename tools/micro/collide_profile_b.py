"""collide() of the objects workload's costliest envs, variant B of collide_instr.py: broad phase / per-hull set-up / plane cull shares."""
import os, sys
os.environ["KP_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from kinpoly_amd import sim as _sim
_sim.load_library(os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_collideB.so"))
import bench
rec, env, policy, sampler, std = bench.run_workload("objects", 0, 4, 64, 40, 10)
pe = env.sim.phase_cycles_env() / 15.0
order = np.argsort(-pe[:, 7])
for tag, idx in (("40 costliest", order[:40]), ("median 256", order[len(order) // 2 - 128: len(order) // 2 + 128])):
    m = pe[idx].mean(0)
    print(f"{tag}: per substep: collide() {m[5]:.0f} cycles: broad phase {m[0]:.0f}, hulls visited {m[4]:.1f} with {m[1]:.0f} cycles of set-up, pairs at the plane cull {m[3]:.1f} with {m[2]:.0f} cycles; substep total {m[7]:.0f}", flush=True)
