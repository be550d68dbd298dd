"""Cycles of the three kinds of solve inside the floor kernel's Newton iteration (instrumented build: python tools/micro/flip_instr.py
tools/micro/bin/libkinpoly_sim_factcost.so cost): what a rank-k update of the factorisation would have to beat."""
import os, sys
os.environ["KP_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from kinpoly_amd import sim as _sim
_sim.load_library(os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_factcost.so"))
import bench
for wl in ("tracked", "random_init"):
    rec, env, policy, sampler, std = bench.run_workload(wl, 0, 4, 64, 12, 6)
    t = env.sim.phase_cycles_env().sum(0)
    print(f"{wl}: first factorisation + solve of a substep {t[0] / max(t[1], 1):.0f} cycles x {t[1] / env.n:.2f} per control step; re-factorisation (dirty levels) + solve "
          f"{t[2] / max(t[3], 1):.0f} x {t[3] / env.n:.2f}; solve through standing factors (aba_resolve) {t[4] / max(t[5], 1):.0f} x {t[5] / env.n:.2f}", flush=True)
