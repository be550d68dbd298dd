"""End state of a few env-steps of a bench workload, for bit-identity checks between two builds of the library (KP_SIM_LIBRARY):
    KP_SIM_LIBRARY=a.so python tools/micro/lib_ab_state.py objects a.npz ; python tools/micro/lib_ab_state.py objects b.npz ; python tools/micro/lib_ab_state.py cmp a.npz b.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    same = True
    for k in a.files:
        eq = np.array_equal(a[k], b[k])
        same &= eq
        print(f"{k}: {'identical' if eq else 'DIFFERENT, max |d| %.3e' % np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()}")
    print("BIT-IDENTICAL" if same else "NOT bit-identical")
    sys.exit(0)

import torch  # noqa: E402
import bench  # noqa: E402

wl, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
env, policy, sampler, std = bench.build_engine(0, 0, 64, wl)
a_track = None
if wl in ("tracked", "wild_eval"):
    a_track = bench.tracking_action(env)
    sampler.start()
if wl in ("tracked", "objects"):
    bench.stagger_episodes(env, sampler, 0, wl == "objects")
bench.rollout_steps(sampler, steps, a_track, wl == "wild_eval", wl == "objects")
torch.cuda.synchronize()
rec = {"qpos": env.sim.get("qpos").cpu().numpy(), "qvel": env.sim.get("qvel").cpu().numpy(), "diag": env.sim.diag()}
if wl == "objects":
    rec["obj_qpos"], rec["obj_qvel"] = env.sim.get("obj_qpos").cpu().numpy(), env.sim.get("obj_qvel").cpu().numpy()
np.savez(out, **rec)
print(wl, "saved", out, "newton it / substep", rec["diag"][:, 1].mean() / 15)
