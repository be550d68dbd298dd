"""One AgentAR.optimize_policy at 4096 envs x 24 steps for `rocprofv3 --kernel-trace --stats` (the update is 8 : 1 of an iteration):
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_prof/update -o stats -- python tools/update_profile.py
KP_TUNE=1 records TunableOp solutions for every GEMM shape the iteration runs (written to gpurun_out/tune/ on exit)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("KP_TUNE") == "1":
    out = os.path.join(ROOT, "gpurun_out", "tune"); os.makedirs(out, exist_ok=True)
    os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="150",
                      PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="10", PYTORCH_TUNABLEOP_FILENAME=os.path.join(out, "tunableop_update_%d.csv"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kinpoly_amd import sim as kpsim  # noqa: E402
import kinpoly_amd.nets as nets  # noqa: E402
if os.environ.get("KP_TUNE") == "1":
    nets.enable_tuned_gemms = lambda *a, **k: False
    import kinpoly_amd.agent as _ag
    _ag.enable_tuned_gemms = nets.enable_tuned_gemms
from kinpoly_amd.agent import AgentAR  # noqa: E402
from kinpoly_amd.env import standing_context  # noqa: E402

std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
n, T = int(os.environ.get("KP_N", 4096)), int(os.environ.get("KP_T", 24))
fk_sim = kpsim.KpSim(kpsim.KpModel(), n, 0)


def context_fn(m):
    ctx = standing_context(m, 100, std["qpos"], std["qvel"], fk_sim, torch.zeros(m))
    ctx["obj_pose"] = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=ctx["qpos"].device).repeat(m, 100, 1)
    return ctx


agent = AgentAR(n, context_fn, device=0, horizon=T, use_init_context=False, pool_depth=2)
for it in range(int(os.environ.get("KP_ITERS", 2))):
    info = agent.optimize_policy(it)
    print(f"iter {it}: T_sample {info['T_sample']:.3f} s  T_update {info['T_update']:.3f} s  ({info['num_steps']} samples)", flush=True)
