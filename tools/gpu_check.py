"""First-contact GPU check: HIP sim vs fp64 oracle on a handful of envs + raw timings.

Run on the GPU box:  python tools/gpu_check.py  [> gpurun_out/gpu_check.log]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinpoly_amd.sim import KpModel, KpSim  # noqa: E402
from oracle.kpo import OracleSim  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))


def make_states(n, seed, lift=0.0, vel=0.5, noise=0.2):
    rng = np.random.default_rng(seed)
    qpos = np.tile(std["qpos"], (n, 1))
    qpos[:, 2] += lift
    qpos[:, 7:] += np.clip(rng.normal(size=(n, 69)) * noise, -np.pi, np.pi)
    qvel = rng.normal(size=(n, 75)) * vel
    qvel[:, :3] = rng.normal(size=(n, 3))
    return qpos, qvel


def compare(name, n, nsub, contact, lift, action_scale, steps=1, threads=64, stale=1):
    qpos, qvel = make_states(n, 1, lift=lift)
    rng = np.random.default_rng(2)
    action = rng.normal(size=(n, 75)) * action_scale
    target = np.tile(std["qpos"], (n, 1))
    model = KpModel(contact=contact, limits=1, threads_per_env=threads, stale_kinematics=stale)
    sim = KpSim(model, n)
    dev = sim.device
    sim.set_state(torch.tensor(qpos, dtype=torch.float32, device=dev), torch.tensor(qvel, dtype=torch.float32, device=dev))
    sim.set_target(torch.tensor(target, dtype=torch.float32, device=dev))
    a = torch.tensor(action, dtype=torch.float32, device=dev)
    for _ in range(steps):
        sim.step_ctrl(a, nsub)
    gq = sim.get("qpos").cpu().numpy().astype(np.float64)
    gv = sim.get("qvel").cpu().numpy().astype(np.float64)
    gx = sim.get("xpos").cpu().numpy().astype(np.float64)
    dg = sim.diag()
    errq, errv, errx = [], [], []
    o = OracleSim(contact=bool(contact), limits=True)
    for e in range(n):
        o.reset(qpos[e], qvel[e])
        for _ in range(steps):
            o.do_simulation(action[e], target[e], nsub)
        errq.append(np.abs(o.get("qpos") - gq[e]).max())
        errv.append(np.abs(o.get("qvel") - gv[e]).max())
        errx.append(np.abs(o.get("xpos") - gx[e]).max())
    print(f"[{name}] n={n} nsub={nsub}x{steps} thr={threads}: max|dqpos|={max(errq):.3e} median={np.median(errq):.3e}  "
          f"max|dqvel|={max(errv):.3e}  max|dxpos|={max(errx):.3e}  ncon(max)={(dg[:, 3] & 255).max()} iters(mean)={dg[:, 1].mean():.1f} flags={dg[:, 2].max()}",
          flush=True)


def timing(n, threads, contact, lift, steps=5):
    qpos, qvel = make_states(n, 3, lift=lift)
    model = KpModel(contact=contact, threads_per_env=threads)
    sim = KpSim(model, n)
    dev = sim.device
    q = torch.tensor(qpos, dtype=torch.float32, device=dev); v = torch.tensor(qvel, dtype=torch.float32, device=dev)
    sim.set_state(q, v); sim.set_target(q.clone())
    a = torch.zeros((n, 75), device=dev)
    sim.step_ctrl(a, 15)
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        sim.step_ctrl(a, 15)
        ts.append(sim.last_step_seconds())
    dg = sim.diag()
    print(f"[timing] n={n} thr={threads} contact={contact} lift={lift}: {np.mean(ts)*1e3:.3f} ms/control-step "
          f"({n/np.mean(ts):.0f} env-steps/s physics only) ncon(mean)={dg[:,0].mean():.1f} iters/substep={dg[:,1].mean()/15:.2f} factorisations/substep={(dg[:,3] >> 8).mean()/15:.2f}", flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), "lds/env", KpModel().get_option("lds_bytes_per_env"), flush=True)
    compare("freefall 1 substep", 16, 1, 0, 10.0, 0.0)
    compare("freefall 15 substeps", 16, 15, 0, 10.0, 0.0)
    compare("freefall+SPD 15", 16, 15, 0, 10.0, 0.3)
    compare("contact 1 substep", 16, 1, 1, 0.0, 0.0)
    compare("contact 15 substeps", 16, 15, 1, 0.0, 0.3)
    compare("contact 15x10", 16, 15, 1, 0.0, 0.3, steps=10)
    for thr in (128, 256):
        compare("contact 15 substeps", 16, 15, 1, 0.0, 0.3, threads=thr)
    for thr in (64, 128, 256):
        timing(4096, thr, 0, 10.0)
        timing(4096, thr, 1, 0.0)
