#!/usr/bin/env python
"""Env-steps/s of the bench's rollout workloads against the number of envs per GPU (bench.py measures 4096, BASELINE's figure).

The control-step launch ends on its costliest env: that chain does not grow with the batch, the work behind it does, so the
share of the launch the tail costs shrinks as envs are added.  This prints, per workload and env count: env-steps/s, ms per step, launch ms.

    python tools/envs_sweep.py [workloads] [env counts]      e.g.  tools/envs_sweep.py tracked,objects 2048,4096,8192,16384
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    wls = (sys.argv[1] if len(sys.argv) > 1 else "tracked,objects").split(",")
    counts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2048,4096,8192,16384").split(",")]
    torch.cuda.set_device(0)
    for wl in wls:
        for n in counts:
            bench.ENVS_PER_GPU = n
            rec, env, policy, sampler, std = bench.run_workload(wl, 0, 4, 64, 40, 15)
            dg = rec["diag"]
            print(json.dumps({"workload": wl, "envs": n, "env_steps_per_s": round(n * 40 / rec["elapsed"]), "ms_per_step": round(rec["elapsed"] / 40 * 1e3, 3),
                              "launch_ms": round(rec["kern_s"] * 1e3, 3), "contacts_mean": round(float(dg[:, 0].mean()), 2),
                              "episodes_ended_per_step_frac": round(rec["n_done"] / (n * 40), 4)}), flush=True)
            del env, policy, sampler
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
