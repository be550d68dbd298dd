#!/usr/bin/env python
"""Live pin of the physics against MuJoCo (dormant: prints `null` until `import mujoco` / `import mujoco_py` succeeds).

    python tools/mujoco_pin.py [--free_fall 1500] [--contact 150] [--no_hip] [--write_xml scene.xml]

The harness itself is test infrastructure (tests/mj_pin.py: MJCF from the compiled blob, model comparison, BASELINE configs[1] / configs[2]
stepped on MuJoCo, on the fp64 oracle and on the HIP simulator); this is its command line.  Reference call sites: uhc/envs/humanoid_im.py:527,
uhc/khrylib/rl/envs/common/mujoco_env.py:23-24."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(1, os.path.join(ROOT, "tests"))


def hip_trajectory(q0, v0, actions, target):
    """the product's qpos after every control step (one env of a KpSim), same inputs as the other backends"""
    import numpy as np
    import torch
    from kinpoly_amd import sim as kpsim
    s = kpsim.KpSim(kpsim.KpModel(), 1, 0)
    dev = lambda a: torch.tensor(np.asarray(a)[None], dtype=torch.float32, device=s.device)      # noqa: E731
    s.set_state(dev(q0), dev(v0)); s.set_target(dev(target))
    out = []
    for a in actions:
        s.step_ctrl(dev(a), 15)
        out.append(s.get("qpos")[0].double().cpu().numpy())
    return np.stack(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--free_fall", type=int, default=1500); ap.add_argument("--contact", type=int, default=150)
    ap.add_argument("--no_hip", action="store_true"); ap.add_argument("--write_xml", type=str, default="")
    ap.add_argument("--report", action="store_true", help="print the model-array comparison table the pin asserts on, entry by entry (worst entry, its body / dof, both values). "
                    "Without a MuJoCo binding the fp64 oracle's own derived arrays stand in for the backend's -- labelled -- so that the table's code path is exercised")
    args = ap.parse_args()
    import mj_pin as MP
    if args.report:
        from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
        kpm = read_kpm(DEFAULT_KPM)
        found = MP.find_mujoco()
        if found is not None:
            backend, who = MP.open_backend(kpm), f"{found[0]} {getattr(found[1], '__version__', '?')}"
            arrays = backend.model_arrays()
        else:
            who = "NO MuJoCo binding importable: the blob against itself (every row must read ok / 0) -- this run checks the report, not the physics"
            arrays = {k: kpm[k] for k in ("body_mass", "body_ipos", "body_inertia", "body_invweight0", "dof_invweight0")}
            bp = kpm["body_pos"].reshape(-1, 3).copy(); bp[0] = kpm["body_gpos0"].reshape(-1, 3)[0]; arrays["body_pos"] = bp.reshape(-1)
        print(f"model arrays: backend = {who}")
        print(MP.format_model_table(MP.model_table(arrays, kpm)))
    if args.write_xml:
        from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
        open(args.write_xml, "w").write(MP.mjcf_from_kpm(read_kpm(DEFAULT_KPM)))
    hip = None
    if not args.no_hip:
        import torch
        hip = hip_trajectory if torch.cuda.is_available() else None
    print(json.dumps({"mujoco_pin": MP.pin_report(n_free_fall=args.free_fall, n_contact=args.contact, hip=hip)}))


if __name__ == "__main__":
    main()
