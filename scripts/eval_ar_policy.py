#!/usr/bin/env python
"""Counterpart of the reference's scripts/eval_ar_policy.py --mode stats on the batched engine: every sequence of the test
set is one environment; writes `<iter>_<data_file>_coverage(.|_full).pkl` in the reference's format.

The reference's test sets are not part of its repository; without `--data` this evaluates synthetic standing sequences.

    python scripts/eval_ar_policy.py --num_seq 256 [--ckpt results/.../iter_0750.p] [--fail_safe] [--ar_mode]
    python scripts/eval_ar_policy.py --data sample_data/features/mocap_annotations.p --ckpt ... [--wild]       # every take of a feature file, played whole
    python scripts/eval_ar_policy.py --cfg kin_poly --config_root /path/to/KinPoly --iter 750 [--data test]     # the reference's command line (--mode stats)

With --cfg the feature file, the take list (`meta/<meta_id>.yml`), the checkpoint (`models_policy/iter_%04d.p`) and the result directory come from
the reference's yml (kinpoly_amd/config.py), and the coverage pickles land where eval_pose_all.py looks for them.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_seq", type=int, default=256)
    ap.add_argument("--clip_len", type=int, default=100)
    ap.add_argument("--ckpt", type=str, default="")
    ap.add_argument("--iter", type=int, default=0)
    ap.add_argument("--fail_safe", action="store_true")
    ap.add_argument("--ar_mode", action="store_true")
    ap.add_argument("--wild", action="store_true")
    ap.add_argument("--result_dir", type=str, default="results/eval")
    ap.add_argument("--data_file", type=str, default="synthetic_standing")
    ap.add_argument("--data", type=str, default="", help="feature file in the reference's schema, or train / test with --cfg")
    ap.add_argument("--metrics", action="store_true", help="with --data: root_dist / mpjpe / accel_dist / vel_dist / head_dist / succ over the takes (eval_pose_all.py's kinematic metrics)")
    ap.add_argument("--cfg", type=str, default=None)
    ap.add_argument("--config_root", type=str, default=None)
    args = ap.parse_args()
    from kinpoly_amd import checkpoint as ck
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.context import PolicyARContext, TrajARNet
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    from kinpoly_amd.evaluate import run_sequences, write_coverage
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    n, T = args.num_seq, args.clip_len
    torch.manual_seed(0)
    cfg, takes = None, None
    if args.cfg:
        from kinpoly_amd.config import Config
        if args.config_root:
            os.chdir(args.config_root)
        cfg = Config(args.cfg, wild=args.wild)
        mode = args.data if args.data in ("train", "test") else "test"
        takes = cfg.takes[mode] or None
        args.data = cfg.feature_path() if args.data in ("", "train", "test") else args.data
        args.result_dir, args.data_file, T = cfg.result_dir, cfg.data_file, int(cfg.fr_num)
        if args.iter > 0 and not args.ckpt:
            args.ckpt = cfg.checkpoint_path(args.iter)
    env = BatchedHumanoidAREnv(n, 0, mode="test", wild=args.wild, ar_mode=args.ar_mode, seed=0)
    if cfg is not None:
        cfg.apply_reward_weights(env)
    net = TrajARNet(log_std=cfg.policy_specs["log_std"] if cfg else -3.2).to(env.device)
    if args.ckpt:
        cp = ck.load_checkpoint(args.ckpt)
        net.load_state_dict(ck.split_policy_dict(cp["policy_dict"]), strict=False)
    if args.data:                                   # every take of the feature file, whole, env.n at a time (run_seq over data_loader.iter_seq)
        from kinpoly_amd import dataset as D
        from kinpoly_amd.evaluate import eval_dataset
        ds = D.StateARDataset(args.data, takes=takes, data_mode="test", fr_num=T, wild=args.wild, seed=0, device=env.device)
        builder = PolicyARContext(net, kpsim.KpSim(env.model, n, 0), smooth=bool(cfg.smooth) if cfg else True, keep_context_feat=False)
        res = eval_dataset(env, net, builder, ds, fail_safe=args.fail_safe)
        cov = write_coverage(res, args.result_dir, args.iter, args.data_file if args.cfg else os.path.splitext(os.path.basename(args.data))[0])
        pct = np.array([r["percent"] for r in res.values()])
        print(f"Coverage of {cov} out of {len(res)} | mean percent {pct.mean():.3f} | fail-safe used in {sum(r['fail_safe'] for r in res.values())}")
        if args.metrics:                            # the kinematic metrics of scripts/eval_pose_all.py --mode stats (kinpoly_amd/metrics.py)
            from kinpoly_amd import metrics as M
            from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
            from kinpoly_amd.supervised import TorchFK
            kpm = read_kpm(DEFAULT_KPM)
            tfk = TorchFK(kpm["body_pos"], kpm["body_parent"], "cpu", dtype=torch.float64)
            gt = {k: {"qpos": ds.data["qpos"][i].double().numpy(), "head_pose": ds.data["head_pose"][i].double().numpy()} for i, k in enumerate(ds.takes)}
            m = M.coverage_metrics(res, gt, lambda q: tuple(x.numpy() for x in tfk.chain_torch(torch.as_tensor(q))))
            print("".join(f"{k}:{v:.3f} \t " for k, v in m.items() if k != "per_take"))
        return
    g = torch.Generator().manual_seed(0)
    ctx = standing_context(n, T, std["qpos"], std["qvel"], env.sim, (torch.rand(n, generator=g) * 2 - 1) * np.pi)
    ctx["obj_pose"] = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=env.device).repeat(n, T, 1)
    ctx = PolicyARContext(net, kpsim.KpSim(env.model, n, 0), smooth=True).init_context(ctx)
    env.load_context(ctx)
    keys = [f"standing-{i:05d}" for i in range(n)]
    res = run_sequences(env, net, keys, fail_safe=args.fail_safe)
    cov = write_coverage(res, args.result_dir, args.iter, args.data_file)
    pct = np.array([r["percent"] for r in res.values()])
    print(f"Coverage of {cov} out of {n} | mean percent {pct.mean():.3f} | fail-safe used in {sum(r['fail_safe'] for r in res.values())}")


if __name__ == "__main__":
    main()
