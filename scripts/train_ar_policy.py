#!/usr/bin/env python
"""Counterpart of the reference's scripts/train_ar_policy.py on the batched MI355X engine.

    python scripts/train_ar_policy.py --num_envs 4096 --iters 3
    python scripts/train_ar_policy.py --cfg kin_poly --config_root /path/to/KinPoly [--iter 750] [--data train]      # the reference's command line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_ar_policy.py

With --cfg the run is configured by the reference's `config/statear/<cfg>.yml` (kinpoly_amd/config.py): optimisers, schedules, PPO and sampling
constants, reward weights, the horizon from `min_batch_size`, `results/all/statear/<cfg>/` for checkpoints (`models_policy/iter_%04d.p`, every
`save_model_interval` iterations, agent_ar.py:341-364), `freq_dict.pt`, `eval_dict_*.pt` and `log/log.txt`; `--iter N` resumes from that checkpoint
(train_ar_policy.py:92-104); every `save_model_interval` iterations the test sets are evaluated (`eval_policy("test")`, agent_ar.py:291-293).

The reference's MoCap dataset and trained UHC weights are not part of its repository (downlaod_data.sh), so this
driver builds synthetic takes in the reference's feature-file schema (all four action classes with their objects, SURVEY.md
section 8(d) config 4) unless --data points at a real feature file, and trains seeded random-init networks on them.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--horizon", type=int, default=99)
    ap.add_argument("--clip_len", type=int, default=100)
    ap.add_argument("--num_optim_epoch", type=int, default=10)
    ap.add_argument("--num_step_update", type=int, default=20)
    ap.add_argument("--pool_depth", type=int, default=4, help="clips kept queued behind every env's current one (one host read per pool_depth steps)")
    ap.add_argument("--cache_init_context", action="store_true", help="look init_qpos / init_qvel of a window up once it has been computed under the same context-network parameters")
    ap.add_argument("--save", type=str, default="")
    ap.add_argument("--synthetic_amp", type=float, default=0.3, help="amplitude bound (rad) of the synthetic takes' joint sinusoids")
    ap.add_argument("--result_dir", type=str, default="", help="without --cfg: where freq_dict.pt / eval_dict_*.pt go (with --cfg: results/all/statear/<cfg>/results)")
    ap.add_argument("--data", type=str, default="", help="feature file in the reference's schema (<data_dir>/features/<data_file>.p)")
    ap.add_argument("--cfg", type=str, default=None, help="config id (config/**/<cfg>.yml under --config_root) or a .yml path, as the reference's --cfg")
    ap.add_argument("--config_root", type=str, default=None, help="directory that holds config/ and the dataset_path of the yml (default: cwd)")
    ap.add_argument("--iter", type=int, default=0, help="resume from models_policy/iter_%%04d.p (the reference's --iter)")
    ap.add_argument("--warm_start", action="store_true", help="AgentAR.train_init before the first iteration of a fresh run: supervised warm start of the kinematic policy "
                    "(policy_specs.warm_update_init / warm_update_full epochs, default 500 / 50; the reference always runs it at --iter 0)")
    ap.add_argument("--warm_update_init", type=int, default=None); ap.add_argument("--warm_update_full", type=int, default=None)
    ap.add_argument("--num_sample", type=int, default=None); ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--cc_ckpt", type=str, default="", help="trained UHC checkpoint in the reference's layout (scripts/train_uhc.py --save); with --cfg the default is "
                    "results/motion_im/<cc_cfg>/models/iter_<cc_iter>.p as in the reference")
    ap.add_argument("--wild", action="store_true")
    ap.add_argument("--update_dtype", choices=("fp32", "fp64"), default="fp32", help="fp64: the reference's training precision (train_ar_policy.py:76-77) on fp64 master copies of the "
                    "policy / value nets (GRUCell loop + torch FK instead of the fused HIP re-unroll); the roll-out stays fp32")
    ap.add_argument("--no_reference_bugs", action="store_true", help="corrected forms instead of the reference's behaviour: gradient clip at every PPO step (the reference's "
                    "generator-consumed clip acts on a run's first step only), LambdaLR continued on --iter N (the reference restarts its decay), min of the workers' min_episode_reward")
    ap.add_argument("--rl_update", type=int, default=1, help="without --cfg: policy_specs.rl_update (PPO epochs)"); ap.add_argument("--step_update", type=int, default=1, help="without --cfg: policy_specs.step_update")
    ap.add_argument("--eval_first_last", action="store_true", help="play every training take whole with mean actions before the first and after the last iteration (no freq_dict "
                    "feedback) and print mean percent / coverage / joint-angle error: a like-for-like measure of what the updates did to the policy")
    ap.add_argument("--eval_every", type=int, default=0, help="with --eval_first_last: also after every N-th iteration")
    ap.add_argument("--load", type=str, default="", help="start from this checkpoint (reference layout) instead of seeded random init; schedules and optimiser state start fresh")
    ap.add_argument("--min_horizon", type=int, default=0, help="with --cfg: lower bound of the per-env horizon derived from min_batch_size (0 = fr_num / 4; ADVICE r4: 10000 / 4096 envs "
                    "would be 3-step fragments that hang on the V bootstrap)")
    ap.add_argument("--min_batch_size", type=int, default=None, help="samples per update (the reference's policy_specs.min_batch_size; default: the cfg's, 0 without --cfg): a sample() call's "
                    "batch is cut into whole-env slices of about this many samples and each slice is one reference iteration (schedules, PPO epochs, supervised steps, epoch += 1); 0 = one update per call")
    ap.add_argument("--no_log", action="store_true")
    ap.add_argument("--test_data", type=str, nargs="*", default=[], help="feature files of the test sets evaluated every save_model_interval iterations")
    ap.add_argument("--test_data_wild", type=str, nargs="*", default=[], help="the same for --wild test sets (evaluated on the ..._mesh_all.xml engine, agent_ar.py:305-314, 464)")
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from kinpoly_amd.agent import AgentAR
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    from kinpoly_amd import dataset as D
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.model_compiler import read_kpm
    # the agent's kinematic twin sim doubles as the FK engine of the feature construction
    fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), args.num_envs, local)
    cfg = None
    if args.cfg:
        from kinpoly_amd.config import Config
        if args.config_root:
            os.chdir(args.config_root)                 # the yml's dataset_path and the results/ tree are relative to it, as in the reference
        cfg = Config(args.cfg, wild=args.wild, create_dirs=(rank == 0))
        args.clip_len = int(cfg.fr_num)
        if not args.data and os.path.exists(cfg.feature_path()):
            args.data = cfg.feature_path()
    if args.data:
        ds = D.StateARDataset(args.data, takes=(cfg.takes["train"] or None) if cfg else None, fr_num=args.clip_len, wild=args.wild, seed=4 + rank, device=fk_sim.device)
    else:       # the reference's MoCap features are not in its repository: same schema, synthetic takes.  ONE
        # data set for the whole job (take seed independent of the rank): the job-wide freq_dict is keyed by take name, so a name must
        # mean the same motion on every rank; only the draw stream (dataset seed) differs per rank
        takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=4, T_range=(args.clip_len + 10, args.clip_len + 60),
                                  body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=4, amp_max=args.synthetic_amp)
        ds = D.StateARDataset(takes, fr_num=args.clip_len, seed=4 + rank, device=fk_sim.device)
    if rank == 0:
        print(f"dataset: {ds.get_len()} takes, {len(ds.freq_indices)} windows of {args.clip_len} frames", flush=True)

    # every episode draws its clip through data_loader.sample_seq(freq_dict, sampling_temp, sampling_freq) (agent_ar.py:519-523): the
    # agent keeps the freq_dict and feeds each finished episode's [percent, fr_start] back (random window starts, adaptive takes)
    cc_ckpt = args.cc_ckpt or (cfg.cc_checkpoint_path() if cfg is not None else None) or None
    if rank == 0 and cc_ckpt:
        print(f"loading model from checkpoint: {cc_ckpt}", flush=True)
    upd_kw = dict(update_dtype=torch.float64 if args.update_dtype == "fp64" else None, reference_bugs=not args.no_reference_bugs)
    if rank == 0 and not args.no_reference_bugs:          # nothing at run time says so otherwise
        print("note: reference_bugs = True (results identical to the reference's): the policy gradient is clipped on the run's FIRST optimiser step only, a resumed run "
              "restarts its LambdaLR decay, LoggerRL.merge takes the max of the workers' min rewards; --no_reference_bugs selects the corrected forms", flush=True)
    if cfg is None:
        agent = AgentAR(args.num_envs, dataset=ds, device=local, horizon=args.horizon, **upd_kw, num_optim_epoch=args.num_optim_epoch, cc_checkpoint=cc_ckpt, result_dir=args.result_dir or None,
                        num_step_update=args.num_step_update, sampling_temp=0.3, sampling_freq=0.5, pool_depth=args.pool_depth, cache_init_context=args.cache_init_context,
                        rl_update=bool(args.rl_update), step_update=bool(args.step_update), min_batch_size=args.min_batch_size or 0)
        first, last, interval = 0, args.iters, 0
    else:
        # the reference collects min_batch_size steps of WHOLE episodes (~ fr_num frames each); a lock-step sampler with thousands of envs would meet
        # that count with a handful of steps per env, so the horizon gets a floor (fr_num / 4 unless --min_horizon says otherwise)
        horizon = cfg.horizon(args.num_envs, world, floor=args.min_horizon or max(1, int(cfg.fr_num) // 4))
        if rank == 0:
            print(f"horizon {horizon} steps per env = {horizon * args.num_envs * world} samples per iteration (min_batch_size {cfg.policy_specs.get('min_batch_size', 10000)})", flush=True)
        agent = AgentAR(args.num_envs, dataset=ds, device=local, horizon=horizon, pool_depth=args.pool_depth, **upd_kw,
                        cache_init_context=args.cache_init_context, result_dir=cfg.result_dir, cc_checkpoint=cc_ckpt, **cfg.agent_kwargs(),
                        min_batch_size=int(cfg.policy_specs.get("min_batch_size", 10000)) if args.min_batch_size is None else args.min_batch_size)
        cfg.apply_reward_weights(agent.env)
        agent.test_datasets = ([D.StateARDataset(p, data_mode="test", fr_num=args.clip_len, wild=args.wild, seed=4, device=fk_sim.device) for p in args.test_data]
                               + [D.StateARDataset(p, data_mode="test", fr_num=args.clip_len, wild=True, seed=4, device=fk_sim.device) for p in args.test_data_wild])
        if args.iter > 0:                              # AgentAR(checkpoint_epoch=args.iter) -> load_checkpoint (agent_ar.py:72-73, 318-339)
            agent.load_checkpoint(cfg.checkpoint_path(args.iter))
            agent.epoch = args.iter
            # the reference builds FRESH LambdaLR schedulers on resume (setup_optimizer runs before load_checkpoint and nothing restores them,
            # agent_ar.py:60-73, 215-225): a resumed run restarts its learning-rate decay at epoch 0.  Reproduced; --no_reference_bugs continues it
            for _ in range(args.iter if args.no_reference_bugs else 0):
                agent.trainer.per_epoch_update(); agent.sched_sup.step()
        first, last, interval = args.iter, (args.iter + args.iters if args.iters else int(cfg.num_epoch)), int(cfg.policy_specs.get("save_model_interval", cfg.save_model_interval))
    if args.load:
        agent.load_checkpoint(args.load)
    log_file = open(os.path.join(cfg.log_dir, "log.txt"), "a") if (cfg is not None and rank == 0 and not args.no_log) else None
    if args.warm_start and first == 0:             # train_init (agent_ar.py:366-385), then save_checkpoint(0) -> iter_0001.p
        ps = cfg.policy_specs if cfg is not None else {}
        y = cfg.yaml_data if cfg is not None else {}
        pick = lambda a, d: d if a is None else a      # noqa: E731
        ws = agent.train_init(pick(args.warm_update_init, int(ps.get("warm_update_init", 500))), pick(args.warm_update_full, int(ps.get("warm_update_full", 50))),
                              pick(args.num_sample, int(y.get("num_sample", 20000))), pick(args.batch_size, int(y.get("batch_size", 128))),
                              noise_std=float(y.get("noise_std", 0.0)) if y.get("add_noise", False) else 0.0)
        if rank == 0:
            print(json.dumps({"warm_start": ws}), flush=True)
            if cfg is not None:
                agent.save_checkpoint(cfg.checkpoint_path(1))
    def fixed_eval(tag):
        from kinpoly_amd.evaluate import eval_dataset
        env_e, builder = agent._eval_engine(None)
        res = eval_dataset(env_e, agent.policy_net, builder, ds)
        pc = np.array([r["percent"] for r in res.values()])
        err = np.mean([np.abs(np.asarray(r["pred"])[:, 7:] - np.asarray(r["target"])[:, 7:]).mean() for r in res.values()])
        if rank == 0:
            print(json.dumps({"fixed_eval": tag, "takes": len(pc), "mean_percent": float(pc.mean()), "coverage": int((pc == 1).sum()), "mean_abs_joint_err": float(err)}), flush=True)
    if args.eval_first_last:
        fixed_eval("before")
    # one optimize_policy call = one sample() of num_envs x horizon steps = as many reference iterations as it has update slices (min_batch_size): `it` counts
    # reference iterations (agent.epoch), which is what the schedules, the checkpoint names and --iters mean
    agent.epoch = first
    while agent.epoch < last:
        it0 = agent.epoch
        if args.eval_first_last and args.eval_every and it0 > first and (it0 - first) % args.eval_every < getattr(agent, "_last_slices", 1):
            fixed_eval(f"iter{it0}")
        info = agent.optimize_policy(it0)
        it = agent.epoch - 1
        agent._last_slices = agent.epoch - it0
        if interval and (it + 1) // interval > it0 // interval:      # optimize_policy's periodic test-set evaluation, then train_ar_policy.py:95-97
            info["log_eval"] = agent.eval_policy("test")
            if rank == 0:
                agent.save_checkpoint(cfg.checkpoint_path(it + 1))
        if rank == 0:
            log = info.pop("log")
            line = agent.log_train({**info, "log": log}, cfg_id=cfg.id if cfg else "synthetic", max_iter_num=int(cfg.policy_specs.get("max_iter_num", last)) if cfg else last)
            if log_file is not None:
                log_file.write(line + "\n"); log_file.flush()
            print(json.dumps({"iter": it, **{k: (float("%.6g" % v) if isinstance(v, float) else v) for k, v in info.items()}, "log": log.as_dict()}), flush=True)
    if args.eval_first_last:
        fixed_eval("after")
    if args.save and rank == 0:
        agent.save_checkpoint(args.save)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
