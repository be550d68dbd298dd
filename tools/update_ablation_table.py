"""Table of tools/update_ablation.sh's runs: per variant the first / last iteration's reward, failure rate, surrogate and step loss, and the
iterations whose final-epoch surrogate was positive."""
import glob
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/update_ablation"
print("| update dtype | terms | iter 0 reward / fail | last reward / fail | mean reward last 10 | surr_loss first / last | iterations with surr_loss > 0 | step_loss first / last | log-ratio std after the 10 epochs, iterations 0 / 10 / 20 / last | action-network weight norm first / last | T_update s | fixed evaluation (all takes whole, mean actions): mean percent / coverage / joint error, before -> after |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(d, "fp*_*.log"))):
    rows = [json.loads(ln) for ln in open(f) if ln.startswith("{")]
    it = [r for r in rows if "iter" in r]
    if not it:
        continue
    name = os.path.basename(f)[:-4]
    dt, terms = name.split("_", 1)
    g = lambda r, k: ("%.4f" % r[k]) if k in r else "-"       # noqa: E731
    pos = [r["iter"] for r in it if r.get("surr_loss", -1) > 0]
    last10 = sum(r["avg_reward"] for r in it[-10:]) / len(it[-10:])
    ev = {r["fixed_eval"]: r for r in rows if "fixed_eval" in r}
    evs = (f"{ev['before']['mean_percent']:.3f} / {ev['before']['coverage']} of {ev['before']['takes']} / {ev['before']['mean_abs_joint_err']:.4f} -> "
           f"{ev['after']['mean_percent']:.3f} / {ev['after']['coverage']} / {ev['after']['mean_abs_joint_err']:.4f}") if "before" in ev and "after" in ev else "-"
    print(f"| {dt} | {terms} | {it[0]['avg_reward']:.3f} / {it[0]['fail_rate']:.4f} | {it[-1]['avg_reward']:.3f} / {it[-1]['fail_rate']:.4f} | {last10:.3f} | "
          f"{g(it[0], 'surr_loss')} / {g(it[-1], 'surr_loss')} | {len(pos)} of {len(it)}{(' (first: %d)' % pos[0]) if pos else ''} | {g(it[0], 'step_loss')} / {g(it[-1], 'step_loss')} | {' / '.join(('%.2f' % it[min(k, len(it) - 1)]['ppo_log_ratio_std']) if 'ppo_log_ratio_std' in it[0] else '-' for k in (0, 10, 20, len(it) - 1))} | {g(it[0], 'policy_param_norm')} / {g(it[-1], 'policy_param_norm')} | {it[-1]['T_update']:.2f} | {evs} |")
