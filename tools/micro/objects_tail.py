"""Which envs does the `objects` launch end on?  bench.py's objects workload; after the warm-up, for a few control-step launches: per-env cycles
(kp_sim_launch_cost), Newton iterations / factorisations / contacts (kp_sim_diag), the env's action class -- the distribution per class and
the 24 costliest envs of every sampled launch, and whether they are the same envs from launch to launch.
    python tools/micro/objects_tail.py [launches]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 6
env, policy, sampler, std = bench.build_engine(0, 0, 64, "objects")
bench.stagger_episodes(env, sampler, 0, True)
bench.rollout_steps(sampler, 20, None, False, True)
names = ["sit", "push", "avoid", "step"]
take = env.ctx["take_ind"].cpu().numpy().astype(int) if "take_ind" in env.ctx else None
prev_top = None
for it in range(n_launch):
    bench.rollout_steps(sampler, 1, None, False, True)
    torch.cuda.synchronize()
    ms = env.sim.last_step_seconds() * 1e3
    c = env.sim.launch_cost().astype(np.float64)
    dg = env.sim.diag().astype(np.int64)
    row = env.row.cpu().numpy().astype(int)
    cls = (take[row] // 8) if take is not None else np.zeros(env.n, int)
    ncon, nit, ncap, maxcon, nfact = dg[:, 0], dg[:, 1], dg[:, 2] >> 8, dg[:, 3] & 255, dg[:, 3] >> 8
    clock = c.max() / (ms * 1e-3) if it == 0 else clock          # the longest env's cycles over the launch's time: a lower bound of the shader clock
    print(f"launch {it}: {ms:.3f} ms; env cycles p50 {np.percentile(c, 50):.0f} p90 {np.percentile(c, 90):.0f} p99 {np.percentile(c, 99):.0f} max {c.max():.0f}; "
          f"sum / 1792 slots {c.sum() / 1792:.0f} cycles = {c.sum() / 1792 / c.max() * ms:.3f} ms at the longest env's rate")
    for k, nm in enumerate(names):
        m = cls == k
        if m.any():
            print(f"   {nm:5s} {int(m.sum()):5d} envs: cycles mean {c[m].mean():9.0f} p99 {np.percentile(c[m], 99):9.0f} max {c[m].max():9.0f}; newton it / substep {nit[m].mean() / 15:.2f} "
                  f"(max {nit[m].max() / 15:.2f}); contacts mean {ncon[m].mean():.1f} max {maxcon[m].max()}; fact / substep {nfact[m].mean() / 15:.2f}; cycles per newton it (slope) "
                  f"{np.polyfit(nit[m], c[m], 1)[0]:.0f}")
    top = np.argsort(-c)[:24]
    print("   costliest envs: " + "; ".join(f"{e}:{names[cls[e]]} {c[e] / 1e6:.2f}M it {nit[e]} f {nfact[e]} con {maxcon[e]} cap {ncap[e]}" for e in top[:24]))
    if prev_top is not None:
        print(f"   of these 24, {len(set(top) & set(prev_top))} were among the previous launch's 24 costliest")
    prev_top = top
    if os.environ.get("KP_PROFILE") == "1":        # per-phase cycles of the LAST job of the control step (run with KP_SUBSTEPS_PER_JOB=15: the whole step is one job)
        pc = env.sim.phase_cycles_env()
        ph = ("spd", "kin_bias", "collide", "constraint", "smooth", "contact", "integrate", "total")
        if len(sys.argv) > 2 and sys.argv[2] == "solve":          # the library built by tools/micro/solve_instr.py
            ph = ("gradient", "factorisation", "schur_columns", "object_system+back_subst", "rows+quad_forms", "line_search", "other_in_solve", "total")
        if len(sys.argv) > 2 and sys.argv[2] == "collide":        # tools/micro/collide_instr.py, variant A
            ph = ("mpr_queries", "mpr_hits", "mpr_cycles", "object_object+object_floor_cycles", "floor_hull_cycles", "box_box_calls", "box_box_cycles", "total")
        order_ = np.argsort(-pc[:, 7])
        if len(sys.argv) > 2 and sys.argv[2] == "flips":          # tools/micro/flip_instr.py: rows that change state between two Newton iterations
            for nm, sel in (("24 costliest", order_[:24]), ("next 200", order_[24:224]), ("median 400", order_[1848:2248]), ("all", order_)):
                t = pc[sel, 0].sum()
                print(f"   active-set tests after a line search, {nm}: {pc[sel, 0].mean() / 15:.2f} per substep; no flip {pc[sel, 1].sum() / t:.1%}, one row {pc[sel, 2].sum() / t:.1%}, "
                      f"two {pc[sel, 3].sum() / t:.1%}, three or four {pc[sel, 4].sum() / t:.1%}, five or more {pc[sel, 5].sum() / t:.1%}; rows per test {pc[sel, 6].sum() / t:.2f}")
            continue
        for nm, sel in (("24 costliest", order_[:24]), ("next 200", order_[24:224]), ("median 400", order_[1848:2248])):
            print(f"   phases / substep, {nm}: " + ", ".join(f"{k} {pc[sel, j].mean() / 15:.0f}" for j, k in enumerate(ph)) + f"; newton it / substep {nit[sel].mean() / 15:.2f}, contacts {maxcon[sel].mean():.1f}")
    print(f"   corr(cycles, newton iterations) {np.corrcoef(c, nit)[0, 1]:.3f}; corr(cycles, max contacts) {np.corrcoef(c, maxcon)[0, 1]:.3f}", flush=True)
