"""Discrete-event model of kp_step_queue_kernel's schedule on measured per-env costs: what would other queue disciplines buy?
Collects the per-env cycles of two consecutive control-step launches (kp_sim_launch_cost) on the standing + contact scene and on the bench's
tracked workload, splits an env's cost over its jobs in proportion to their substeps (+ a fixed hand-over cost), and list-schedules
the jobs on 2048 waves under several disciplines.   gpurun python tools/micro/queue_policy_sim.py"""
import heapq, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HAND = float(__import__('os').environ.get('KP_HAND', 28e3))          # cycles per hand-over (state in / out + job set-up; 49e3 before round 3's torque hand-over), DESIGN.md section 6
CLK = 2.38e3         # cycles per microsecond


def simulate(cost, sizes, slots, policy, prev=None, theta=1.25):
    """cost [n] cycles of the whole control step per env (hand-overs excluded); returns makespan in cycles"""
    n = len(cost)
    frac = np.asarray(sizes, float) / sum(sizes)
    jobs = cost[:, None] * frac[None, :] + HAND                       # [n, parts]
    order = np.arange(n)
    if policy in ("lpt_true", "prio_lpt_true"):
        order = np.argsort(-cost, kind="stable")
    if policy in ("lpt_prev", "prio_lpt_prev"):
        order = np.argsort(-prev, kind="stable")
    fifo = [(e, 0) for e in order]          # FIFO of (env, part); appended as jobs get published
    head = 0
    prio = []                               # priority FIFO
    phead = 0
    avg = jobs.mean(axis=0)
    free = [(0.0, w) for w in range(slots)]
    heapq.heapify(free)
    running = []                            # (end_time, env, part)
    t_end = 0.0
    # event-driven: waves become free at times; published jobs become available at their predecessor's end
    pending = []                            # heap of (avail_time, seq, env, part, to_prio)
    seq = 0
    done = 0
    total = n * len(sizes)
    while done < total:
        t, w = heapq.heappop(free)
        # make every job published up to t visible, in publication order
        while pending and pending[0][0] <= t:
            at, _, e, p, pr = heapq.heappop(pending)
            (prio if pr else fifo).append((e, p))
        if policy == "prio_rem" and (head < len(fifo)):
            # best visible job by remaining work (jobs of envs that have run at least one job: measured rate x substeps left; first jobs: FIFO order, lowest priority)
            best, bi = None, -1
            for i in range(head, len(fifo)):
                e_, p_ = fifo[i]
                if e_ < 0: continue
                key = (cost[e_] * (1.0 - sum(frac[:p_]))) if p_ > 0 else -1.0 - i * 1e-9
                if best is None or key > best: best, bi = key, i
            e, p = fifo[bi]; fifo[bi] = (-1, 0)
            while head < len(fifo) and fifo[head][0] < 0: head += 1
        elif phead < len(prio):
            e, p = prio[phead]; phead += 1
        elif head < len(fifo):
            e, p = fifo[head]; head += 1
        else:
            # nothing visible: wait for the next publication
            at = pending[0][0]
            heapq.heappush(free, (at, w))
            continue
        d = jobs[e, p]
        end = t + d
        t_end = max(t_end, end)
        done += 1
        if p + 1 < len(sizes):
            to_prio = False
            if policy.startswith("prio"):
                to_prio = d > theta * avg[p]
            if policy == "continue_first":
                to_prio = True
            heapq.heappush(pending, (end, seq, e, p + 1, to_prio)); seq += 1
        heapq.heappush(free, (end, w))
    return t_end


def collect(kind):
    import torch
    from kinpoly_amd.sim import KpModel, KpSim
    if kind == "standing":
        std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
        n = 4096
        rng = np.random.default_rng(3)
        qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.2
        qvel = rng.normal(size=(n, 75)) * 0.5
        sim = KpSim(KpModel(), n)
        q = torch.tensor(qpos, dtype=torch.float32, device="cuda"); v = torch.tensor(qvel, dtype=torch.float32, device="cuda")
        sim.set_state(q, v); sim.set_target(q.clone())
        a = torch.zeros((n, 75), dtype=torch.float32, device="cuda")
        out = []
        for it in range(8):
            sim.step_ctrl(a, 15)
            out.append((sim.launch_cost().astype(np.float64), sim.last_step_seconds() * 1e3))
        return out[-2], out[-1]
    import bench
    env, policy, sampler, std = bench.build_engine(0, 4, 64, "tracked")
    a_track = bench.tracking_action(env)
    sampler.start()
    bench.rollout_steps(sampler, 12, a_track, False)
    c0 = (env.sim.launch_cost().astype(np.float64), env.sim.last_step_seconds() * 1e3)
    bench.rollout_steps(sampler, 1, a_track, False)
    c1 = (env.sim.launch_cost().astype(np.float64), env.sim.last_step_seconds() * 1e3)
    return c0, c1


if __name__ == "__main__":
    for kind in ("standing", "tracked"):
        (prev, ms0), (cost, ms1) = collect(kind)
        cost = cost - 3 * HAND          # launch_cost includes the jobs' hand-overs
        prev = prev - 3 * HAND
        print(f"== {kind}: measured launch {ms1:.3f} ms; env cycles mean {cost.mean() / 1e6:.2f} M p50 {np.percentile(cost, 50) / 1e6:.2f} p90 {np.percentile(cost, 90) / 1e6:.2f} "
              f"p99 {np.percentile(cost, 99) / 1e6:.2f} max {cost.max() / 1e6:.2f} M = {cost.max() / CLK / 1e3:.3f} ms; sum / 2048 = {(cost + 3 * HAND).sum() / 2048 / CLK / 1e3:.3f} ms; "
              f"corr with the previous launch {np.corrcoef(cost, prev)[0, 1]:.2f}")
        for theta in (1.1, 1.25, 1.4, 1.6, 2.0):
            print(f"   jobs (7, 5, 3): priority lane for jobs that ran > {theta} x the mean: makespan {simulate(cost, (7, 5, 3), 2048, 'prio', prev, theta) / CLK / 1e3:.3f} ms")
        for sizes in ((6, 5, 4), (7, 5, 3), (5, 4, 3, 2, 1), (15,)):
            for pol in ("fifo", "lpt_prev", "lpt_true", "prio", "prio_rem", "prio_lpt_prev", "continue_first"):
                if len(sizes) == 1 and pol not in ("fifo", "lpt_prev", "lpt_true"):
                    continue
                mk = simulate(cost, sizes, 2048, pol, prev)
                print(f"   jobs {sizes}: {pol:15s} makespan {mk / CLK / 1e3:.3f} ms")
