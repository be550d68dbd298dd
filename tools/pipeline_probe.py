"""Does splitting the 4096 envs of a GPU into S sub-batches on S HIP streams hide the tail of the control-step launch?
    python tools/pipeline_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context  # noqa: E402
from kinpoly_amd.nets import KinPolicy  # noqa: E402
from kinpoly_amd.rollout import VectorSampler  # noqa: E402

N, CLIP = 4096, 100
std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))


def one_step(sampler):
    env, pol = sampler.env, sampler.policy
    action, sampler.hx = pol.select_action(sampler.obs, sampler.hx, False, env.gen)
    _, _, done, info = env.step(action.contiguous())
    sampler.obs = env.reset(done).clone()
    sampler.hx = sampler.hx * (~done).float().unsqueeze(1)


for S in [int(x) for x in os.environ.get('KP_PIPE_S', '1,2,4').split(',')]:
    torch.manual_seed(4)
    policy = KinPolicy().cuda().float()
    streams = [torch.cuda.Stream() for _ in range(S)]
    parts = []
    cc = None
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            n = N // S
            opts = {k.lower()[3:]: int(v) for k, v in os.environ.items() if k in ("KP_QUEUE_SLOTS", "KP_LEAN_QUEUE", "KP_QUEUE_PRIO", "KP_QUEUE_LATE")}
            env = BatchedHumanoidAREnv(n, 0, mode="train", seed=4 + i, cc_policy=cc, model_options=opts)
            cc = env.cc_policy
            g = torch.Generator().manual_seed(4 + i)
            headings = (torch.rand(n, generator=g) * 2 - 1) * np.pi
            env.load_context(standing_context(n, CLIP, std["qpos"], std["qvel"], env.sim, headings))
            sm = VectorSampler(env, policy)
            sm.start()
            parts.append(sm)
    torch.cuda.synchronize()
    use_graph = os.environ.get("KP_PIPE_GRAPH", "0") == "1"
    graphs = []
    if use_graph:          # one hipGraph per sub-batch: the whole env-step (policies, C-ABI launches, reset) replayed with one host call
        with torch.no_grad():
            for st, sm in zip(streams, parts):
                with torch.cuda.stream(st):
                    for _ in range(3):
                        one_step(sm)
                    sm.static_obs, sm.static_hx = sm.obs.clone(), sm.hx.clone()
                st.synchronize()
                g = torch.cuda.CUDAGraph()
                g.register_generator_state(sm.env.gen)
                with torch.cuda.graph(g, stream=st):
                    sm.obs, sm.hx = sm.static_obs, sm.static_hx
                    one_step(sm)
                    sm.static_obs.copy_(sm.obs); sm.static_hx.copy_(sm.hx)
                graphs.append(g)
        torch.cuda.synchronize()

    def run(k):
        with torch.no_grad():
            for _ in range(k):
                for j, (st, sm) in enumerate(zip(streams, parts)):
                    with torch.cuda.stream(st):
                        if use_graph:
                            graphs[j].replay()
                        else:
                            one_step(sm)
    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(30)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"   host enqueue time {t_enq / 30 * 1e3:.3f} ms per 4096-env step")
    print(f"S={S} graph={int(use_graph)}: {dt / 30 * 1e3:.3f} ms per 4096-env step -> {N * 30 / dt:.0f} env-steps/s", flush=True)
