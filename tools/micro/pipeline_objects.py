"""Does splitting the GPU's 4096 envs into S sub-batches on S HIP streams, each running the whole env-step (policies, physics launch, reset), hide the tail of
the control-step launch?  The objects workload's launch ends on one env's serial chain (1.1 ms after the work ran out); the floor workload's tail is 0.3 ms.
    python tools/micro/pipeline_objects.py [workload] [S,S,...]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "objects"
Ss = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4").split(",")]
N = int(os.environ.get("KP_PIPE_N", bench.ENVS_PER_GPU))      # KP_PIPE_N=3072: three sub-batches of 1024 instead of 1365
for S in Ss:
    streams = [torch.cuda.Stream() for _ in range(S)]
    parts = []
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            n = N // S
            env, policy, sampler, std = bench.build_engine(0, 4 + i, 64, wl, n_envs=n)
            a_track = None
            if wl == "tracked":
                a_track = bench.tracking_action(env); sampler.start()
            bench.stagger_episodes(env, sampler, 4 + i, wl == "objects")
            parts.append((env, sampler, a_track))
    torch.cuda.synchronize()

    def run(k):
        for _ in range(k):
            for st, (env, sampler, a_track) in zip(streams, parts):
                with torch.cuda.stream(st):
                    bench.rollout_steps(sampler, 1, a_track, False, wl == "objects")
    run(15)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(40)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{wl} S={S}: {dt / 40 * 1e3:.3f} ms per {N // S * S}-env step -> {N // S * S * 40 / dt:.0f} env-steps/s (host enqueue {t_enq / 40 * 1e3:.2f} ms per step)", flush=True)
    del parts
    torch.cuda.empty_cache()
