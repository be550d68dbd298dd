"""kinpoly_amd: MI355X-native batched SMPL-humanoid simulator and rollout engine for the KinPoly hot path.

    sim       ctypes binding of libkinpoly_sim.so (the C ABI of include/kinpoly_sim.h); no CPU fallback
    env       BatchedHumanoidAREnv / HumanoidAREnv (kin_poly/envs/humanoid_ar_v1.py surface)
    uhc_env   BatchedHumanoidEnv, expert precompute, UHC reward (uhc/envs/humanoid_im.py surface)
    nets      PolicyMCP, KinPolicy, Value (checkpoint-compatible parameter names)
    context   TrajARNet / PolicyAR.init_context, batched over episodes
    rollout   VectorSampler, GAE + PPO, the advantage exchange step across ranks
    agent     AgentAR training iteration;  dataset  StateARDataset;  checkpoint  the reference's pickle layout
"""
