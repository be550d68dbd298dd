"""Does a third wave on EVERY SIMD pay?  (DESIGN "occupancy".)

The product kernel is capped at 8 envs per CU = 2 waves per SIMD by LDS (18 128 B per env), not by registers.  Round 5 compared 9 vs 8 envs per CU
(one SIMD in four with a third wave): under-powered.  This script builds the free-fall instantiation of the SAME kernel source
(-DKP_LEAN_FREEFALL=1: BASELINE configs[1], no contact, no joint limits, no Newton solve; the arrays that instantiation never reads share one
union: 12 736 B per env = 10 LDS granules) for 3 waves per SIMD (-DKP_WAVES_PER_SIMD=3: <= 168 VGPRs; it needs 152, no scratch) and runs ONE binary at

    8 envs per CU   (LDS padded to 20 480 B, 2048 queue slots)      2 waves on every SIMD
    10 envs per CU  (LDS padded to 15 360 B, 2560 queue slots)      2.5
    12 envs per CU  (natural size,           3072 queue slots)      3 waves on every SIMD

on 4096 and 6144 envs (stable-PD torque + residual force + forward pass + one articulated-body solve per substep, 15 substeps), 60 timed control
steps per measurement, three interleaved rounds.  The product library on the same workload (contact = 0, limits = 0) is the cross-check that the
lean build is the same arithmetic (max |dqpos| after 4 control steps) and costs the same at 8 envs per CU.

    python tools/micro/occupancy_lean.py build                      (here: hipcc cross-compiles)
    gpurun python tools/micro/occupancy_lean.py run                 (A/B table -> stdout)
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- python tools/micro/occupancy_lean.py one <pad> <slots> <n>     (one variant, for the counters)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_lean3.so")
VARIANTS = [("lean", 20480, 2048), ("lean", 15360, 2560), ("lean", 0, 3072), ("product", 0, 2048)]


def build():
    from kinpoly_amd.build import OPT_FLAGS
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-array-bounds",
           "-DKP_LEAN_FREEFALL=1", "-DKP_WAVES_PER_SIMD=3", os.path.join(ROOT, "kinpoly_amd", "csrc", "kp_sim.hip"), "-o", LIB]
    subprocess.check_call(cmd)
    print("built", LIB)


def one(kind, pad, slots, n, steps=60, warm=5):
    """one measurement in this process: mean launch ms over `steps` control steps, and the state after 4 control steps from the common start"""
    import numpy as np
    import torch
    from kinpoly_amd import sim as kpsim
    if kind == "lean":
        kpsim.load_library(LIB)
        if pad:
            os.environ["KP_LDS_PAD"] = str(pad)
    from kinpoly_amd.sim import KpModel, KpSim
    std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
    rng = np.random.default_rng(7)
    qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 2] += 10.0; qpos[:, 7:] += np.clip(rng.normal(size=(n, 69)) * 0.2, -np.pi, np.pi)
    qvel = rng.normal(size=(n, 75)) * 0.5
    tgt = np.tile(std["qpos"], (n, 1)); tgt[:, 7:] += rng.normal(size=(n, 69)) * 0.1
    act = rng.normal(size=(n, 75)) * 0.3
    dev = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    sim = KpSim(KpModel(contact=0, limits=0, queue_slots=slots), n)
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(tgt))
    a = dev(act)
    for _ in range(4):
        sim.step_ctrl(a, 15)
    q4 = sim.get("qpos").double().cpu().numpy()
    for _ in range(warm):
        sim.step_ctrl(a, 15)
    sim.timing_reset()
    for _ in range(steps):
        sim.step_ctrl(a, 15)
    sec, k = sim.timing_mean_seconds()
    bad = int(sim.diag()[:, 2].max())
    return {"kind": kind, "pad": pad, "slots": slots, "envs_per_cu": slots // 256, "n": n, "launch_ms": sec * 1e3, "launches": k, "bad": bad,
            "lds_bytes_per_env": int(KpModel().get_option("lds_bytes_per_env")), "q4": q4}


def run():
    import numpy as np
    rows, ref = [], {}
    for n in (4096, 6144):
        for rnd in range(3):
            for kind, pad, slots in VARIANTS:
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "one", kind, str(pad), str(slots), str(n), "--json"], capture_output=True, text=True, timeout=600)
                line = [l for l in out.stdout.splitlines() if l.startswith("{")]
                if not line:
                    print("FAILED", kind, pad, slots, n, out.stderr[-2000:], flush=True)
                    continue
                r = json.loads(line[-1])
                q4 = np.array(r.pop("q4"))
                key = (n,)
                if kind == "product":
                    ref[key] = q4
                r["round"] = rnd
                r["dq_vs_product"] = float(np.abs(q4 - ref[key]).max()) if key in ref else None
                rows.append(r)
                print(json.dumps(r), flush=True)
    print("\n| envs | build | envs per CU | launch ms (3 rounds) | mean | vs 8 per CU |")
    print("|---|---|---|---|---|---|")
    for n in (4096, 6144):
        base = None
        for kind, pad, slots in VARIANTS:
            ms = [r["launch_ms"] for r in rows if r["n"] == n and r["kind"] == kind and r["pad"] == pad and r["slots"] == slots]
            if not ms:
                continue
            m = sum(ms) / len(ms)
            if kind == "lean" and slots == 2048:
                base = m
            rel = f"{(m / base - 1) * 100:+.1f} %" if base else ""
            print(f"| {n} | {kind} | {slots // 256} | {' / '.join(f'{x:.3f}' for x in ms)} | {m:.3f} | {rel} |")


if __name__ == "__main__":
    if sys.argv[1:2] == ["build"]:
        build()
    elif sys.argv[1:2] == ["one"]:
        kind, pad, slots, n = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        r = one(kind, pad, slots, n)
        if "--json" in sys.argv:
            r["q4"] = r["q4"][:64].tolist()          # the first 64 envs are enough for the cross-check
            print(json.dumps(r))
        else:
            r.pop("q4"); print(r)
    else:
        run()
