"""Does a third wave per SIMD pay?  Premise test for the LDS diet: a build of the simulator with D_MAXCON = 8 (17.7 KB of LDS per env:
9 envs per CU fit) and 168 VGPRs, run on the standing + contact scene with (a) LDS padded back to 20 480 B and 8 x 256 queue slots,
(b) its natural LDS size and 9 x 256 slots.  The physics differs from the product build (contacts are capped at 8); only the ratio
(a) / (b) means anything.     python tools/micro/occupancy_premise.py build && gpurun python tools/micro/occupancy_premise.py"""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.build import OPT_FLAGS  # noqa: E402  (the product build's optimisation flags)
LIB = os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_premise.so")


def build():
    tmp = tempfile.mkdtemp(prefix="kp_premise_")
    src = os.path.join(tmp, "kinpoly_amd", "csrc")
    shutil.copytree(os.path.join(ROOT, "kinpoly_amd", "csrc"), src)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))

    def edit(name, a, b):
        p = os.path.join(src, name)
        t = open(p).read()
        assert a in t, a
        open(p, "w").write(t.replace(a, b, 1))
    edit("kp_device.hpp", "constexpr int D_MAXCON = 64;", "constexpr int D_MAXCON = 8;")
    edit("kp_step_kernel.hpp", "__global__ __launch_bounds__(64, 2) void kp_step_queue_kernel", "__global__ __launch_bounds__(64, 3) void kp_step_queue_kernel")
    edit("kp_sim.hip", "    const int spj = s->model->substeps_per_job;",
         "    if (const char* e = std::getenv(\"KP_LDS_PAD\")) lds = std::max<size_t>(lds, (size_t)std::atoi(e));\n    const int spj = s->model->substeps_per_job;")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           os.path.join(src, "kp_sim.hip"), "-o", LIB])
    shutil.rmtree(tmp)
    print("built", LIB)


def run():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from kinpoly_amd import sim as kpsim
    kpsim.load_library(LIB)
    from kinpoly_amd.sim import KpModel, KpSim
    std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
    n = 4608        # 18 x 256: whole rounds for both 8 and 9 envs per CU
    rng = np.random.default_rng(3)
    qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.2
    qvel = rng.normal(size=(n, 75)) * 0.5
    for pad, slots in ((20480, 2048), (0, 2304), (20480, 2048), (0, 2304)):
        os.environ["KP_LDS_PAD"] = str(pad)
        sim = KpSim(KpModel(queue_slots=slots), n)
        q = torch.tensor(qpos, dtype=torch.float32, device="cuda"); v = torch.tensor(qvel, dtype=torch.float32, device="cuda")
        sim.set_state(q, v); sim.set_target(q.clone())
        a = torch.zeros((n, 75), dtype=torch.float32, device="cuda")
        for _ in range(3):
            sim.step_ctrl(a, 15)
        sim.timing_reset()
        for _ in range(20):
            sim.step_ctrl(a, 15)
        ms, k = sim.timing_mean_seconds()
        print(f"LDS pad {pad:6d} B, {slots} slots ({slots // 256} envs per CU): {ms * 1e3:.3f} ms / launch of {n} envs", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
