"""Where does a control-step launch of kp_step_queue_kernel spend its time?  Instrumented build (temp copy of the sources): every resident wave
records its life span in shader cycles (s_memtime) and in the constant 100 MHz clock (s_memrealtime) and the cycles it spent inside jobs.
Prints the real shader clock under this load, the mean busy fraction of a wave, and the launch span against the sum of the jobs.
    python tools/micro/queue_timeline.py build && gpurun python tools/micro/queue_timeline.py"""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.build import OPT_FLAGS  # noqa: E402  (the product build's optimisation flags)
LIB = os.path.join(ROOT, "tools", "micro", "bin", "libkinpoly_sim_timeline.so")


def build():
    tmp = tempfile.mkdtemp(prefix="kp_timeline_")
    src = os.path.join(tmp, "kinpoly_amd", "csrc")
    shutil.copytree(os.path.join(ROOT, "kinpoly_amd", "csrc"), src)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))

    def edit(name, a, b):
        p = os.path.join(src, name)
        t = open(p).read()
        assert a in t, a
        open(p, "w").write(t.replace(a, b, 1))
    edit("kp_sim.hip", "&& s->n <= 0xFFFFFF && !s->prof;", "&& s->n <= 0xFFFFFF;")
    edit("kp_sim.hip", "    for (int k = 0; k < 8; k++) { double acc = 0; for (int e = 0; e < s->n; e++) acc += (double)h[(size_t)e * 8 + k]; out[k] = acc / s->n; }",
         "    for (int k = 0; k < 8; k++) out[k] = (double)h[k];")
    edit("kp_step_kernel.hpp", "    const bool prof = A.prof != nullptr;", "    const bool prof = false;")
    edit("kp_step_kernel.hpp", "    const unsigned total = (unsigned)A.n_envs * (unsigned)A.n_parts;\n    for (;;) {",
         "    const unsigned total = (unsigned)A.n_envs * (unsigned)A.n_parts;\n"
         "    const unsigned long long life_m0 = __builtin_readcyclecounter(), life_r0 = __builtin_amdgcn_s_memrealtime();\n"
         "    unsigned long long busy = 0, last_m = life_m0, last_r = life_r0; unsigned njobs = 0;\n"
         "#define KP_LIFE_END() if (threadIdx.x == 0 && A.prof) { atomicAdd(&A.prof[0], last_m - life_m0); atomicAdd(&A.prof[1], last_r - life_r0); atomicAdd(&A.prof[2], busy); \\\n"
         "        atomicMin(&A.prof[3], life_r0); atomicMax(&A.prof[4], last_r); atomicAdd(&A.prof[5], (unsigned long long)njobs); atomicMax(&A.prof[6], life_r0); atomicAdd(&A.prof[7], 1ull); }\n"
         "    for (;;) {")
    edit("kp_step_kernel.hpp", "        if (idx >= total) return;\n        unsigned e, spins = 0;", "        if (idx >= total) { KP_LIFE_END() return; }\n        unsigned e, spins = 0;")
    edit("kp_step_kernel.hpp", "        step_body<64, OBJ, false, true>(A, env, part);\n",
         "        const unsigned long long jb = __builtin_readcyclecounter();\n        step_body<64, OBJ, false, true>(A, env, part);\n"
         "        last_m = __builtin_readcyclecounter(); last_r = __builtin_amdgcn_s_memrealtime(); busy += last_m - jb; njobs++;\n")
    edit("kp_step_kernel.hpp", "__global__ void k_queue_init(int n_envs, unsigned total, unsigned* __restrict__ jobq, unsigned* __restrict__ jobctr, const int* __restrict__ order) {",
         "__global__ void k_queue_init(int n_envs, unsigned total, unsigned* __restrict__ jobq, unsigned* __restrict__ jobctr, const int* __restrict__ order, unsigned long long* prof) {\n"
         "    if (prof && blockIdx.x == 0 && threadIdx.x < 8) prof[threadIdx.x] = threadIdx.x == 3 ? ~0ull : 0ull;")
    edit("kp_sim.hip", "s->stream, s->n, total, s->jobq, s->jobctr, A.order);", "s->stream, s->n, total, s->jobq, s->jobctr, A.order, s->prof);")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           os.path.join(src, "kp_sim.hip"), "-o", LIB])
    shutil.rmtree(tmp)
    print("built", LIB)


def run():
    os.environ["KP_PROFILE"] = "1"
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from kinpoly_amd import sim as kpsim
    kpsim.load_library(LIB)
    from kinpoly_amd.sim import KpModel, KpSim
    std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
    n = 4096
    rng = np.random.default_rng(3)
    qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.2
    qvel = rng.normal(size=(n, 75)) * 0.5
    sim = KpSim(KpModel(), n)
    q = torch.tensor(qpos, dtype=torch.float32, device="cuda"); v = torch.tensor(qvel, dtype=torch.float32, device="cuda")
    sim.set_state(q, v); sim.set_target(q.clone())
    a = torch.zeros((n, 75), dtype=torch.float32, device="cuda")
    for _ in range(4):
        sim.step_ctrl(a, 15)
    for rep in range(3):
        sim.step_ctrl(a, 15)
        ms = sim.last_step_seconds() * 1e3
        life_m, life_r, busy, r_min, r_max, njobs, r_last_start, nslots = list(sim.phase_cycles().values())
        clock = life_m / life_r * 100.0          # MHz: shader cycles per tick of the 100 MHz clock
        span_us = (r_max - r_min) / 100.0
        print(f"launch {ms:.3f} ms (events); first wave start -> last job end {span_us / 1e3:.3f} ms; last wave started {(r_last_start - r_min) / 100.0:.1f} us after the first; "
              f"{int(nslots)} waves, {njobs / nslots:.2f} jobs each; shader clock {clock:.0f} MHz; mean wave life {life_r / nslots / 100.0 / 1e3:.3f} ms, "
              f"busy inside jobs {busy / life_m * 100:.1f} %; sum of job cycles / waves = {busy / nslots / 1e6:.3f} M cycles = {busy / nslots / clock / 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
