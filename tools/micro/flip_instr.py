"""Instrumented copy of the library: how many constraint rows change state between two Newton iterations of solve_constraints_obj() (the question a
rank-k update of the coupled factorisation turns on)?  KP_PROFILE slots per env and control step = [0] active-set tests after a line search, of which
[1] found no flip, [2] one flipped row, [3] two, [4] three or four, [5] five or more; [6] flipped rows in total; [7] control-step cycles.
    python tools/micro/flip_instr.py tools/micro/bin/libkinpoly_sim_flips.so
    KP_SIM_LIBRARY=tools/micro/bin/libkinpoly_sim_flips.so KP_PROFILE=1 KP_SUBSTEPS_PER_JOB=15 python tools/micro/objects_tail.py 3 flips"""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.build import OPT_FLAGS  # noqa: E402


def patch(s):
    def rep(a, b, n=1):
        nonlocal s
        assert s.count(a) == n, (s.count(a), a[:90])
        s = s.replace(a, b)
    rep("__device__ __forceinline__ float active_set_changed(EnvLds& s, const Params& P, int tid, float& deep) {",
        "__device__ __forceinline__ float active_set_changed(EnvLds& s, const Params& P, int tid, float& deep, float* nflip = nullptr) {")
    rep("        if (m != s.con_act[c]) changed = 1.f;\n", "        if (m != s.con_act[c]) changed = 1.f;\n        if (nflip) *nflip += (float)__popc(m ^ (unsigned)s.con_act[c]);\n")
    rep("__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap) {\n",
        "__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap, unsigned long long* np) {\n    float nfl_ = 0.f;\n")
    # the object solver's own lambda: count limit flips and contact-row flips
    rep("            if (ex != s.extra[i]) changed = 1.f;\n            s.extra[i] = ex;\n            if (ex != 0.f) deep = fmaxf(deep, (float)s.bdep[s.dbody[i]]);     // active joint limit: its body's level is dirty\n        }\n        changed += active_set_changed<NT>(s, P, tid, deep);\n        changed = (NT == 64) ? (__ballot(changed > 0.f) != 0ull ? 1.f : 0.f) : block_sum<NT>(s, changed, tid);     // one wave: a ballot is the whole reduction\n        KP_SYNC();\n    };\n    active_set();\n    for (; it < P.max_iter; it++) {\n        // gradient: humanoid dofs (mres - J^T f) and object wrenches\n",
        "            if (ex != s.extra[i]) { changed = 1.f; nfl_ += 1.f; }\n            s.extra[i] = ex;\n            if (ex != 0.f) deep = fmaxf(deep, (float)s.bdep[s.dbody[i]]);\n        }\n        changed += active_set_changed<NT>(s, P, tid, deep, &nfl_);\n        changed = (NT == 64) ? (__ballot(changed > 0.f) != 0ull ? 1.f : 0.f) : block_sum<NT>(s, changed, tid);\n        nfl_ = wave_sum(nfl_);\n        KP_SYNC();\n    };\n    active_set();\n    for (; it < P.max_iter; it++) {\n")
    rep("        if (improvement < P.tol) { it++; done = true; break; }\n        active_set();\n        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }     // gradient(new) = (1 - alpha) gradient(old): see solve_constraints_direct\n    }\n    if (!done) ncap++;          // the solver stopped at opt.iterations: counted per env in diag (flags >> 8)\n",
        "        if (improvement < P.tol) { it++; done = true; break; }\n        nfl_ = 0.f;\n        active_set();\n        { const int k_ = (int)(nfl_ + 0.5f); np[0]++; np[k_ == 0 ? 1 : k_ == 1 ? 2 : k_ == 2 ? 3 : k_ <= 4 ? 4 : 5]++; np[6] += (unsigned long long)k_; }\n"
        "        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }\n    }\n    if (!done) ncap++;\n")
    rep("            niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total);", "            niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total, pc);")
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    return s


def main(out):
    tmp = tempfile.mkdtemp(prefix="kp_flip_instr_")
    src = os.path.join(tmp, "kinpoly_amd", "csrc")
    shutil.copytree(os.path.join(ROOT, "kinpoly_amd", "csrc"), src)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(src, "kp_step_kernel.hpp")
    text = patch(open(p).read())
    open(p, "w").write(text)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", os.path.join(src, "kp_sim.hip"), "-o", out])
    shutil.rmtree(tmp)
    print("built", out)


if __name__ == "__main__":
    main(*sys.argv[1:])
