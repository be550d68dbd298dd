"""Instrumented copy of the library for solve_constraints_obj(): KP_PROFILE slots (cycles per env and control step) = [0] gradient (con_prepare, wrench_project,
obj_gradient, its norm), [1] factorisation (obj_hessian, aba_solve, obj_coupling_u), [2] schur_columns, [3] dense object system + back-substitution pass
(or the re-solve through standing factors), [4] eval_rows of the search direction + the quadratic forms, [5] line search, [6] everything else inside the solve
(smooth object accelerations, first row evaluation, iterate update, active-set test), [7] control-step total.
    python tools/micro/solve_instr.py tools/micro/bin/libkinpoly_sim_solve.so
    KP_SIM_LIBRARY=tools/micro/bin/libkinpoly_sim_solve.so KP_PROFILE=1 KP_SUBSTEPS_PER_JOB=15 python tools/micro/objects_tail.py 3 solve"""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.build import OPT_FLAGS  # noqa: E402

TICK = "{ const unsigned long long t_ = __builtin_readcyclecounter(); np[%d] += t_ - tp_; tp_ = t_; }\n"


def patch(s):
    def rep(a, b):
        nonlocal s
        assert s.count(a) == 1, (s.count(a), a[:90])
        s = s.replace(a, b, 1)
    rep("__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap) {\n",
        "__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap, unsigned long long* np) {\n"
        "    unsigned long long tp_ = __builtin_readcyclecounter();\n")
    rep("    active_set();\n    for (; it < P.max_iter; it++) {\n        // gradient: humanoid dofs (mres - J^T f) and object wrenches\n", "    active_set();\n    " + TICK % 6 + "    for (; it < P.max_iter; it++) {\n")
    rep("        if (P.scale * sqrtf(g2) < P.tol) { done = true; break; }\n        // search direction\n", "        " + TICK % 0 + "        if (P.scale * sqrtf(g2) < P.tol) { done = true; break; }\n        // search direction\n")
    rep("        if (refactor && couple) schur_columns(s, P, cmask, tid);", "        " + TICK % 1 + "        if (refactor && couple) schur_columns(s, P, cmask, tid);\n        " + TICK % 2)
    rep("        if (tid < no6) s.sv[6 * D_NB + tid] = s.osrch[tid];\n        KP_SYNC();\n        eval_rows<NT, true>(s, s.search, s.jv3, s.lim_jv, false, tid);",
        "        " + TICK % 3 + "        if (tid < no6) s.sv[6 * D_NB + tid] = s.osrch[tid];\n        KP_SYNC();\n        eval_rows<NT, true>(s, s.search, s.jv3, s.lim_jv, false, tid);")
    rep("        float rownew, rc0 = 0.f;\n        const float alpha = line_search<NT>(s, P, g0, h0, tid, rownew, it == 0, rc0);\n        if (it == 0) rowcost = rc0;\n        if (!(alpha > 0.f)) { done = true; break; }\n        for (int i = tid; i < D_NV; i += NT) s.qacc[i] += alpha * s.search[i];\n        for (int i = tid; i < D_NB * 6; i += NT) sacc[i] += alpha * s.sv[i];\n        if (tid < no6) { s.oa[tid]",
        "        float rownew, rc0 = 0.f;\n        " + TICK % 4 + "        const float alpha = line_search<NT>(s, P, g0, h0, tid, rownew, it == 0, rc0);\n        " + TICK % 5 +
        "        if (it == 0) rowcost = rc0;\n        if (!(alpha > 0.f)) { done = true; break; }\n        for (int i = tid; i < D_NV; i += NT) s.qacc[i] += alpha * s.search[i];\n        for (int i = tid; i < D_NB * 6; i += NT) sacc[i] += alpha * s.sv[i];\n        if (tid < no6) { s.oa[tid]")
    rep("        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }     // gradient(new) = (1 - alpha) gradient(old): see solve_constraints_direct\n    }\n    if (!done) ncap++;          // the solver stopped at opt.iterations: counted per env in diag (flags >> 8)\n",
        "        " + TICK % 6 + "        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }\n    }\n    if (!done) ncap++;\n")
    rep("            niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total);", "            niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total, pc);")
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    return s


def main(out):
    tmp = tempfile.mkdtemp(prefix="kp_solve_instr_")
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
