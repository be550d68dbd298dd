"""Instrumented copy of the simulator library for the OBJECT kernel: the KP_PROFILE slots hold the sub-phases of solve_constraints_obj
(init + candidates | gradient (wrench_project + con_prepare + obj_gradient + active set) | H_hh factorisation (+ obj_hessian) |
Schur-complement column passes | dense solve + back-substitution pass | rows + quad forms + line search | update + cost | total).

    python tools/micro/obj_instr.py tools/micro/bin/libkinpoly_sim_objnewton.so     # then python tools/micro/obj_profile.py
The patched sources live in a temp dir; the tree is not touched."""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.build import OPT_FLAGS  # noqa: E402  (the product build's optimisation flags)


def patch(s):
    def rep(a, b, cnt=1):
        nonlocal s
        assert a in s, a[:80]
        s = s.replace(a, b, cnt)
    rep('''__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap) {
''', '''__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap, unsigned long long* np) {
    unsigned long long t0_ = __builtin_readcyclecounter();
#define NP(i) { unsigned long long t1_ = __builtin_readcyclecounter(); np[i] += t1_ - t0_; t0_ = t1_; }
''')
    rep('''    for (; it < P.max_iter; it++) {
        // gradient: humanoid dofs (mres - J^T f) and object wrenches''', '''    NP(0)
    for (; it < P.max_iter; it++) {
        // gradient: humanoid dofs (mres - J^T f) and object wrenches''')
    rep('''        const bool refactor = it == 0 || changed > 0.f;
        nfact += refactor;''', '''        NP(1)
        const bool refactor = it == 0 || changed > 0.f;
        nfact += refactor;''')
    rep('''        if (refactor && couple) schur_columns(s, P, cmask, tid);''', '''        NP(2)
        if (refactor && couple) schur_columns(s, P, cmask, tid);
        NP(3)''')
    rep('''        if (tid < no6) s.sv[6 * D_NB + tid] = s.osrch[tid];
        KP_SYNC();
        eval_rows<NT, true>(s, s.search, s.jv3, s.lim_jv, false, tid);''', '''        NP(4)
        if (tid < no6) s.sv[6 * D_NB + tid] = s.osrch[tid];
        KP_SYNC();
        eval_rows<NT, true>(s, s.search, s.jv3, s.lim_jv, false, tid);''')
    rep('''        const float alpha = line_search<NT>(s, P, g0, h0, tid, rownew, it == 0, rc0);
        if (it == 0) rowcost = rc0;
        if (!(alpha > 0.f)) { done = true; break; }
        for (int i = tid; i < D_NV; i += NT) s.qacc[i] += alpha * s.search[i];
        for (int i = tid; i < D_NB * 6; i += NT) sacc[i] += alpha * s.sv[i];
        if (tid < no6)''', '''        const float alpha = line_search<NT>(s, P, g0, h0, tid, rownew, it == 0, rc0);
        NP(5)
        if (it == 0) rowcost = rc0;
        if (!(alpha > 0.f)) { done = true; break; }
        for (int i = tid; i < D_NV; i += NT) s.qacc[i] += alpha * s.search[i];
        for (int i = tid; i < D_NB * 6; i += NT) sacc[i] += alpha * s.sv[i];
        if (tid < no6)''')
    rep('''        rowcost = rownew;
        if (improvement < P.tol) { it++; done = true; break; }
        active_set();
        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }     // gradient(new) = (1 - alpha) gradient(old): see solve_constraints_direct''', '''        rowcost = rownew;
        NP(6)
        if (improvement < P.tol) { it++; done = true; break; }
        active_set();
        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }''')
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    rep("niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total);",
        "niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total, pc);")
    return s


def main(out):
    tmp = tempfile.mkdtemp(prefix="kp_obj_instr_")
    src = os.path.join(tmp, "kinpoly_amd", "csrc")
    shutil.copytree(os.path.join(ROOT, "kinpoly_amd", "csrc"), src)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(src, "kp_step_kernel.hpp")
    text = patch(open(p).read())
    open(p, "w").write(text)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           os.path.join(src, "kp_sim.hip"), "-o", out])
    shutil.rmtree(tmp)
    print("built", out)


if __name__ == "__main__":
    main(*sys.argv[1:])
