"""Build an instrumented copy of the simulator library whose KP_PROFILE slots hold the sub-phases of the Newton contact solve instead of the
substep phases (init | gradient | factor+solve | rows+quad forms | line search | update+cost | - | total) of solve_constraints_direct.

    python tools/micro/newton_instr.py /tmp/libkinpoly_sim_newton.so        # then KP_LIB=... python tools/micro/newton_profile.py
The patched sources live in a temp dir; the tree is not touched."""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.build import OPT_FLAGS  # noqa: E402  (the product build's optimisation flags)


def patch(s):
    def rep(a, b):
        nonlocal s
        assert a in s, a[:70]
        s = s.replace(a, b, 1)
    rep('''__device__ __forceinline__ int solve_constraints_direct(EnvLds& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap) {
    if (s.ncon == 0 && s.nlim == 0) {''', '''__device__ __forceinline__ int solve_constraints_direct(EnvLds& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap, unsigned long long* np) {
    unsigned long long t0_ = __builtin_readcyclecounter();
#define NP(i) { unsigned long long t1_ = __builtin_readcyclecounter(); np[i] += t1_ - t0_; t0_ = t1_; }
    if (s.ncon == 0 && s.nlim == 0) {''')
    rep('''    for (; it < P.max_iter; it++) {
        wrench_project<NT, false>(s, P, sacc, s.qacc, nullptr, grad, true, true, tid, nullptr, s.fb, s.applied);''', '''    NP(0)
    for (; it < P.max_iter; it++) {
        wrench_project<NT, false>(s, P, sacc, s.qacc, nullptr, grad, true, true, tid, nullptr, s.fb, s.applied);''')
    rep('''        g2 = block_sum<NT>(s, g2, tid);
        KP_SYNC();
        if (P.scale * sqrtf(g2) < P.tol) { done = true; break; }
        if (it == 0 || changed > 0.f) {
            const int clean''', '''        g2 = block_sum<NT>(s, g2, tid);
        KP_SYNC();
        NP(1)
        if (P.scale * sqrtf(g2) < P.tol) { done = true; break; }
        if (it == 0 || changed > 0.f) {
            const int clean''')
    rep('''        else aba_resolve(s, L8, s.x, nullptr, s.search);
        eval_rows<NT, false>(s, s.search, s.jv3, s.lim_jv, false, tid);           // aref''', '''        else aba_resolve(s, L8, s.x, nullptr, s.search);
        NP(2)
        eval_rows<NT, false>(s, s.search, s.jv3, s.lim_jv, false, tid);           // aref''')
    rep("        const float alpha = line_search<NT>(s, P, g0, h0, tid, rownew, it == 0, rc0);\n        if (it == 0) rowcost = rc0;\n        if (!(alpha > 0.f)) { done = true; break; }\n        for (int i = tid; i < D_NV; i += NT) s.qacc[i] += alpha * s.search[i];\n        for (int i = tid; i < D_NB * 6; i += NT) sacc[i] += alpha * s.sv[i];\n        for (int k = tid",
        "        NP(3)\n        const float alpha = line_search<NT>(s, P, g0, h0, tid, rownew, it == 0, rc0);\n        NP(4)\n        if (it == 0) rowcost = rc0;\n        if (!(alpha > 0.f)) { done = true; break; }\n        for (int i = tid; i < D_NV; i += NT) s.qacc[i] += alpha * s.search[i];\n        for (int i = tid; i < D_NB * 6; i += NT) sacc[i] += alpha * s.sv[i];\n        for (int k = tid")
    rep('''        rowcost = rownew;
        if (improvement < P.tol) { it++; done = true; break; }
        // The Newton step of a model that was exact''', '''        rowcost = rownew;
        NP(5)
        if (improvement < P.tol) { it++; done = true; break; }
        // The Newton step of a model that was exact''')
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    rep("niter_total += solve_constraints_direct<NT>(s, P, L8, depth, tid, nfact_total, ncap_total);",
        "niter_total += solve_constraints_direct<NT>(s, P, L8, depth, tid, nfact_total, ncap_total, pc);")
    return s


def main(out, which="phases"):
    tmp = tempfile.mkdtemp(prefix="kp_newton_instr_")
    src = os.path.join(tmp, "kinpoly_amd", "csrc")
    shutil.copytree(os.path.join(ROOT, "kinpoly_amd", "csrc"), src)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(src, "kp_step_kernel.hpp")
    assert which == "phases", "the round-2 sub-modes (gradient / factor) patched the two-candidate solver; see the history of this file"
    text = patch(open(p).read())
    open(p, "w").write(text)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           os.path.join(src, "kp_sim.hip"), "-o", out])
    shutil.rmtree(tmp)
    print("built", out)


if __name__ == "__main__":
    main(*sys.argv[1:])
