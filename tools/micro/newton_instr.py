"""Build an instrumented copy of the simulator library whose KP_PROFILE slots hold the sub-phases of the Newton contact solve instead of the
substep phases (init | gradient | factor+solve | rows+quad forms | line search | update+cost | #line-search evaluations | total).

    python tools/micro/newton_instr.py /tmp/libkinpoly_sim_newton.so        # then KP_LIB=... python tools/micro/newton_profile.py
The patched sources live in a temp dir; the tree is not touched."""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def patch(s):
    def rep(a, b):
        nonlocal s
        assert a in s, a[:70]
        s = s.replace(a, b, 1)
    rep('''__device__ __forceinline__ int solve_constraints(EnvLds& s, const Params& P, const Lane8& L8, int tid, int& nfact, int& ncap) {
    if (s.ncon == 0 && s.nlim == 0) {''', '''__device__ __forceinline__ int solve_constraints(EnvLds& s, const Params& P, const Lane8& L8, int tid, int& nfact, int& ncap, unsigned long long* np) {
    unsigned long long t0_ = __builtin_readcyclecounter();
#define NP(i) { unsigned long long t1_ = __builtin_readcyclecounter(); np[i] += t1_ - t0_; t0_ = t1_; }
    if (s.ncon == 0 && s.nlim == 0) {''')
    rep('''    for (; it < P.max_iter; it++) {
        // gradient = M (qacc - qacc_s)''', '''    NP(0)
    for (; it < P.max_iter; it++) {
        // gradient = M (qacc - qacc_s)''')
    rep('''        KP_SYNC();
        if (P.scale * sqrtf(g2) < P.tol) { done = true; break; }''', '''        KP_SYNC();
        NP(1)
        if (P.scale * sqrtf(g2) < P.tol) { done = true; break; }''')
    rep('''        else aba_resolve(s, L8, s.x, nullptr, s.search);
        eval_rows<NT, OBJ>''', '''        else aba_resolve(s, L8, s.x, nullptr, s.search);
        NP(2)
        eval_rows<NT, OBJ>''')
    rep("        const float alpha = line_search<NT>(s, P, g0, h0, tid, rowcost);\n", "        NP(3)\n        const float alpha = line_search<NT>(s, P, g0, h0, tid, rowcost);\n        NP(4)\n")
    rep('''        cost = newcost;
        if (improvement < P.tol) { it++; done = true; break; }''', '''        cost = newcost;
        NP(5)
        if (improvement < P.tol) { it++; done = true; break; }''')
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    rep("else niter_total += solve_constraints<NT, OBJ>(s, P, L8, tid, nfact_total, ncap_total);",
        "else niter_total += solve_constraints<NT, OBJ>(s, P, L8, tid, nfact_total, ncap_total, pc);")
    return s


def patch_gradient(s):
    """slots: stage-1 body wrenches | subtree sums | projection | g2 / x / active set / sums | first_clean_level | rest of the solve"""
    def rep(a, b):
        nonlocal s
        assert a in s, a[:70]
        s = s.replace(a, b, 1)
    rep("const float* vb, float* out, bool with_inertia, bool with_forces, int tid) {\n    if (tid < D_NB) {",
        "const float* vb, float* out, bool with_inertia, bool with_forces, int tid, unsigned long long* np = nullptr) {\n"
        "    unsigned long long t0_ = __builtin_readcyclecounter();\n"
        "#define NPW(i) if (np) { unsigned long long t1_ = __builtin_readcyclecounter(); np[i] += t1_ - t0_; t0_ = t1_; }\n"
        "    if (tid < D_NB) {")
    rep("    KP_SYNC();\n    subtree_sums<NT>(s, tid);\n", "    KP_SYNC();\n    NPW(0)\n    subtree_sums<NT>(s, tid);\n    NPW(1)\n")
    rep("        out[d] = v;\n    }\n    KP_SYNC();\n}", "        out[d] = v;\n    }\n    KP_SYNC();\n    NPW(2)\n}")
    rep("const Lane8& L8, int tid, int& nfact, int& ncap) {\n    if (s.ncon == 0 && s.nlim == 0) {",
        "const Lane8& L8, int tid, int& nfact, int& ncap, unsigned long long* np) {\n"
        "    unsigned long long t0_ = __builtin_readcyclecounter();\n"
        "#define NP(i) { unsigned long long t1_ = __builtin_readcyclecounter(); np[i] += t1_ - t0_; t0_ = t1_; }\n"
        "    if (s.ncon == 0 && s.nlim == 0) {")
    rep("wrench_project<NT, OBJ>(s, P, sacc, s.qacc, s.qacc_s, s.grad(), true, true, tid);",
        "NP(5) wrench_project<NT, OBJ>(s, P, sacc, s.qacc, s.qacc_s, s.grad(), true, true, tid, np); t0_ = __builtin_readcyclecounter();")
    rep("        KP_SYNC();\n        if (P.scale * sqrtf(g2) < P.tol) { done = true; break; }",
        "        KP_SYNC();\n        NP(3)\n        if (P.scale * sqrtf(g2) < P.tol) { done = true; break; }")
    rep("lev_hist = max(lev_hist, first_clean_level<NT>(s, deep, tid));", "lev_hist = max(lev_hist, first_clean_level<NT>(s, deep, tid)); NP(4)")
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    rep("else niter_total += solve_constraints<NT, OBJ>(s, P, L8, tid, nfact_total, ncap_total);",
        "else niter_total += solve_constraints<NT, OBJ>(s, P, L8, tid, nfact_total, ncap_total, pc);")
    return s


def patch_factor(s):
    """slots (Newton factorisations only): level set-up + child loads | contact inertia | 3-dof elimination + stores | clean-level half | forward pass | rest"""
    def rep(a, b):
        nonlocal s
        assert a in s, a[:70]
        s = s.replace(a, b, 1)
    rep("const float* warm = nullptr, float* dacc = nullptr, unsigned conlev = 0xFFFFFFFFu) {\n",
        "const float* warm = nullptr, float* dacc = nullptr, unsigned conlev = 0xFFFFFFFFu, unsigned long long* np = nullptr) {\n"
        "    unsigned long long t0_ = __builtin_readcyclecounter();\n"
        "#define NPA(i) if (np) { unsigned long long t1_ = __builtin_readcyclecounter(); np[i] += t1_ - t0_; t0_ = t1_; }\n")
    rep("            if (active && rowok) s.pAa[6 * b + r] = pA;\n            KP_SYNC();\n            continue;",
        "            if (active && rowok) s.pAa[6 * b + r] = pA;\n            KP_SYNC();\n            NPA(3)\n            continue;")
    rep("        if (contact_inertia && active && ((conlev >> lev) & 1u)) {", "        NPA(0)\n        if (contact_inertia && active && ((conlev >> lev) & 1u)) {")
    rep("        if (lev == 0) {\n            aba_elim3(s, L, rhs, 3, active, IAx, pA);", "        NPA(1)\n        if (lev == 0) {\n            aba_elim3(s, L, rhs, 3, active, IAx, pA);")
    rep("            s.pAa[6 * b + r] = pA;\n        }\n        KP_SYNC();\n    }\n    aba_forward<WARM>(s, L, out, D_NLEV - 1, warm, dacc);\n}",
        "            s.pAa[6 * b + r] = pA;\n        }\n        KP_SYNC();\n        NPA(2)\n    }\n    aba_forward<WARM>(s, L, out, D_NLEV - 1, warm, dacc);\n    NPA(4)\n}")
    rep("const Lane8& L8, int tid, int& nfact, int& ncap) {\n    if (s.ncon == 0 && s.nlim == 0) {",
        "const Lane8& L8, int tid, int& nfact, int& ncap, unsigned long long* np) {\n"
        "    unsigned long long t0_ = __builtin_readcyclecounter();\n"
        "#define NP(i) { unsigned long long t1_ = __builtin_readcyclecounter(); np[i] += t1_ - t0_; t0_ = t1_; }\n"
        "    if (s.ncon == 0 && s.nlim == 0) {")
    rep("            aba_solve<NT, OBJ>(s, P, L8, s.x, s.search, true, tid, lev_hist, nullptr, nullptr, nullptr, conlev);",
        "            NP(5) aba_solve<NT, OBJ>(s, P, L8, s.x, s.search, true, tid, lev_hist, nullptr, nullptr, nullptr, conlev, np); t0_ = __builtin_readcyclecounter();")
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    rep("else niter_total += solve_constraints<NT, OBJ>(s, P, L8, tid, nfact_total, ncap_total);",
        "else niter_total += solve_constraints<NT, OBJ>(s, P, L8, tid, nfact_total, ncap_total, pc);")
    return s


def main(out, which="phases"):
    tmp = tempfile.mkdtemp(prefix="kp_newton_instr_")
    src = os.path.join(tmp, "kinpoly_amd", "csrc")
    shutil.copytree(os.path.join(ROOT, "kinpoly_amd", "csrc"), src)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(src, "kp_step_kernel.hpp")
    text = {"gradient": patch_gradient, "factor": patch_factor, "phases": patch}[which](open(p).read())
    open(p, "w").write(text)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           os.path.join(src, "kp_sim.hip"), "-o", out])
    shutil.rmtree(tmp)
    print("built", out)


if __name__ == "__main__":
    main(*sys.argv[1:])
