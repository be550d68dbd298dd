"""Finer instrumentation of solve_constraints_obj than obj_instr.py (two builds, 7 slots each + total), for the coupled heavy envs that bound the
`objects` launch.   python tools/micro/obj_instr2.py A tools/micro/bin/libkinpoly_sim_objfineA.so ; ... B ...objfineB.so ; then obj_heavy2.py
  A: con_prepare | wrench_project | obj_gradient + |g| | obj_hessian | H_hh factorisation (aba_solve) | coupling rhs + Schur columns | everything else
  B: up to the dense solve | dense solve | hull coupling wrench + back-substitution pass | row evaluation | quadratic forms | line search | update + cost + active set"""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.build import OPT_FLAGS  # noqa: E402

HEAD = ('''__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap) {
''', '''__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap, unsigned long long* np) {
    unsigned long long t0_ = __builtin_readcyclecounter();
#define NP(i) { unsigned long long t1_ = __builtin_readcyclecounter(); np[i] += t1_ - t0_; t0_ = t1_; }
''')
# (anchor text, slot charged with the time since the previous mark, mark goes BEFORE the anchor)
MARKS = {
    "A": [("        con_prepare<NT>(s, P, tid);                  // lane = contact", 6),
          ("        wrench_project<NT, true>(s, P, sacc, s.qacc, nullptr, grad, true, true, tid, s.jv3, s.fb, s.applied);", 0),
          ("        if (nobj > 0) obj_gradient(s, tid);", 1),
          ("        const bool refactor = it == 0 || changed > 0.f;", 2),
          ("            if (no6 > 0) obj_hessian(s, P, s.ogr, -1.0f, tid);", 6),
          ("            lev_hist = max(lev_hist, first_clean_level<NT>(s, deep, tid));", 3),
          ("            if (couple) { obj_coupling_u(s, P, s.ot, tid); if (tid < no6) s.Sm[ST * tid + no6] += s.ot[tid]; KP_SYNC(); }     // rhs_o -= H_oh y0", 4),
          ("        if (!refactor) {                                                  // factors reused", 5)],
    "B": [("        if (no6 > 0) {\n            dense_solve(s, no6, tid, refactor);", 0),
          ("            if (tid < no6) s.osrch[tid] = s.Sm[ST * tid + no6];\n            KP_SYNC();\n            if (couple) {", 1),
          ("        if (tid < no6) s.sv[6 * D_NB + tid] = s.osrch[tid];\n        KP_SYNC();\n        eval_rows<NT, true>(s, s.search, s.jv3, s.lim_jv, false, tid);", 2),
          ("        if (tid < nobj) sts6(s.oMv + 6 * tid, inert_mul(s.oIe + 10 * tid, lds6(s.osrch + 6 * tid)));", 3),
          ("        float rownew, rc0 = 0.f;\n        const float alpha = line_search<NT>(s, P, g0, h0, tid, rownew, it == 0, rc0);\n        if (it == 0) rowcost = rc0;\n        if (!(alpha > 0.f)) { done = true; break; }\n        for (int i = tid; i < D_NV; i += NT) s.qacc[i] += alpha * s.search[i];\n        for (int i = tid; i < D_NB * 6; i += NT) sacc[i] += alpha * s.sv[i];\n        if (tid < no6)", 4),
          ("        if (it == 0) rowcost = rc0;\n        if (!(alpha > 0.f)) { done = true; break; }\n        for (int i = tid; i < D_NV; i += NT) s.qacc[i] += alpha * s.search[i];\n        for (int i = tid; i < D_NB * 6; i += NT) sacc[i] += alpha * s.sv[i];\n        if (tid < no6)", 5),
          ("        con_prepare<NT>(s, P, tid);                  // lane = contact", 6)],
}


def patch(s, which):
    def rep(a, b):
        nonlocal s
        assert s.count(a) == 1, (s.count(a), a[:90])
        s = s.replace(a, b, 1)
    rep(*HEAD)
    for anchor, slot in MARKS[which]:
        rep(anchor, f"        NP({slot})\n" + anchor)
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    rep("niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total);",
        "niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total, pc);")
    return s


def main(which, out):
    tmp = tempfile.mkdtemp(prefix="kp_obj_instr2_")
    src = os.path.join(tmp, "kinpoly_amd", "csrc")
    shutil.copytree(os.path.join(ROOT, "kinpoly_amd", "csrc"), src)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(src, "kp_step_kernel.hpp")
    text = patch(open(p).read(), which)
    open(p, "w").write(text)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           os.path.join(src, "kp_sim.hip"), "-o", out])
    shutil.rmtree(tmp)
    print("built", out)


if __name__ == "__main__":
    main(*sys.argv[1:])
