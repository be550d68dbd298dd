"""Build an instrumented copy of the simulator library that counts, per env and control step, HOW the active set changes between Newton
iterations (is a rank-k update of the factorisation worth building?).  KP_PROFILE slots:
    [0] re-factorisations after a substep's first (= iterations whose active set differs from the previous iterate's)
    [1] ... of which exactly 1 row flipped   [2] exactly 2   [3] 3 or 4   [4] 5 or more          (pyramid rows + joint-limit rows)
    [5] ... of which only rows of object-side contacts flipped (vertex entity is an object: H_hh and H_ho keep their values)   [objects kernel]
    [6] rows flipped in all   [7] cycles of the control step (as in the product)

    python tools/micro/flip_instr.py tools/micro/bin/libkinpoly_sim_flips.so ; then tools/micro/flip_profile.py on the GPU box"""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.build import OPT_FLAGS  # noqa: E402


def patch(s):
    def rep(a, b, count=1):
        nonlocal s
        assert s.count(a) >= count, (s.count(a), a[:90])
        s = s.replace(a, b, count)
    # per-contact flips, by the side the contact's vertex entity is on
    rep('''__device__ __forceinline__ float active_set_changed(EnvLds& s, const Params& P, int tid, float& deep) {''',
        '''__device__ __forceinline__ float active_set_changed(EnvLds& s, const Params& P, int tid, float& deep, float& fh, float& fo) {''')
    rep('''        if (m != s.con_act[c]) changed = 1.f;''',
        '''        if (m != s.con_act[c]) { changed = 1.f; (s.con_body[c] < D_NB ? fh : fo) += (float)__popc(m ^ (unsigned)s.con_act[c]); }''')
    rep('''            if (ex != s.extra[i]) changed = 1.f;''', '''            if (ex != s.extra[i]) { changed = 1.f; kp_fh += 1.f; }''', 2)
    rep('''        changed += active_set_changed<NT>(s, P, tid, deep);''', '''        changed += active_set_changed<NT>(s, P, tid, deep, kp_fh, kp_fo);''', 2)
    rep('''    float changed = 0.f, deep = 0.f;
    auto active_set = [&]() {
        changed = 0.f; deep = 0.f;''', '''    float changed = 0.f, deep = 0.f, kp_fh = 0.f, kp_fo = 0.f;
    auto active_set = [&]() {
        changed = 0.f; deep = 0.f; kp_fh = 0.f; kp_fo = 0.f;''', 2)
    record = '''        { const float fh_ = wave_sum(kp_fh), fo_ = wave_sum(kp_fo); const int k_ = (int)(fh_ + fo_ + 0.5f);
          if (changed > 0.f) { np[0]++; np[k_ <= 1 ? 1 : k_ == 2 ? 2 : k_ <= 4 ? 3 : 4]++; if (fh_ == 0.f) np[5]++; np[6] += (unsigned long long)k_; } }
'''
    rep('''        active_set();
        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }     // gradient(new) = (1 - alpha) gradient(old) on an unchanged active set''',
        '''        active_set();
''' + record + '''        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }''')
    rep('''        active_set();
        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }     // gradient(new) = (1 - alpha) gradient(old): see solve_constraints_direct''',
        '''        active_set();
''' + record + '''        if (changed == 0.f && P.scale * fabsf(1.0f - alpha) * sqrtf(g2) < P.tol) { it++; done = true; break; }''')
    rep('''__device__ __forceinline__ int solve_constraints_direct(EnvLds& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap) {''',
        '''__device__ __forceinline__ int solve_constraints_direct(EnvLds& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap, unsigned long long* np) {''')
    rep('''__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap) {''',
        '''__device__ __forceinline__ int solve_constraints_obj(EnvLdsObj& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap, unsigned long long* np) {''')
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    rep("niter_total += solve_constraints_direct<NT>(s, P, L8, depth, tid, nfact_total, ncap_total);", "niter_total += solve_constraints_direct<NT>(s, P, L8, depth, tid, nfact_total, ncap_total, pc);")
    rep("niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total);", "niter_total += solve_constraints_obj<NT>(s, P, L8, depth, tid, nfact_total, ncap_total, pc);")
    return s


def patch_cost(s):
    """mode `cost`: what a rank-k update would compete with.  Slots of the floor solver: [0] cycles / [1] calls of a substep's first factorisation + solve,
    [2] / [3] of the later re-factorisations (dirty levels only) + solve, [4] / [5] of aba_resolve (solve through the standing factors)."""
    def rep(a, b):
        nonlocal s
        assert a in s, a[:90]
        s = s.replace(a, b, 1)
    rep('''            aba_solve<NT, false>(s, P, L8, s.x, s.search, true, tid, it == 0 ? D_NLEV : lev_hist, nullptr, conlev);
            nfact++;
        }
        else aba_resolve(s, L8, s.x, nullptr, s.search);''', '''            const unsigned long long c0_ = __builtin_readcyclecounter();
            aba_solve<NT, false>(s, P, L8, s.x, s.search, true, tid, it == 0 ? D_NLEV : lev_hist, nullptr, conlev);
            np[it == 0 ? 0 : 2] += __builtin_readcyclecounter() - c0_; np[it == 0 ? 1 : 3]++;
            nfact++;
        }
        else { const unsigned long long c0_ = __builtin_readcyclecounter(); aba_resolve(s, L8, s.x, nullptr, s.search); np[4] += __builtin_readcyclecounter() - c0_; np[5]++; }''')
    rep('''__device__ __forceinline__ int solve_constraints_direct(EnvLds& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap) {''',
        '''__device__ __forceinline__ int solve_constraints_direct(EnvLds& s, const Params& P, const Lane8& L8, int depth, int tid, int& nfact, int& ncap, unsigned long long* np) {''')
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    rep("niter_total += solve_constraints_direct<NT>(s, P, L8, depth, tid, nfact_total, ncap_total);", "niter_total += solve_constraints_direct<NT>(s, P, L8, depth, tid, nfact_total, ncap_total, pc);")
    return s


def main(out, mode="flips"):
    tmp = tempfile.mkdtemp(prefix="kp_flip_instr_")
    src = os.path.join(tmp, "kinpoly_amd", "csrc")
    shutil.copytree(os.path.join(ROOT, "kinpoly_amd", "csrc"), src)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(src, "kp_step_kernel.hpp")
    text = (patch_cost if mode == "cost" else patch)(open(p).read())
    open(p, "w").write(text)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           os.path.join(src, "kp_sim.hip"), "-o", out])
    shutil.rmtree(tmp)
    print("built", out)


if __name__ == "__main__":
    main(*sys.argv[1:])
