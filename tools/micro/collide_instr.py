"""Instrumented copy of the library for collide(): KP_PROFILE slots = [0] hull-object MPR queries, [1] of which returned a contact, [2] cycles inside them,
[3] object-object / object-floor narrow phases: cycles, [4] floor-hull (mjc_PlaneConvex) cycles, [5] box-box calls, [6] cycles of box-box, [7] control-step total.
    python tools/micro/collide_instr.py tools/micro/bin/libkinpoly_sim_collide.so ; then tools/micro/collide_profile.py on the GPU box"""
import os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.build import OPT_FLAGS  # noqa: E402


def patch(s):
    def rep(a, b):
        nonlocal s
        assert s.count(a) == 1, (s.count(a), a[:90])
        s = s.replace(a, b, 1)
    rep("__device__ __forceinline__ void collide(EnvLds& s, const DevTables& T, const Params& P, int tid) {",
        "__device__ __forceinline__ void collide(EnvLds& s, const DevTables& T, const Params& P, int tid, unsigned long long* np) {")
    rep("                    Contact c;\n                    if (convex_pair(ga, hb, P.margin, c, mpr_scratch(s)) && ncon < D_MAXCON) {",
        "                    Contact c;\n                    const unsigned long long c0_ = __builtin_readcyclecounter();\n                    const int hit_ = convex_pair(ga, hb, P.margin, c, mpr_scratch(s));\n"
        "                    np[0]++; np[1] += hit_; np[2] += __builtin_readcyclecounter() - c0_;\n                    if (hit_ && ncon < D_MAXCON) {")
    rep("                if (gi < 0) {\n                    // mjc_PlaneConvex:", "                if (gi < 0) {\n                    const unsigned long long c1_ = __builtin_readcyclecounter();\n                    // mjc_PlaneConvex:")
    rep("                    ncon += min(cnt, max(room, 0));\n", "                    ncon += min(cnt, max(room, 0));\n                    np[4] += __builtin_readcyclecounter() - c1_;\n")
    rep("            unsigned long long geoms = __ballot(gbits != 0u);", "            const unsigned long long c2_ = __builtin_readcyclecounter();\n            unsigned long long geoms = __ballot(gbits != 0u);")
    rep("            while (slot_done < D_MAXOBJ) { slot_done++; if (tid == 0) s.con_start[D_NB + slot_done] = ncon; }\n        }",
        "            while (slot_done < D_MAXOBJ) { slot_done++; if (tid == 0) s.con_start[D_NB + slot_done] = ncon; }\n            np[3] += __builtin_readcyclecounter() - c2_;\n        }")
    rep("                            if (tid == 0) n = box_box(g, h, P.margin, rec, s.U);\n                            n = __builtin_amdgcn_readfirstlane(n);",
        "                            const unsigned long long c3_ = __builtin_readcyclecounter();\n                            if (tid == 0) n = box_box(g, h, P.margin, rec, s.U);\n                            n = __builtin_amdgcn_readfirstlane(n);\n"
        "                            np[5]++; np[6] += __builtin_readcyclecounter() - c3_;")
    rep("        collide<NT, OBJ>(s, T, P, tid);", "        collide<NT, OBJ>(s, T, P, tid, pc);")
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    return s


def patch_b(s):
    """variant B: [0] broad phase (per-hull target masks) cycles, [1] per-hull set-up (pose, vertex transform) cycles, [2] separating-plane cull cycles,
    [3] pairs that reach the cull, [4] hulls visited, [5] whole collide() cycles, [7] control-step total"""
    def rep(a, b):
        nonlocal s
        assert s.count(a) == 1, (s.count(a), a[:90])
        s = s.replace(a, b, 1)
    rep("__device__ __forceinline__ void collide(EnvLds& s, const DevTables& T, const Params& P, int tid) {",
        "__device__ __forceinline__ void collide(EnvLds& s, const DevTables& T, const Params& P, int tid, unsigned long long* np) {\n    const unsigned long long cA_ = __builtin_readcyclecounter();")
    rep("        unsigned long long bodies = __ballot(mybits != 0u);\n", "        np[0] += __builtin_readcyclecounter() - cA_;\n        unsigned long long bodies = __ballot(mybits != 0u);\n")
    rep("            const int b = __ffsll((long long)bodies) - 1;\n            bodies &= bodies - 1ull;\n            unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)mybits, b);",
        "            const unsigned long long cB_ = __builtin_readcyclecounter();\n            const int b = __ffsll((long long)bodies) - 1;\n            bodies &= bodies - 1ull;\n            unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)mybits, b);")
    rep("            if (tid < nvb) { v = ld3(T.verts + 3 * (vadr + tid)); xw = xb + mulmat(R, v); }\n", "            if (tid < nvb) { v = ld3(T.verts + 3 * (vadr + tid)); xw = xb + mulmat(R, v); }\n            np[1] += __builtin_readcyclecounter() - cB_; np[4]++;\n")
    rep("                    {\n                        const float* Rg = g + 7;\n                        const V3 dv = xw - ld3(g + 4);", "                    const unsigned long long cC_ = __builtin_readcyclecounter();\n                    np[3]++;\n                    {\n                        const float* Rg = g + 7;\n                        const V3 dv = xw - ld3(g + 4);")
    rep("                        if (sep) continue;\n                    }", "                        np[2] += __builtin_readcyclecounter() - cC_;\n                        if (sep) continue;\n                    }")
    rep("        if (tid == 0) { s.ncon = ncon; s.nlim = 0; }\n    }\n    KP_SYNC();\n}\n\n// efc_D and the reference acceleration", "        if (tid == 0) { s.ncon = ncon; s.nlim = 0; }\n        np[5] += __builtin_readcyclecounter() - cA_;\n    }\n    KP_SYNC();\n}\n\n// efc_D and the reference acceleration")
    rep("        collide<NT, OBJ>(s, T, P, tid);", "        collide<NT, OBJ>(s, T, P, tid, pc);")
    rep("#define KP_T(i) if (prof) { t1 = __builtin_readcyclecounter(); pc[i] += t1 - t0; t0 = t1; }", "#define KP_T(i)")
    return s


def main(out, variant="A"):
    tmp = tempfile.mkdtemp(prefix="kp_collide_instr_")
    src = os.path.join(tmp, "kinpoly_amd", "csrc")
    shutil.copytree(os.path.join(ROOT, "kinpoly_amd", "csrc"), src)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(src, "kp_step_kernel.hpp")
    text = (patch_b if variant == "B" else patch)(open(p).read())
    open(p, "w").write(text)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", os.path.join(src, "kp_sim.hip"), "-o", out])
    shutil.rmtree(tmp)
    print("built", out)


if __name__ == "__main__":
    main(*sys.argv[1:])
