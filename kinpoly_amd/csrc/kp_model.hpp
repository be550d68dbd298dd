// kp_model.hpp -- host-side compiled-model (KPM blob) loader + device table builder.
//
// Replaces what `mujoco_py.load_model_from_path` gives the reference
// (uhc/khrylib/rl/envs/common/mujoco_env.py:23-24): the KPM blob is produced by
// kinpoly_amd/model_compiler.py from the same XML + STL hulls.  Everything here is plain C++
// (no torch types); device uploads happen in kp_sim.hip.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace kp {

constexpr int NB = 24, NV = 75, NQ = 76, NU = 69, NM = 1221, MAXDEPTH = 30;

struct HostModel {
    int nb = 0, nv = 0, nq = 0, nu = 0, nM = 0, nvert = 0;
    std::vector<int> body_parent, body_depth, body_subtree, dof_body, dof_parent, dof_depth, dof_madr, jnt_limited, vert_adr, obj_geom_adr, vert_nbr_adr, vert_nbr;
    std::vector<double> body_pos, body_ipos, body_mass, body_inertia, body_rbound, body_invweight0, dof_invweight0,
        dof_armature, jnt_range, verts, kp, kd, torque_lim, a_scale, opt, body_diffw, obj_geoms, obj_mass, obj_inertial, mesh_rbound, planemesh;
    std::string error;
};

struct KpmEntry { char name[32]; uint32_t dtype, pad; uint64_t count, off; };

inline bool kpm_get(const std::vector<unsigned char>& buf, const char* name, std::vector<double>* f, std::vector<int>* iv) {
    uint32_t n; std::memcpy(&n, buf.data() + 8, 4);
    for (uint32_t i = 0; i < n; i++) {
        KpmEntry e; std::memcpy(&e, buf.data() + 12 + 56 * i, 56);
        if (std::strncmp(e.name, name, 32)) continue;
        if (e.off + e.count * (e.dtype == 0 ? 8 : 4) > buf.size()) return false;
        if (e.dtype == 0 && f) { f->resize(e.count); std::memcpy(f->data(), buf.data() + e.off, 8 * e.count); return true; }
        if (e.dtype == 1 && iv) { iv->resize(e.count); std::memcpy(iv->data(), buf.data() + e.off, 4 * e.count); return true; }
        return false;
    }
    return false;
}

inline bool load_kpm(const char* path, HostModel& m) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { m.error = std::string("cannot open ") + path; return false; }
    std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> buf(sz);
    size_t rd = std::fread(buf.data(), 1, sz, f); std::fclose(f);
    if (rd != (size_t)sz || sz < 12) { m.error = "short read"; return false; }
    uint32_t magic; std::memcpy(&magic, buf.data(), 4);
    if (magic != 0x314D504Bu) { m.error = "not a KPM1 blob"; return false; }
    std::vector<int> dims;
#define KPF(field, nm) if (!kpm_get(buf, nm, &m.field, nullptr)) { m.error = std::string("missing field ") + nm; return false; }
#define KPI(field, nm) if (!kpm_get(buf, nm, nullptr, &m.field)) { m.error = std::string("missing field ") + nm; return false; }
    if (!kpm_get(buf, "dims", nullptr, &dims) || dims.size() < 6) { m.error = "missing dims"; return false; }
    m.nb = dims[0]; m.nv = dims[1]; m.nq = dims[2]; m.nu = dims[3]; m.nM = dims[4]; m.nvert = dims[5];
    if (m.nb != NB || m.nv != NV || m.nq != NQ || m.nu != NU || m.nM != NM) { m.error = "model dims differ from the compiled-in SMPL layout"; return false; }
    KPI(body_parent, "body_parent") KPI(body_depth, "body_depth") KPI(body_subtree, "body_subtree")
    KPI(dof_body, "dof_body") KPI(dof_parent, "dof_parent") KPI(dof_depth, "dof_depth") KPI(dof_madr, "dof_madr")
    KPI(jnt_limited, "jnt_limited") KPI(vert_adr, "vert_adr")
    if (!kpm_get(buf, "vert_nbr_adr", nullptr, &m.vert_nbr_adr) || !kpm_get(buf, "vert_nbr", nullptr, &m.vert_nbr) || (int)m.vert_nbr_adr.size() != m.nvert + 1) {
        m.error = "blob has no hull graph (vert_nbr_adr / vert_nbr): recompile the model with kinpoly_amd/model_compiler.py (KPM version 6)"; return false; }
    KPF(body_pos, "body_pos") KPF(body_ipos, "body_ipos") KPF(body_mass, "body_mass") KPF(body_inertia, "body_inertia")
    KPF(body_rbound, "body_rbound")
    if (!kpm_get(buf, "mesh_rbound", &m.mesh_rbound, nullptr) || (int)m.mesh_rbound.size() != m.nb) {
        m.error = "blob has no mesh_rbound (geom_rbound of the hull meshes): recompile the model with kinpoly_amd/model_compiler.py (KPM version 7)"; return false; }
    if (!kpm_get(buf, "planemesh", &m.planemesh, nullptr) || m.planemesh.size() != 2) m.planemesh = {3.0, 0.3};   // mjc_PlaneConvex: maxplanemesh, tolplanemesh
    KPF(body_invweight0, "body_invweight0") KPF(dof_invweight0, "dof_invweight0")
    KPF(dof_armature, "dof_armature") KPF(jnt_range, "jnt_range") KPF(verts, "verts")
    KPF(kp, "kp") KPF(kd, "kd") KPF(torque_lim, "torque_lim") KPF(a_scale, "a_scale") KPF(opt, "opt") KPF(body_diffw, "body_diffw")
    // free objects (optional): [ngeom, 18] = object id, type, size3, local pos3, local R9, mass; adr [nobj + 1]
    if (!kpm_get(buf, "obj_geoms", &m.obj_geoms, nullptr) || !kpm_get(buf, "obj_geom_adr", nullptr, &m.obj_geom_adr) ||
        !kpm_get(buf, "obj_mass", &m.obj_mass, nullptr)) { m.obj_geoms.clear(); m.obj_geom_adr.clear(); m.obj_mass.clear(); }
    if (!kpm_get(buf, "obj_inertial", &m.obj_inertial, nullptr)) m.obj_inertial.clear();
#undef KPF
#undef KPI
    if (m.opt.size() < 25) { m.error = "opt too short"; return false; }
    for (int d = 0; d < NV; d++) if (m.dof_depth[d] >= MAXDEPTH) { m.error = "dof tree too deep"; return false; }
    for (int b = 0; b < NB; b++) if (m.vert_adr[b + 1] - m.vert_adr[b] > 64) { m.error = "hull with more than 64 vertices"; return false; }
    return true;
}

// opt[] indices (kinpoly_amd/model_compiler.py OPT_FIELDS)
enum { OPT_TIMESTEP = 0, OPT_GX, OPT_GY, OPT_GZ, OPT_SOLREF_TC, OPT_SOLREF_DR, OPT_IMP_D0, OPT_IMP_DW, OPT_IMP_W, OPT_IMP_MID,
       OPT_IMP_POW, OPT_FRIC, OPT_FRIC_SPIN, OPT_FRIC_ROLL, OPT_MARGIN, OPT_IMPRATIO, OPT_MEANINERTIA, OPT_RFC_SCALE, OPT_RFC_LIM,
       OPT_BR_W, OPT_BR_X, OPT_BR_Y, OPT_BR_Z, OPT_SOLVER_ITER, OPT_SOLVER_TOL, OPT_NV_FULL };

}  // namespace kp
