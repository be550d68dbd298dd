// kp_compile.hpp -- the model compiler behind the C ABI: XML + STL meshes + uhc.yml -> KPM blob, no Python.
//
// Replaces what `mujoco_py.load_model_from_path(xml)` does for the reference (uhc/khrylib/rl/envs/common/mujoco_env.py:23) for the two
// scenes the path uses (assets/mujoco_models/humanoid_smpl_neutral_mesh_all.xml, ..._all_step.xml): it restates
// kinpoly_amd/model_compiler.py step for step (same field order, same rules, IEEE double arithmetic without contraction), so that both
// write the same blob -- integer tables identical, floating-point fields equal to rounding (the Python side goes through BLAS / LAPACK
// for the 75 x 75 inverse and the 3 x 3 eigenvectors).  tests/test_host_cpu.py compares the two.  Plain host C++17, no dependencies:
// a subset XML reader (elements, attributes, comments, self-closing tags), a line reader for the few uhc.yml keys, binary STL.
#pragma once
#pragma clang fp contract(off)
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

namespace kpc {

// ------------------------------------------------------------------------------------------------ XML subset
struct Xml {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attr;
    std::vector<Xml> kids;
    const std::string* get(const char* k) const { for (auto& a : attr) if (a.first == k) return &a.second; return nullptr; }
    std::string gets(const char* k, const std::string& dflt = "") const { auto* p = get(k); return p ? *p : dflt; }
    const Xml* find(const char* n) const { for (auto& c : kids) if (c.name == n) return &c; return nullptr; }
    std::vector<const Xml*> findall(const char* n) const { std::vector<const Xml*> r; for (auto& c : kids) if (c.name == n) r.push_back(&c); return r; }
};

inline bool xml_parse(const std::string& t, Xml& root, std::string& err) {
    std::vector<Xml*> stack;
    Xml top; top.name = "#doc";
    stack.push_back(&top);
    size_t i = 0, n = t.size();
    auto skip_ws = [&]() { while (i < n && std::isspace((unsigned char)t[i])) i++; };
    while (i < n) {
        size_t lt = t.find('<', i);
        if (lt == std::string::npos) break;
        i = lt;
        if (t.compare(i, 4, "<!--") == 0) { size_t e = t.find("-->", i); if (e == std::string::npos) { err = "unterminated comment"; return false; } i = e + 3; continue; }
        if (t.compare(i, 2, "<?") == 0) { size_t e = t.find("?>", i); if (e == std::string::npos) { err = "unterminated declaration"; return false; } i = e + 2; continue; }
        if (t.compare(i, 2, "</") == 0) {
            size_t e = t.find('>', i); if (e == std::string::npos || stack.size() < 2) { err = "bad closing tag"; return false; }
            stack.pop_back(); i = e + 1; continue;
        }
        i++;
        size_t s0 = i;
        while (i < n && !std::isspace((unsigned char)t[i]) && t[i] != '>' && t[i] != '/') i++;
        Xml el; el.name = t.substr(s0, i - s0);
        bool selfclose = false;
        for (;;) {
            skip_ws();
            if (i >= n) { err = "unterminated tag"; return false; }
            if (t[i] == '/') { selfclose = true; i++; continue; }
            if (t[i] == '>') { i++; break; }
            size_t k0 = i;
            while (i < n && t[i] != '=' && !std::isspace((unsigned char)t[i])) i++;
            std::string key = t.substr(k0, i - k0);
            skip_ws();
            if (i >= n || t[i] != '=') { err = "attribute without value: " + key; return false; }
            i++; skip_ws();
            if (i >= n || (t[i] != '"' && t[i] != '\'')) { err = "unquoted attribute value: " + key; return false; }
            const char q = t[i++];
            size_t v0 = i;
            while (i < n && t[i] != q) i++;
            el.attr.emplace_back(key, t.substr(v0, i - v0));
            i++;
        }
        stack.back()->kids.push_back(std::move(el));
        if (!selfclose) stack.push_back(&stack.back()->kids.back());
    }
    if (top.kids.empty()) { err = "no root element"; return false; }
    root = std::move(top.kids[0]);
    return true;
}

inline std::vector<double> floats(const std::string& s) {
    std::vector<double> v;
    const char* p = s.c_str();
    while (*p) {
        while (*p && (std::isspace((unsigned char)*p) || *p == ',')) p++;
        if (!*p) break;
        char* e = nullptr;
        double x = std::strtod(p, &e);
        if (e == p) break;
        v.push_back(x); p = e;
    }
    return v;
}

inline bool read_file(const std::string& path, std::string& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    out.resize((size_t)sz);
    size_t rd = sz ? std::fread(&out[0], 1, (size_t)sz, f) : 0;
    std::fclose(f);
    return rd == (size_t)sz;
}

using V3d = std::array<double, 3>;
using M3d = std::array<double, 9>;
inline V3d sub(const V3d& a, const V3d& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline V3d add(const V3d& a, const V3d& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline V3d cross(const V3d& a, const V3d& b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
inline double dot(const V3d& a, const V3d& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// ------------------------------------------------------------------------------------------------ blob writer (model_compiler.write_kpm)
struct Blob {
    struct Entry { std::string name; int dtype; std::vector<double> f; std::vector<int32_t> i; };
    std::vector<Entry> e;
    void addf(const char* n, const std::vector<double>& v) { e.push_back({n, 0, v, {}}); }
    void addi(const char* n, const std::vector<int32_t>& v) { e.push_back({n, 1, {}, v}); }
    bool write(const std::string& path, uint32_t version) const {
        const uint32_t magic = 0x314D504Bu, cnt = (uint32_t)e.size();
        size_t off = (12 + 56 * e.size() + 7) / 8 * 8;
        std::vector<unsigned char> buf;
        std::vector<uint64_t> offs;
        for (auto& x : e) { offs.push_back(off); size_t bytes = x.dtype == 0 ? 8 * x.f.size() : 4 * x.i.size(); off = (off + bytes + 7) / 8 * 8; }
        buf.assign(off, 0);
        std::memcpy(&buf[0], &magic, 4); std::memcpy(&buf[4], &version, 4); std::memcpy(&buf[8], &cnt, 4);
        for (size_t k = 0; k < e.size(); k++) {
            unsigned char* t = &buf[12 + 56 * k];
            std::memset(t, 0, 56);
            std::memcpy(t, e[k].name.c_str(), std::min<size_t>(e[k].name.size(), 32));
            uint32_t dt = (uint32_t)e[k].dtype, pad = 0;
            uint64_t c = e[k].dtype == 0 ? e[k].f.size() : e[k].i.size(), o = offs[k];
            std::memcpy(t + 32, &dt, 4); std::memcpy(t + 36, &pad, 4); std::memcpy(t + 40, &c, 8); std::memcpy(t + 48, &o, 8);
            if (e[k].dtype == 0) { if (c) std::memcpy(&buf[o], e[k].f.data(), 8 * c); } else if (c) std::memcpy(&buf[o], e[k].i.data(), 4 * c);
        }
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) return false;
        const bool ok = std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
        std::fclose(f);
        return ok;
    }
};

// ------------------------------------------------------------------------------------------------ meshes
// triangles of a binary STL as doubles (the file holds float32)
inline bool read_stl(const std::string& path, std::vector<std::array<V3d, 3>>& tris) {
    std::string b;
    if (!read_file(path, b) || b.size() < 84) return false;
    uint32_t n; std::memcpy(&n, &b[80], 4);
    if (b.size() < 84 + 50ull * n) return false;
    tris.resize(n);
    for (uint32_t t = 0; t < n; t++)
        for (int v = 0; v < 3; v++)
            for (int k = 0; k < 3; k++) { float x; std::memcpy(&x, &b[84 + 50ull * t + 12 + 12 * v + 4 * k], 4); tris[t][v][k] = (double)x; }
    return true;
}

// model_compiler.polyhedron_mass_props: signed tetrahedra against the vertex centroid with |volume| per tetrahedron
inline void mass_props(const std::vector<std::array<V3d, 3>>& tris, double density, double& mass, V3d& com, M3d& inertia) {
    V3d ref = {0, 0, 0};
    for (auto& t : tris) for (int v = 0; v < 3; v++) ref = add(ref, t[v]);
    const double n3 = 3.0 * (double)tris.size();
    ref = {ref[0] / n3, ref[1] / n3, ref[2] / n3};
    double V = 0; V3d cr = {0, 0, 0}; double C[9] = {0};
    for (auto& t : tris) {
        const V3d a = sub(t[0], ref), b = sub(t[1], ref), c = sub(t[2], ref);
        const double vol = std::fabs(dot(a, cross(b, c))) / 6.0;
        V += vol;
        for (int k = 0; k < 3; k++) cr[k] += vol * ((a[k] + b[k] + c[k]) / 4.0);
        const V3d s = {a[0] + b[0] + c[0], a[1] + b[1] + c[1], a[2] + b[2] + c[2]};
        const V3d* vs[4] = {&a, &b, &c, &s};
        for (int q = 0; q < 4; q++) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3 * i + j] += (vol / 20.0) * (*vs[q])[i] * (*vs[q])[j];
    }
    const V3d cm = {cr[0] / V, cr[1] / V, cr[2] / V};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3 * i + j] -= V * cm[i] * cm[j];
    const double tr = C[0] + C[4] + C[8];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) inertia[3 * i + j] = ((i == j ? tr : 0.0) - C[3 * i + j]) * density;
    mass = V * density; com = add(ref, cm);
}

// eigenvectors of a symmetric 3 x 3 (cyclic Jacobi); columns of `vec`
inline void eigh3(const M3d& A, M3d& vec) {
    double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = A[3 * i + j];
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-40) break;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            if (std::fabs(a[p][q]) < 1e-300) continue;
            const double th = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
            const double tt = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
            const double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
            for (int k = 0; k < 3; k++) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
            for (int k = 0; k < 3; k++) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
            for (int k = 0; k < 3; k++) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
        }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) vec[3 * i + j] = v[i][j];
}

// model_compiler.hull_graph: faces = vertex triples with every other vertex on one side (1e-9 m counts as on the plane), boundary edges +
// fan diagonals from the lowest-numbered vertex, neighbours in ascending vertex number
inline bool hull_graph(const std::vector<V3d>& v, std::vector<std::vector<int>>& lists) {
    const int n = (int)v.size();
    if (n < 4 || n > 64) return false;
    const double tol = 1e-9, PI = 3.14159265358979323846;
    std::vector<char> adj((size_t)n * n, 0);
    std::set<std::vector<int>> seen;
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) {
        const V3d eij = sub(v[j], v[i]);
        for (int k = j + 1; k < n; k++) {
            const V3d eik = sub(v[k], v[i]);
            double nx = eij[1] * eik[2] - eij[2] * eik[1], ny = eij[2] * eik[0] - eij[0] * eik[2], nz = eij[0] * eik[1] - eij[1] * eik[0];
            const double ln = std::sqrt(nx * nx + ny * ny + nz * nz);
            if (ln < 1e-14) continue;
            nx /= ln; ny /= ln; nz /= ln;
            bool pos = false, neg = false;
            std::vector<int> face;
            for (int m = 0; m < n; m++) {
                const double d = nx * (v[m][0] - v[i][0]) + ny * (v[m][1] - v[i][1]) + nz * (v[m][2] - v[i][2]);
                if (d > tol) pos = true; else if (d < -tol) neg = true; else face.push_back(m);
                if (pos && neg) break;
            }
            if (pos && neg) continue;
            if (!seen.insert(face).second) continue;
            if (pos) { nx = -nx; ny = -ny; nz = -nz; }
            double cx = 0, cy = 0, cz = 0;
            for (int m : face) { cx += v[m][0]; cy += v[m][1]; cz += v[m][2]; }
            cx /= (double)face.size(); cy /= (double)face.size(); cz /= (double)face.size();
            const int f0 = face[0];
            const double ux = v[f0][0] - cx, uy = v[f0][1] - cy, uz = v[f0][2] - cz;
            const double wx = ny * uz - nz * uy, wy = nz * ux - nx * uz, wz = nx * uy - ny * ux;
            std::vector<std::pair<double, int>> ang;
            for (int m : face) {
                if (m == f0) continue;
                const double px = v[m][0] - cx, py = v[m][1] - cy, pz = v[m][2] - cz;
                double a = std::atan2(px * wx + py * wy + pz * wz, px * ux + py * uy + pz * uz);
                if (!(a > 0)) a += 2 * PI;
                ang.emplace_back(a, m);
            }
            std::sort(ang.begin(), ang.end());
            std::vector<int> ring = {f0};
            for (auto& p : ang) ring.push_back(p.second);
            const int L = (int)ring.size();
            for (int t = 0; t < L; t++) { const int a = ring[t], b = ring[(t + 1) % L]; adj[(size_t)a * n + b] = adj[(size_t)b * n + a] = 1; }
            for (int t = 2; t < L - 1; t++) adj[(size_t)f0 * n + ring[t]] = adj[(size_t)ring[t] * n + f0] = 1;
        }
    }
    lists.assign(n, {});
    for (int i = 0; i < n; i++) { for (int m = 0; m < n; m++) if (adj[(size_t)i * n + m]) lists[i].push_back(m); if (lists[i].size() < 3) return false; }
    return true;
}

// dense symmetric positive definite inverse (Gauss-Jordan with partial pivoting; n <= 80)
inline bool invert(std::vector<double> A, int n, std::vector<double>& inv) {
    inv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) inv[(size_t)i * n + i] = 1.0;
    for (int c = 0; c < n; c++) {
        int p = c; double best = std::fabs(A[(size_t)c * n + c]);
        for (int r = c + 1; r < n; r++) if (std::fabs(A[(size_t)r * n + c]) > best) { best = std::fabs(A[(size_t)r * n + c]); p = r; }
        if (best < 1e-300) return false;
        if (p != c) for (int k = 0; k < n; k++) { std::swap(A[(size_t)p * n + k], A[(size_t)c * n + k]); std::swap(inv[(size_t)p * n + k], inv[(size_t)c * n + k]); }
        const double d = 1.0 / A[(size_t)c * n + c];
        for (int k = 0; k < n; k++) { A[(size_t)c * n + k] *= d; inv[(size_t)c * n + k] *= d; }
        for (int r = 0; r < n; r++) if (r != c) {
            const double f = A[(size_t)r * n + c];
            if (f == 0.0) continue;
            for (int k = 0; k < n; k++) { A[(size_t)r * n + k] -= f * A[(size_t)c * n + k]; inv[(size_t)r * n + k] -= f * inv[(size_t)c * n + k]; }
        }
    }
    return true;
}

inline M3d euler_deg_to_mat(const std::vector<double>& e) {      // MuJoCo default eulerseq 'xyz' (intrinsic): R = Rx Ry Rz
    const double R2D = 3.14159265358979323846 / 180.0;
    const double ax = e[0] * R2D, ay = e[1] * R2D, az = e[2] * R2D;
    const double cx = std::cos(ax), sx = std::sin(ax), cy = std::cos(ay), sy = std::sin(ay), cz = std::cos(az), sz = std::sin(az);
    const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    double T[9]; M3d R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Rx[3 * i + k] * Ry[3 * k + j]; T[3 * i + j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += T[3 * i + k] * Rz[3 * k + j]; R[3 * i + j] = s; }
    return R;
}

// ------------------------------------------------------------------------------------------------ uhc.yml (the keys the compiler reads)
struct UhcCfg {
    bool have = false;
    std::vector<std::string> joint_names, body_names;
    std::vector<std::array<double, 5>> joint_rows;      // k_p, k_d, a_ref, a_scale, torque_limit
    std::vector<double> body_w;
    double rfc_scale = 200.0, rfc_lim = 100.0;          // cfg.get(..., default) of model_compiler.compile_model
    std::vector<double> base_rot = {0.7071, 0.7071, 0.0, 0.0};
};
inline bool parse_uhc_yml(const std::string& path, UhcCfg& c, std::string& err) {
    std::string t;
    if (!read_file(path, t)) { err = "cannot read " + path; return false; }
    c.have = true;
    std::string section;
    size_t i = 0;
    while (i < t.size()) {
        size_t e = t.find('\n', i); if (e == std::string::npos) e = t.size();
        std::string line = t.substr(i, e - i); i = e + 1;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        size_t a = line.find_first_not_of(" \t\r");
        if (a == std::string::npos) continue;
        const std::string body = line.substr(a);
        if (body[0] == '-') {                                   // list row: - ["name", numbers ...]
            const size_t lb = body.find('['), rb = body.rfind(']');
            if (lb == std::string::npos || rb == std::string::npos) continue;
            const std::string inner = body.substr(lb + 1, rb - lb - 1);
            const size_t q0 = inner.find_first_of("\"'");
            if (q0 == std::string::npos) continue;
            const size_t q1 = inner.find(inner[q0], q0 + 1);
            const std::string name = inner.substr(q0 + 1, q1 - q0 - 1);
            const std::vector<double> nums = floats(inner.substr(q1 + 1));
            if (section == "joint_params") { if (nums.size() < 5) { err = "joint_params row too short"; return false; } c.joint_names.push_back(name); c.joint_rows.push_back({nums[0], nums[1], nums[2], nums[3], nums[4]}); }
            else if (section == "body_params") { if (nums.empty()) { err = "body_params row too short"; return false; } c.body_names.push_back(name); c.body_w.push_back(nums[0]); }
            continue;
        }
        const size_t col = body.find(':');
        if (col == std::string::npos) continue;
        const std::string key = body.substr(0, col);
        std::string val = body.substr(col + 1);
        const size_t v0 = val.find_first_not_of(" \t\r");
        val = v0 == std::string::npos ? "" : val.substr(v0);
        if (a == 0) section = val.empty() ? key : "";          // a top-level key opens a section when it has no inline value
        if (key == "residual_force_scale" && a == 0) c.rfc_scale = std::strtod(val.c_str(), nullptr);
        else if (key == "residual_force_lim" && a == 0) c.rfc_lim = std::strtod(val.c_str(), nullptr);
        else if (key == "base_rot" && section == "data_specs") { const size_t lb = val.find('['), rb = val.rfind(']'); if (lb != std::string::npos && rb != std::string::npos) c.base_rot = floats(val.substr(lb + 1, rb - lb - 1)); }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ the compiler
struct BodyRec { std::string name; int parent; V3d gpos; std::vector<const Xml*> joints; const Xml* geom; };

inline std::string attr_or(const Xml* el, const Xml* dflt, const char* key, const std::string& fallback) {
    if (const std::string* p = el->get(key)) return *p;
    if (dflt) if (const std::string* p = dflt->get(key)) return *p;
    return fallback;
}

inline int compile(const std::string& xml_path, const char* yml_path, const std::string& out_path, std::string& err) {
    const double density = 1000.0, gravity[3] = {0.0, 0.0, -9.81}, solref[2] = {0.02, 1.0}, solimp[5] = {0.9, 0.95, 0.001, 0.5, 2.0},
                 geom_friction[3] = {1.0, 0.005, 0.0001}, impratio = 1.0, solver_iterations = 100, solver_tolerance = 1e-8, maxplanemesh = 3, tolplanemesh = 0.3;
    std::string text;
    if (!read_file(xml_path, text)) { err = "cannot read " + xml_path; return -1; }
    Xml root;
    if (!xml_parse(text, root, err)) return -1;
    const Xml* comp = root.find("compiler");
    if (!comp || comp->gets("coordinate") != "global" || comp->gets("angle") != "degree" || comp->gets("inertiafromgeom") != "true") { err = "compiler element: need coordinate=global angle=degree inertiafromgeom=true"; return -1; }
    const Xml* dflt = root.find("default");
    const Xml* jd = dflt ? dflt->find("joint") : nullptr;
    const Xml* gd = dflt ? dflt->find("geom") : nullptr;
    const Xml* opt = root.find("option");
    const Xml* asset = root.find("asset");
    const Xml* wb = root.find("worldbody");
    if (!opt || !asset || !wb || !opt->get("timestep")) { err = "missing option / asset / worldbody"; return -1; }
    const double timestep = std::strtod(opt->gets("timestep").c_str(), nullptr);
    std::string base = xml_path; { const size_t sl = base.find_last_of('/'); base = sl == std::string::npos ? "." : base.substr(0, sl); }
    std::map<std::string, std::string> meshes;
    for (const Xml* m : asset->findall("mesh")) {
        const std::string f = m->gets("file");
        std::string nm = f; { const size_t sl = nm.find_last_of('/'); if (sl != std::string::npos) nm = nm.substr(sl + 1); const size_t dt = nm.find_last_of('.'); if (dt != std::string::npos) nm = nm.substr(0, dt); }
        meshes[m->gets("name", nm)] = base + "/" + f;
    }
    const Xml* floor = nullptr;
    for (const Xml* g : wb->findall("geom")) if (g->gets("type") == "plane") floor = g;
    if (!floor) { err = "no plane geom in worldbody"; return -1; }

    std::vector<BodyRec> bodies;
    struct ObjRec { std::string name; std::vector<const Xml*> geoms; };
    std::vector<ObjRec> objects;
    struct Walker { std::vector<BodyRec>& B; void walk(const Xml* el, int parent) {
        const int idx = (int)B.size();
        const std::vector<double> p = floats(el->gets("pos"));
        BodyRec r; r.name = el->gets("name"); r.parent = parent; r.gpos = {p.size() > 0 ? p[0] : 0, p.size() > 1 ? p[1] : 0, p.size() > 2 ? p[2] : 0};
        r.joints = el->findall("joint");
        const auto gs = el->findall("geom"); r.geom = gs.empty() ? nullptr : gs[0];
        if (gs.size() != 1) r.geom = nullptr;
        B.push_back(r);
        for (const Xml* ch : el->findall("body")) walk(ch, idx);
    } } walker{bodies};
    for (const Xml* top : wb->findall("body")) {
        bool humanoid = false;
        for (const Xml* g : top->findall("geom")) if (g->gets("type") == "mesh") humanoid = true;
        if (humanoid) walker.walk(top, -1);
        else objects.push_back({top->gets("name"), top->findall("geom")});
    }
    const int nb = (int)bodies.size();
    if (nb < 1) { err = "no humanoid body"; return -1; }

    std::vector<int32_t> parent(nb);
    std::vector<double> gpos(3 * nb), body_pos(3 * nb), mass(nb), ipos(3 * nb), rbound(nb), mesh_rbound(nb), verts;
    std::vector<M3d> inertia(nb);
    std::vector<int32_t> vert_adr = {0}, nbr_adr = {0}, nbr;
    for (int i = 0; i < nb; i++) { parent[i] = bodies[i].parent; for (int k = 0; k < 3; k++) gpos[3 * i + k] = bodies[i].gpos[k]; }
    for (int i = 0; i < nb; i++) for (int k = 0; k < 3; k++) body_pos[3 * i + k] = parent[i] >= 0 ? gpos[3 * i + k] - gpos[3 * parent[i] + k] : gpos[3 * i + k];
    for (int i = 0; i < nb; i++) {
        const Xml* g = bodies[i].geom;
        if (!g || attr_or(g, gd, "type", "") != "mesh") { err = "body " + bodies[i].name + ": expected exactly one mesh geom"; return -1; }
        auto it = meshes.find(g->gets("mesh"));
        if (it == meshes.end()) { err = "unknown mesh " + g->gets("mesh"); return -1; }
        std::vector<std::array<V3d, 3>> tris;
        if (!read_stl(it->second, tris)) { err = "cannot read binary STL " + it->second; return -1; }
        double m; V3d com; M3d I;
        mass_props(tris, density, m, com, I);
        mass[i] = m; inertia[i] = I;
        for (int k = 0; k < 3; k++) ipos[3 * i + k] = com[k] - gpos[3 * i + k];
        std::vector<V3d> uv;                                   // np.unique(rows): lexicographic order, exact duplicates removed
        for (auto& t : tris) for (int v = 0; v < 3; v++) uv.push_back(t[v]);
        std::sort(uv.begin(), uv.end());
        uv.erase(std::unique(uv.begin(), uv.end()), uv.end());
        std::vector<V3d> v(uv.size());
        double rb = 0;
        for (size_t q = 0; q < uv.size(); q++) { v[q] = sub(uv[q], bodies[i].gpos); rb = std::max(rb, std::sqrt(dot(v[q], v[q]))); for (int k = 0; k < 3; k++) verts.push_back(v[q][k]); }
        if (v.size() > 64) { err = "hull with more than 64 vertices: " + bodies[i].name; return -1; }
        vert_adr.push_back(vert_adr.back() + (int32_t)v.size());
        rbound[i] = rb;
        M3d axes; eigh3(I, axes);
        double half[3] = {0, 0, 0};
        for (auto& p : v) { const V3d d = sub(add(p, bodies[i].gpos), com); for (int c = 0; c < 3; c++) half[c] = std::max(half[c], std::fabs(d[0] * axes[c] + d[1] * axes[3 + c] + d[2] * axes[6 + c])); }
        mesh_rbound[i] = std::sqrt(half[0] * half[0] + half[1] * half[1] + half[2] * half[2]);
        std::vector<std::vector<int>> lists;
        if (!hull_graph(v, lists)) { err = "mesh " + bodies[i].name + " has vertices inside its convex hull (or fewer than 4 / more than 64)"; return -1; }
        for (auto& l : lists) { for (int x : l) nbr.push_back(x); nbr_adr.push_back((int32_t)nbr.size()); }
    }
    // ---- dofs
    const double D2R = 3.14159265358979323846 / 180.0;
    std::vector<int32_t> dof_body, dof_trans, jlimited;
    std::vector<V3d> dof_axis;
    std::vector<double> arm, jrange;
    for (int i = 0; i < nb; i++) for (const Xml* j : bodies[i].joints) {
        const std::string type = attr_or(j, jd, "type", "hinge");
        if (type == "free") {
            const double a = std::strtod(attr_or(j, jd, "armature", "0").c_str(), nullptr);
            for (int tr = 1; tr >= 0; tr--) for (int k = 0; k < 3; k++) { dof_body.push_back(i); dof_axis.push_back({k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0}); dof_trans.push_back(tr); arm.push_back(a); }
        } else if (type == "hinge") {
            const std::vector<double> p = floats(j->gets("pos")), ax = floats(j->gets("axis")), r = floats(attr_or(j, jd, "range", "0 0"));
            if (p.size() != 3 || ax.size() != 3 || r.size() != 2) { err = "hinge needs pos / axis / range"; return -1; }
            for (int k = 0; k < 3; k++) if (std::fabs(p[k] - bodies[i].gpos[k]) > 1e-8 + 1e-5 * std::fabs(bodies[i].gpos[k])) { err = "hinge not anchored at its body origin: " + j->gets("name"); return -1; }
            dof_body.push_back(i); dof_axis.push_back({ax[0], ax[1], ax[2]}); dof_trans.push_back(0);
            arm.push_back(std::strtod(attr_or(j, jd, "armature", "0").c_str(), nullptr));
            jrange.push_back(r[0] * D2R); jrange.push_back(r[1] * D2R);
            jlimited.push_back(attr_or(j, jd, "limited", "false") == "true" ? 1 : 0);
        } else { err = "unsupported joint type " + type; return -1; }
    }
    const int nv = (int)dof_body.size(), nu = nv - 6;
    for (int i = 1; i < nb; i++) for (int k = 0; k < 3; k++) {
        const V3d want = {k == 2 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 0 ? 1.0 : 0.0};
        const int d = 6 + 3 * (i - 1) + k;
        if (d >= nv || dof_body[d] != i || std::fabs(dof_axis[d][0] - want[0]) + std::fabs(dof_axis[d][1] - want[1]) + std::fabs(dof_axis[d][2] - want[2]) > 1e-8) { err = "unexpected hinge order in body " + bodies[i].name; return -1; }
    }
    std::vector<int32_t> dof_parent(nv, -1), dof_depth(nv, 0), dof_madr(nv + 1, 0), last(nb, -1);
    for (int d = 0; d < nv; d++) {
        const int b = dof_body[d];
        if (d > 0 && dof_body[d - 1] == b) dof_parent[d] = d - 1; else if (parent[b] >= 0) dof_parent[d] = last[parent[b]];
        last[b] = d;
    }
    for (int d = 0; d < nv; d++) dof_depth[d] = dof_parent[d] < 0 ? 0 : dof_depth[dof_parent[d]] + 1;
    for (int d = 0; d < nv; d++) dof_madr[d + 1] = dof_madr[d] + dof_depth[d] + 1;
    const int nM = dof_madr[nv];
    std::vector<int32_t> subtree(nb, 1), body_depth(nb, 0);
    for (int i = nb - 1; i > 0; i--) subtree[parent[i]] += subtree[i];
    for (int i = 1; i < nb; i++) body_depth[i] = body_depth[parent[i]] + 1;

    // ---- M(qpos0), invweight0 (model_compiler._mass_matrix_qpos0 and the block after it)
    std::vector<char> anc((size_t)nb * nb, 0);
    for (int b = 0; b < nb; b++) for (int k = b; k >= 0; k = parent[k]) anc[(size_t)b * nb + k] = 1;
    std::vector<double> M0((size_t)nv * nv, 0.0);
    std::vector<std::vector<double>> JV(nb), JW(nb);
    for (int b = 0; b < nb; b++) {
        std::vector<double>& Jv = JV[b]; std::vector<double>& Jw = JW[b];
        Jv.assign(3 * (size_t)nv, 0.0); Jw.assign(3 * (size_t)nv, 0.0);
        const V3d comg = {gpos[3 * b] + ipos[3 * b], gpos[3 * b + 1] + ipos[3 * b + 1], gpos[3 * b + 2] + ipos[3 * b + 2]};
        for (int d = 0; d < nv; d++) {
            if (!anc[(size_t)b * nb + dof_body[d]]) continue;
            const V3d ax = dof_axis[d];
            if (dof_trans[d]) for (int k = 0; k < 3; k++) Jv[(size_t)k * nv + d] = ax[k];
            else {
                const V3d r = {comg[0] - gpos[3 * dof_body[d]], comg[1] - gpos[3 * dof_body[d] + 1], comg[2] - gpos[3 * dof_body[d] + 2]};
                const V3d c = cross(ax, r);
                for (int k = 0; k < 3; k++) { Jw[(size_t)k * nv + d] = ax[k]; Jv[(size_t)k * nv + d] = c[k]; }
            }
        }
        for (int d1 = 0; d1 < nv; d1++) {
            double iw[3];
            for (int k = 0; k < 3; k++) iw[k] = inertia[b][3 * k] * Jw[d1] + inertia[b][3 * k + 1] * Jw[(size_t)nv + d1] + inertia[b][3 * k + 2] * Jw[2 * (size_t)nv + d1];
            for (int d2 = 0; d2 < nv; d2++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += mass[b] * Jv[(size_t)k * nv + d1] * Jv[(size_t)k * nv + d2] + Jw[(size_t)k * nv + d2] * iw[k];
                M0[(size_t)d1 * nv + d2] += s;
            }
        }
    }
    for (int d = 0; d < nv; d++) M0[(size_t)d * nv + d] += arm[d];
    std::vector<double> Minv;
    if (!invert(M0, nv, Minv)) { err = "singular M(qpos0)"; return -1; }
    std::vector<double> body_invw(2 * (size_t)nb), dof_invw(nv);
    for (int b = 0; b < nb; b++) {
        double tr[2] = {0, 0};
        for (int part = 0; part < 2; part++) {
            const std::vector<double>& J = part == 0 ? JV[b] : JW[b];
            for (int k = 0; k < 3; k++) {
                double s = 0;
                for (int d1 = 0; d1 < nv; d1++) { if (J[(size_t)k * nv + d1] == 0.0) continue; double t = 0; for (int d2 = 0; d2 < nv; d2++) t += Minv[(size_t)d1 * nv + d2] * J[(size_t)k * nv + d2]; s += J[(size_t)k * nv + d1] * t; }
                tr[part] += s;
            }
        }
        body_invw[2 * b] = tr[0] / 3.0; body_invw[2 * b + 1] = tr[1] / 3.0;
    }
    for (int d = 0; d < nv; d++) dof_invw[d] = Minv[(size_t)d * nv + d];
    { const double a = (dof_invw[0] + dof_invw[1] + dof_invw[2]) / 3.0, b = (dof_invw[3] + dof_invw[4] + dof_invw[5]) / 3.0; for (int k = 0; k < 3; k++) { dof_invw[k] = a; dof_invw[3 + k] = b; } }

    // ---- floor, controller gains
    std::vector<double> ff = floor->get("friction") ? floats(floor->gets("friction")) : (gd && gd->get("friction") ? floats(gd->gets("friction")) : std::vector<double>{geom_friction[0], geom_friction[1], geom_friction[2]});
    while (ff.size() < 3) ff.push_back(geom_friction[ff.size()]);
    const double geom_margin = gd && gd->get("margin") ? std::strtod(gd->gets("margin").c_str(), nullptr) : 0.0;
    double fric[3]; for (int k = 0; k < 3; k++) fric[k] = std::max(ff[k], geom_friction[k]);
    const int condim = std::max(std::atoi(attr_or(floor, gd, "condim", "3").c_str()), gd && gd->get("condim") ? std::atoi(gd->gets("condim").c_str()) : 3);
    std::vector<double> kp(nu, 0.0), kd(nu, 0.0), tlim(nu, 0.0), a_scale(nu, 1.0), diffw(nb, 1.0), uhc_b_diffw(nb, 1.0);
    double rfc_scale = 100.0, rfc_lim = 100.0;
    std::vector<double> base_rot = {0.7071, 0.7071, 0.0, 0.0};
    if (yml_path && yml_path[0]) {
        UhcCfg c;
        if (!parse_uhc_yml(yml_path, c, err)) return -1;
        if ((int)c.joint_names.size() != nu) { err = "uhc.yml joint_params: expected " + std::to_string(nu) + " rows"; return -1; }
        for (int i = 1, q = 0; i < nb; i++) for (const char* a : {"z", "y", "x"}) { if (c.joint_names[q] != bodies[i].name + "_" + a) { err = "uhc.yml joint order differs from the XML dof order"; return -1; } q++; }
        for (int j = 0; j < nu; j++) { kp[j] = c.joint_rows[j][0]; kd[j] = c.joint_rows[j][1]; a_scale[j] = c.joint_rows[j][3]; tlim[j] = c.joint_rows[j][4]; }
        rfc_scale = c.rfc_scale; rfc_lim = c.rfc_lim; base_rot = c.base_rot;
        if (!c.body_names.empty()) {
            if ((int)c.body_names.size() != nb - 1) { err = "uhc.yml body_params: expected " + std::to_string(nb - 1) + " rows"; return -1; }
            for (int i = 1; i < nb; i++) { if (c.body_names[i - 1] != bodies[i].name) { err = "uhc.yml body order differs from the XML body order"; return -1; } uhc_b_diffw[i] = c.body_w[i - 1]; }
        }
    }
    if (base_rot.size() != 4) { err = "base_rot needs four numbers"; return -1; }

    // ---- free objects
    std::vector<double> obj_geoms;
    const int nobj = (int)objects.size();
    for (int oi = 0; oi < nobj; oi++) for (const Xml* g : objects[oi].geoms) {
        const std::string type = attr_or(g, gd, "type", "sphere");
        const int typ = type == "box" ? 0 : type == "cylinder" ? 1 : -1;
        if (typ < 0) { err = "object geom type " + type + " not supported (box / cylinder)"; return -1; }
        std::vector<double> size = floats(g->gets("size")); size.push_back(0.0); while (size.size() < 3) size.push_back(0.0);
        const std::vector<double> pos = floats(g->gets("pos", "0 0 0")), eul = floats(g->gets("euler", "0 0 0"));
        if (pos.size() != 3 || eul.size() != 3 || !g->get("mass")) { err = "object geom needs pos / euler (3 numbers) and mass"; return -1; }
        const M3d R = euler_deg_to_mat(eul);
        obj_geoms.push_back((double)oi); obj_geoms.push_back((double)typ);
        for (int k = 0; k < 3; k++) obj_geoms.push_back(size[k]);
        for (int k = 0; k < 3; k++) obj_geoms.push_back(pos[k]);
        for (int k = 0; k < 9; k++) obj_geoms.push_back(R[k]);
        obj_geoms.push_back(std::strtod(g->gets("mass").c_str(), nullptr));
    }
    const int ngeom = (int)(obj_geoms.size() / 18);
    std::vector<int32_t> obj_geom_adr(nobj + 1, 0);
    std::vector<double> obj_mass(nobj, 0.0), obj_inertial(13 * (size_t)nobj, 0.0);
    for (int g = 0; g < ngeom; g++) { const int oi = (int)obj_geoms[18 * g]; for (int k = oi + 1; k <= nobj; k++) obj_geom_adr[k] += 1; obj_mass[oi] += obj_geoms[18 * g + 17]; }
    const double obj_arm = jd && jd->get("armature") ? std::strtod(jd->gets("armature").c_str(), nullptr) : 0.0;
    double obj_trace = 0.0;
    for (int oi = 0; oi < nobj; oi++) {
        double mo = 0; V3d com = {0, 0, 0};
        for (int g = 0; g < ngeom; g++) if ((int)obj_geoms[18 * g] == oi) mo += obj_geoms[18 * g + 17];
        for (int g = 0; g < ngeom; g++) if ((int)obj_geoms[18 * g] == oi) for (int k = 0; k < 3; k++) com[k] += obj_geoms[18 * g + 17] * obj_geoms[18 * g + 5 + k];
        for (int k = 0; k < 3; k++) com[k] /= mo;
        double Io[9] = {0};
        for (int g = 0; g < ngeom; g++) if ((int)obj_geoms[18 * g] == oi) {
            const double* G = &obj_geoms[18 * g];
            const double mg = G[17], *sz = G + 2, *Rg = G + 8;
            double Il[3];
            if ((int)G[1] == 0) { Il[0] = mg / 3.0 * (sz[1] * sz[1] + sz[2] * sz[2]); Il[1] = mg / 3.0 * (sz[0] * sz[0] + sz[2] * sz[2]); Il[2] = mg / 3.0 * (sz[0] * sz[0] + sz[1] * sz[1]); }
            else { const double ixx = mg * (3.0 * sz[0] * sz[0] + (2.0 * sz[1]) * (2.0 * sz[1])) / 12.0; Il[0] = ixx; Il[1] = ixx; Il[2] = 0.5 * mg * sz[0] * sz[0]; }
            const V3d dd = {G[5] - com[0], G[6] - com[1], G[7] - com[2]};
            const double d2 = dot(dd, dd);
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += Rg[3 * i + k] * Il[k] * Rg[3 * j + k];
                Io[3 * i + j] += s + mg * ((i == j ? d2 : 0.0) - dd[i] * dd[j]);
            }
        }
        // generalized mass matrix of the free joint at the identity pose: dofs = [lin (world); ang (body axes, about the body origin)]
        const double rx[9] = {0, -com[2], com[1], com[2], 0, -com[0], -com[1], com[0], 0};
        double Jv[18], Jw[18];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Jv[6 * i + j] = i == j ? 1.0 : 0.0; Jv[6 * i + 3 + j] = -rx[3 * i + j]; Jw[6 * i + j] = 0.0; Jw[6 * i + 3 + j] = i == j ? 1.0 : 0.0; }
        std::vector<double> Mo(36, 0.0), Moi;
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += mo * Jv[6 * k + a] * Jv[6 * k + b];
            for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) s += Jw[6 * k + a] * Io[3 * k + l] * Jw[6 * l + b];
            Mo[6 * a + b] = s + (a == b ? obj_arm : 0.0);
        }
        if (!invert(Mo, 6, Moi)) { err = "singular object inertia"; return -1; }
        double trv = 0, trw = 0;
        for (int k = 0; k < 3; k++) for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { trv += Jv[6 * k + a] * Moi[6 * a + b] * Jv[6 * k + b]; trw += Jw[6 * k + a] * Moi[6 * a + b] * Jw[6 * k + b]; }
        double* o = &obj_inertial[13 * (size_t)oi];
        o[0] = mo; o[1] = com[0]; o[2] = com[1]; o[3] = com[2]; o[4] = Io[0]; o[5] = Io[4]; o[6] = Io[8]; o[7] = Io[1]; o[8] = Io[2]; o[9] = Io[5];
        o[10] = trv / 3.0; o[11] = trw / 3.0; o[12] = obj_arm;
        for (int a = 0; a < 6; a++) obj_trace += Mo[6 * a + a];
    }
    const int nv_full = nv + 6 * nobj;
    double trM = 0; for (int d = 0; d < nv; d++) trM += M0[(size_t)d * nv + d];
    const double meaninertia = (trM + obj_trace) / (double)nv_full;

    // ---- the blob, fields in model_compiler.compile_model's order
    Blob B;
    B.addi("dims", {nb, nv, nv + 1, nu, nM, (int32_t)(verts.size() / 3), nobj, ngeom, condim});
    B.addi("body_parent", parent); B.addi("body_depth", body_depth); B.addi("body_subtree", subtree);
    B.addf("body_pos", body_pos); B.addf("body_ipos", ipos); B.addf("body_mass", mass);
    { std::vector<double> bi; for (int i = 0; i < nb; i++) for (int k : {0, 4, 8, 1, 2, 5}) bi.push_back(inertia[i][k]); B.addf("body_inertia", bi); }
    B.addf("body_gpos0", gpos); B.addf("body_rbound", rbound); B.addf("mesh_rbound", mesh_rbound); B.addf("body_diffw", diffw); B.addf("uhc_b_diffw", uhc_b_diffw);
    B.addf("planemesh", {maxplanemesh, tolplanemesh});
    B.addf("body_invweight0", body_invw); B.addf("dof_invweight0", dof_invw);
    B.addi("dof_body", dof_body); B.addi("dof_parent", dof_parent); B.addi("dof_depth", dof_depth); B.addi("dof_madr", dof_madr);
    B.addf("dof_armature", arm); B.addf("jnt_range", jrange); B.addi("jnt_limited", jlimited);
    B.addi("vert_adr", vert_adr); B.addf("verts", verts); B.addi("vert_nbr_adr", nbr_adr); B.addi("vert_nbr", nbr);
    B.addf("kp", kp); B.addf("kd", kd); B.addf("torque_lim", tlim); B.addf("a_scale", a_scale);
    B.addf("opt", {timestep, gravity[0], gravity[1], gravity[2], solref[0], solref[1], solimp[0], solimp[1], solimp[2], solimp[3], solimp[4], fric[0], fric[1], fric[2],
                   geom_margin, impratio, meaninertia, rfc_scale, rfc_lim, base_rot[0], base_rot[1], base_rot[2], base_rot[3], solver_iterations, solver_tolerance, (double)nv_full});
    B.addf("obj_geoms", obj_geoms); B.addi("obj_geom_adr", obj_geom_adr); B.addf("obj_mass", obj_mass); B.addf("obj_inertial", obj_inertial);
    B.addf("M0", M0);
    if (!B.write(out_path, 7u)) { err = "cannot write " + out_path; return -1; }
    return 0;
}

}  // namespace kpc
