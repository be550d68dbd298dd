// kp_collide.hpp -- narrow phases of the control-step kernel, wavefront versions (gfx950, wave64).
//
// What MuJoCo 2.1.0 runs for the geom pairs of the reference's scenes (sim.step(), uhc/envs/humanoid_im.py:527) [MJ-ext]; the fp64
// statement of each routine, with the recalled MuJoCo / libccd source it follows, is oracle/kp_collide.h:
//   floor - hull mesh          mjc_PlaneConvex     plane_mesh()      lane = hull vertex: wave arg-min + walk of the hull graph
//   floor - box / cylinder     mjc_PlaneBox / mjc_PlaneCylinder      plane_box() lane = corner, plane_cylinder() uniform
//   box / cylinder - hull mesh mjc_Convex (libccd MPR, one contact)  mpr<GeomSupport, HullSupport>: the portal refinement is
//                              wave-uniform scalar work, every support query of the hull is a 64-lane arg-max over its vertices
//   cylinder - box / cylinder  mjc_Convex                            mpr<GeomSupport, GeomSupport>
//   box - box                  mjc_BoxBox (SAT + face clipping)      box_box(): one lane, polygon scratch in LDS
// fp32 throughout: libccd's CCD_EPS becomes FLT_EPSILON, mpr_tolerance (1e-6) and mpr_iterations (50) are MuJoCo's defaults.
#pragma once
#include "kp_device.hpp"

namespace kp {

// The MPR query runs in fp64 on the fp32 poses.  libccd's portal refinement is a branching iteration whose OUTPUT depends on the
// path taken (the final portal triangle is any three vertices of the Minkowski-difference face the origin ray leaves through, and on
// curved shapes the refinement stops at a 1e-6 tolerance): in fp32 the branch decisions (signs of near-zero triple products) flip
// against the fp64 statement in ~8 % of random interpenetrating scenes and the contact normal then differs by 1e-2.  In fp64 the two
// follow the same path unless the configuration itself is degenerate.  The cost is confined to hull - object pairs.
constexpr double C_EPS = 2.220446049250313e-16;
constexpr double C_MPR_TOL = 1e-6;
constexpr int C_MPR_ITER = 50;

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 d3(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 d3(V3 v) { return D3{(double)v.x, (double)v.y, (double)v.z}; }
__device__ __forceinline__ V3 f3(D3 v) { return V3{(float)v.x, (float)v.y, (float)v.z}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return D3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__device__ __forceinline__ bool c_is_zero(double x) { return fabs(x) < C_EPS; }
__device__ __forceinline__ bool c_eq(double a, double b) {
    const double ab = fabs(a - b);
    if (ab < C_EPS) return true;
    a = fabs(a); b = fabs(b);
    return b > a ? ab < C_EPS * b : ab < C_EPS * a;
}
__device__ __forceinline__ D3 c_normalize(D3 a) { const double inv = 1.0 / sqrt(dot(a, a)); return d3(a.x * inv, a.y * inv, a.z * inv); }   // one fp64 division instead of three (an fp64 division is ~15 instructions); within 1 ulp of a / |a|
__device__ __forceinline__ V3 c_normalize(V3 a) { const float n = sqrtf(dot(a, a)); return v3(a.x / n, a.y / n, a.z / n); }
__device__ __forceinline__ D3 mulmat_t(const double* m, D3 v) {      // R^T v
    return D3{m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z};
}
__device__ __forceinline__ D3 mulmat(const double* m, D3 v) {
    return D3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
__device__ __forceinline__ float bcast_lane(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
#define KP_DPP64(v, CTRL) ([&]() { const unsigned long long u_ = __builtin_bit_cast(unsigned long long, (v)); \
    const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u_, CTRL, 0xF, 0xF, true), hi_ = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u_ >> 32), CTRL, 0xF, 0xF, true); \
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi_ << 32) | (unsigned long long)(unsigned)lo_); }())
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// wave-wide fp64 maximum without LDS traffic: xor butterflies inside each 16-lane row with DPP, the four row maxima meet through v_readlane
__device__ __forceinline__ double wave_max_d(double v) {
    v = fmax(v, KP_DPP64(v, 0xB1)); v = fmax(v, KP_DPP64(v, 0x4E)); v = fmax(v, KP_DPP64(v, 0x141)); v = fmax(v, KP_DPP64(v, 0x140));
    return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
}

// ---- support functions (mjccd_support): farthest point along the unit world direction dir, inflated by margin along dir.
// Both functors read their shape from LDS on every query (wave-uniform addresses: broadcast reads) instead of carrying it in registers:
// the MPR program is wave-uniform, so everything it keeps live costs a VGPR per value on all 64 lanes, and the register file of the object
// kernel is what its articulated-body loops need (tools/micro/spill_report.py).
struct GeomSupport {                 // box (type 0) / z-axis cylinder (type 1); record (LDS) = type, size[3], pos[3], mat[9]
    const float* g;
    __device__ __forceinline__ explicit GeomSupport(const float* g_) : g(g_) {}
    __device__ __forceinline__ D3 center() const { return d3((double)g[4], (double)g[5], (double)g[6]); }
    __device__ __forceinline__ D3 operator()(D3 dir, double margin) const {
        double R[9];
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = (double)g[7 + k];
        const D3 l = mulmat_t(R, dir);
        const double s0 = (double)g[1], s1 = (double)g[2], s2 = (double)g[3];
        D3 p;
        if (g[0] == 0.f) p = d3(l.x > 0.0 ? s0 : -s0, l.y > 0.0 ? s1 : -s1, l.z > 0.0 ? s2 : -s2);
        else {
            const double n = sqrt(l.x * l.x + l.y * l.y);
            p = n > 1e-15 ? d3(l.x / n * s0, l.y / n * s0, 0.0) : d3(0.0, 0.0, 0.0);
            p.z = l.z > 0.0 ? s1 : (l.z < 0.0 ? -s1 : 0.0);
        }
        return center() + mulmat(R, p) + margin * dir;
    }
};
struct HullSupport {                 // lane v < nv holds body-frame vertex v of the hull (fp32 model data); exhaustive arg-max in fp64, first maximum
    const float* h;                  // LDS: xb[3], com[3], R[9] of the hull (written by hull_support_store)
    V3 vert; bool has;
    __device__ __forceinline__ HullSupport(const float* h_, V3 vert_, bool has_) : h(h_), vert(vert_), has(has_) {}
    __device__ __forceinline__ D3 center() const { return d3((double)h[3], (double)h[4], (double)h[5]); }
    __device__ __forceinline__ D3 operator()(D3 dir, double margin) const {
        double R[9];
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = (double)h[6 + k];
        const D3 l = mulmat_t(R, dir);
        const double d = has ? dot(l, d3(vert)) : -1.0e300;
        const double dmax = wave_max_d(d);
        const int idx = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(d == dmax)) - 1);
        const D3 p = d3((double)bcast_lane(vert.x, idx), (double)bcast_lane(vert.y, idx), (double)bcast_lane(vert.z, idx));
        return d3((double)h[0], (double)h[1], (double)h[2]) + mulmat(R, p) + margin * dir;
    }
};
// every lane stores the same (wave-uniform) values: no exec-mask change, one LDS write per word
__device__ __forceinline__ void hull_support_store(float* h, V3 xb, V3 com, const float* R) {
    st3(h, xb); st3(h + 3, com);
#pragma unroll
    for (int k = 0; k < 9; k++) h[6 + k] = R[k];
}

// ---- libccd MPR (ccdMPRPenetration).  Every lane runs the same scalar program; the two support functors may use wave collectives.
// A portal vertex is a point v = v1 - v2 of the Minkowski difference with its two witnesses.  Only v takes part in the iteration; the
// witnesses are read once, by findPos, so they live in LDS: slot k (0 = v0 ... 3 = v3, 4 = the candidate v4) = 6 doubles at pm + 6 k.
template <class SA, class SB>
__device__ __forceinline__ D3 mink(const SA& a, const SB& b, D3 dir, double margin, double* pm, int slot) {
    const D3 v1 = a(dir, margin), v2 = b(d3(-dir.x, -dir.y, -dir.z), margin);
    double* m = pm + 6 * slot;
    m[0] = v1.x; m[1] = v1.y; m[2] = v1.z; m[3] = v2.x; m[4] = v2.y; m[5] = v2.z;
    return v1 - v2;
}
__device__ __forceinline__ void slot_copy(double* pm, int dst, int src) {
#pragma unroll
    for (int k = 0; k < 6; k++) pm[6 * dst + k] = pm[6 * src + k];
}
__device__ __forceinline__ D3 slot_v1(const double* pm, int k) { return d3(pm[6 * k], pm[6 * k + 1], pm[6 * k + 2]); }
__device__ __forceinline__ D3 slot_v2(const double* pm, int k) { return d3(pm[6 * k + 3], pm[6 * k + 4], pm[6 * k + 5]); }
__device__ __forceinline__ D3 portal_dir(D3 p1, D3 p2, D3 p3) { return c_normalize(cross(p2 - p1, p3 - p1)); }
__device__ __forceinline__ bool reach_tol(D3 p1, D3 p2, D3 p3, D3 v4, D3 dir) {
    const double dv4 = dot(v4, dir);
    const double m = fmin(dv4 - dot(p1, dir), fmin(dv4 - dot(p2, dir), dv4 - dot(p3, dir)));
    return c_eq(m, C_MPR_TOL) || m < C_MPR_TOL;
}
// replaces one portal vertex by v4 (slot 4); returns nothing: p1..p3 and the witness slots are updated together
__device__ __forceinline__ void expand_portal(D3 p0, D3& p1, D3& p2, D3& p3, D3 v4, double* pm) {
    const D3 v4v0 = cross(v4, p0);
    int k;
    if (dot(p1, v4v0) > 0.0) k = dot(p2, v4v0) > 0.0 ? 1 : 3;
    else k = dot(p3, v4v0) > 0.0 ? 2 : 1;
    if (k == 1) p1 = v4; else if (k == 2) p2 = v4; else p3 = v4;
    slot_copy(pm, k, 4);
}
__device__ __forceinline__ double pt_seg_dist2(D3 x0, D3 b, D3& w) {          // closest point of segment x0-b to the origin
    const D3 d = b - x0;
    const double t = -dot(x0, d) / dot(d, d);
    if (t < 0.0 || c_is_zero(t)) w = x0;
    else if (t > 1.0 || c_eq(t, 1.0)) w = b;
    else w = x0 + t * d;
    return dot(w, w);
}
__device__ __forceinline__ double pt_tri_dist2(D3 x0, D3 B, D3 C, D3& w) {    // closest point of triangle x0 B C to the origin
    const D3 d1 = B - x0, d2 = C - x0;
    const double v = dot(d1, d1), ww = dot(d2, d2), p = dot(x0, d1), q = dot(x0, d2), r = dot(d1, d2);
    const double s = (q * r - ww * p) / (ww * v - r * r), t = (-s * r - q) / ww;
    if ((c_is_zero(s) || s > 0.0) && (c_eq(s, 1.0) || s < 1.0) && (c_is_zero(t) || t > 0.0) && (c_eq(t, 1.0) || t < 1.0) && (c_eq(t + s, 1.0) || t + s < 1.0)) {
        w = x0 + s * d1 + t * d2;
        return dot(w, w);
    }
    D3 w2;
    double dist = pt_seg_dist2(x0, B, w), d;
    d = pt_seg_dist2(x0, C, w2); if (d < dist) { dist = d; w = w2; }
    d = pt_seg_dist2(B, C, w2); if (d < dist) { dist = d; w = w2; }
    return dist;
}
// 0 = the inflated shapes intersect: depth, dir (from shape A towards shape B), pos.  -1 = no intersection.
// pm: 30 doubles of LDS scratch (8-byte aligned) for the witnesses of the portal vertices.
template <class SA, class SB>
__device__ __forceinline__ int mpr(const SA& A, const SB& B, double margin, double& depth, D3& dir_out, D3& pos, double* pm) {
    D3 p0, p1, p2, p3, v4;
    {
        const D3 c1 = A.center(), c2 = B.center();
        pm[0] = c1.x; pm[1] = c1.y; pm[2] = c1.z; pm[3] = c2.x; pm[4] = c2.y; pm[5] = c2.z;
        p0 = c1 - c2;
    }
    if (c_eq(p0.x, 0.0) && c_eq(p0.y, 0.0) && c_eq(p0.z, 0.0)) p0.x += C_EPS * 10.0;
    D3 dir = c_normalize(d3(-p0.x, -p0.y, -p0.z));
    p1 = mink(A, B, dir, margin, pm, 1);
    double dt = dot(p1, dir);
    if (c_is_zero(dt) || dt < 0.0) return -1;
    dir = cross(p0, p1);
    if (c_is_zero(dot(dir, dir))) {
        pos = 0.5 * (slot_v1(pm, 1) + slot_v2(pm, 1));
        if (c_eq(p1.x, 0.0) && c_eq(p1.y, 0.0) && c_eq(p1.z, 0.0)) { depth = 0.0; dir_out = d3(0.0, 0.0, 0.0); return 0; }
        depth = sqrt(dot(p1, p1)); dir_out = c_normalize(p1);
        return 0;
    }
    dir = c_normalize(dir);
    p2 = mink(A, B, dir, margin, pm, 2);
    dt = dot(p2, dir);
    if (c_is_zero(dt) || dt < 0.0) return -1;
    dir = c_normalize(cross(p1 - p0, p2 - p0));
    if (dot(dir, p0) > 0.0) {
        const D3 t = p1; p1 = p2; p2 = t; dir = d3(-dir.x, -dir.y, -dir.z);
        slot_copy(pm, 4, 1); slot_copy(pm, 1, 2); slot_copy(pm, 2, 4);
    }
    for (int guard = 0; guard < 64; guard++) {
        p3 = mink(A, B, dir, margin, pm, 3);
        dt = dot(p3, dir);
        if (c_is_zero(dt) || dt < 0.0) return -1;
        bool cont = false;
        dt = dot(cross(p1, p3), p0);
        if (dt < 0.0 && !c_is_zero(dt)) { p2 = p3; slot_copy(pm, 2, 3); cont = true; }
        if (!cont) { dt = dot(cross(p3, p2), p0); if (dt < 0.0 && !c_is_zero(dt)) { p1 = p3; slot_copy(pm, 1, 3); cont = true; } }
        if (!cont) break;
        dir = c_normalize(cross(p1 - p0, p2 - p0));
    }
    for (int guard = 0; guard < 64; guard++) {                           // refinePortal
        dir = portal_dir(p1, p2, p3);
        dt = dot(dir, p1);
        if (c_is_zero(dt) || dt > 0.0) break;
        v4 = mink(A, B, dir, margin, pm, 4);
        dt = dot(v4, dir);
        if (!(c_is_zero(dt) || dt > 0.0) || reach_tol(p1, p2, p3, v4, dir)) return -1;
        expand_portal(p0, p1, p2, p3, v4, pm);
    }
    for (int it = 0;; it++) {                                            // findPenetr
        dir = portal_dir(p1, p2, p3);
        v4 = mink(A, B, dir, margin, pm, 4);
        if (reach_tol(p1, p2, p3, v4, dir) || it > C_MPR_ITER) {
            D3 w;
            depth = sqrt(pt_tri_dist2(p1, p2, p3, w));
            if (c_is_zero(depth)) w = dir;
            dir_out = c_normalize(w);
            // findPos: barycentric coordinates of the origin in the tetrahedron (v0, v1, v2, v3)
            double b0 = dot(cross(p1, p2), p3), b1 = dot(cross(p3, p2), p0), b2 = dot(cross(p0, p1), p3), b3 = dot(cross(p2, p1), p0);
            double sum = b0 + b1 + b2 + b3;
            if (c_is_zero(sum) || sum < 0.0) {
                b0 = 0.0; b1 = dot(cross(p2, p3), dir); b2 = dot(cross(p3, p1), dir); b3 = dot(cross(p1, p2), dir);
                sum = b1 + b2 + b3;
            }
            const D3 q1 = b0 * slot_v1(pm, 0) + b1 * slot_v1(pm, 1) + b2 * slot_v1(pm, 2) + b3 * slot_v1(pm, 3);
            const D3 q2 = b0 * slot_v2(pm, 0) + b1 * slot_v2(pm, 1) + b2 * slot_v2(pm, 2) + b3 * slot_v2(pm, 3);
            pos = (0.5 / sum) * (q1 + q2);
            return 0;
        }
        expand_portal(p0, p1, p2, p3, v4, pm);
    }
}

struct Contact { float dist; V3 pos, n; };

// mjc_Convex: shapes inflated by margin / 2, dist = margin - depth, normal = libccd's direction (geom 1 -> geom 2)
template <class SA, class SB>
__device__ __forceinline__ int convex_pair(const SA& g1, const SB& g2, float margin, Contact& c, double* pm) {
    double depth; D3 dir, pos;
    if (mpr(g1, g2, 0.5 * (double)margin, depth, dir, pos, pm) != 0) return 0;
    if (dir.x == 0.0 && dir.y == 0.0 && dir.z == 0.0) return 0;
    c.dist = (float)((double)margin - depth); c.n = f3(dir); c.pos = f3(pos);
    return 1;
}

// contact record in LDS scratch: dist, pos[3], normal[3]
__device__ __forceinline__ void put_rec(float* rec, int i, float dist, V3 pos, V3 n) { rec[7 * i] = dist; st3(rec + 7 * i + 1, pos); st3(rec + 7 * i + 4, n); }

// mjc_PlaneCylinder against the floor z = 0 (uniform scalar work; the lanes with writer = true store the records).  g = geom record.
// Up to 4 contacts.
__device__ __forceinline__ int plane_cylinder(const float* g, float margin, float* rec, bool writer) {
    const float* R = g + 7;
    const V3 normal = v3(0.f, 0.f, 1.f), pos = ld3(g + 4);
    V3 axis = v3(R[2], R[5], R[8]);
    float prjaxis = axis.z;
    if (prjaxis > 0.f) { axis = v3(-axis.x, -axis.y, -axis.z); prjaxis = -prjaxis; }
    const float dist0 = pos.z;
    // -normal with its axial component removed, (n . a) a - n.  Its z component a_z^2 - 1 is a difference of two numbers next to 1 for an
    // upright cylinder (tilt 1e-3 rad: 1e-6 known to 1e-7 in fp32, i.e. the rim height r sin(tilt) off by tens of per cent); for a unit axis
    // it equals -(a_x^2 + a_y^2), which has no cancellation
    V3 vec = v3(prjaxis * axis.x, prjaxis * axis.y, -(axis.x * axis.x + axis.y * axis.y));
    const float len2 = dot(vec, vec);
    if (len2 >= 1e-12f) vec = (g[1] / sqrtf(len2)) * vec;
    else vec = v3(R[0] * g[1], R[3] * g[1], R[6] * g[1]);
    const float prjvec = vec.z;
    axis = g[2] * axis; prjaxis *= g[2];
    if (dist0 + prjaxis + prjvec > margin) return 0;
    int cnt = 0;
    { const float d = dist0 + prjaxis + prjvec; if (writer) put_rec(rec, cnt, d, pos + vec + axis - (0.5f * d) * normal, normal); cnt++; }
    if (dist0 - prjaxis + prjvec <= margin) { const float d = dist0 - prjaxis + prjvec; if (writer) put_rec(rec, cnt, d, pos + vec - axis - (0.5f * d) * normal, normal); cnt++; }
    const float prjvec1 = -0.5f * prjvec;
    if (dist0 + prjaxis + prjvec1 <= margin) {
        const V3 vec1 = (g[1] * 0.8660254037844386f) * c_normalize(cross(vec, axis));
        const float d = dist0 + prjaxis + prjvec1;
        if (writer) { put_rec(rec, cnt, d, pos + vec1 + axis - 0.5f * vec - (0.5f * d) * normal, normal); put_rec(rec, cnt + 1, d, pos - vec1 + axis - 0.5f * vec - (0.5f * d) * normal, normal); }
        cnt += 2;
    }
    return cnt;
}

// box - box (see oracle/kp_collide.h kpo_box_box: SAT over the 15 axes, then face clipping or one edge-edge contact).  Runs on ONE
// lane; `scr` = 2 x 16 x 3 floats of LDS scratch for the polygon being clipped, `rec` = contact records.  Normal from box a to box b.
// Up to 8 contacts.
__device__ inline int box_box(const float* ga, const float* gb, float margin, float* rec, float* scr) {
    const V3 pa = ld3(ga + 4), pb = ld3(gb + 4), d = pb - pa;
    const float* Ma = ga + 7; const float* Mb = gb + 7;
    V3 Ra[3], Rb[3];
    for (int i = 0; i < 3; i++) { Ra[i] = v3(Ma[i], Ma[3 + i], Ma[6 + i]); Rb[i] = v3(Mb[i], Mb[3 + i], Mb[6 + i]); }
    const float sa[3] = {ga[1], ga[2], ga[3]}, sb[3] = {gb[1], gb[2], gb[3]};
    float best = -3.0e38f; int bt = -1, bi = 0, bj = 0; V3 bn = v3(0.f, 0.f, 0.f);
    for (int t = 0; t < 2; t++) for (int i = 0; i < 3; i++) {
        const V3 n = t == 0 ? Ra[i] : Rb[i];
        float ra = 0.f, rb = 0.f;
        for (int k = 0; k < 3; k++) { ra += sa[k] * fabsf(dot(Ra[k], n)); rb += sb[k] * fabsf(dot(Rb[k], n)); }
        const float s = fabsf(dot(d, n)) - ra - rb;
        if (s > margin) return 0;
        if (s > best) { best = s; bt = t; bi = i; bn = dot(d, n) < 0.f ? v3(-n.x, -n.y, -n.z) : n; }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        V3 n = cross(Ra[i], Rb[j]);
        const float l = sqrtf(dot(n, n));
        if (l < 1e-6f) continue;
        n = (1.0f / l) * n;
        float ra = 0.f, rb = 0.f;
        for (int k = 0; k < 3; k++) { ra += sa[k] * fabsf(dot(Ra[k], n)); rb += sb[k] * fabsf(dot(Rb[k], n)); }
        const float s = fabsf(dot(d, n)) - ra - rb;
        if (s > margin) return 0;
        if (s > best + 0.05f * fabsf(best) + 1e-9f) { best = s; bt = 2; bi = i; bj = j; bn = dot(d, n) < 0.f ? v3(-n.x, -n.y, -n.z) : n; }
    }
    if (bt == 2) {
        V3 qa = pa, qb = pb;
        for (int a = 0; a < 3; a++) if (a != bi) qa = qa + ((dot(bn, Ra[a]) > 0.f ? 1.f : -1.f) * sa[a]) * Ra[a];
        for (int a = 0; a < 3; a++) if (a != bj) qb = qb + ((dot(bn, Rb[a]) > 0.f ? -1.f : 1.f) * sb[a]) * Rb[a];
        const V3 ua = Ra[bi], ub = Rb[bj], w = qa - qb;
        const V3 uxu = cross(ua, ub);                                   // 1 - (ua . ub)^2 without the cancellation
        const float b = dot(ua, ub), dd = dot(ua, w), e = dot(ub, w), den = dot(uxu, uxu);
        float ta = den > 1e-12f ? (b * e - dd) / den : 0.f, tb = den > 1e-12f ? (e - b * dd) / den : 0.f;
        ta = fmaxf(-sa[bi], fminf(sa[bi], ta)); tb = fmaxf(-sb[bj], fminf(sb[bj], tb));
        put_rec(rec, 0, best, 0.5f * ((qa + ta * ua) + (qb + tb * ub)), bn);
        return 1;
    }
    const bool refA = bt == 0;
    const V3 rpos = refA ? pa : pb, ipos = refA ? pb : pa;
    const V3* Rr = refA ? Ra : Rb; const V3* Ri = refA ? Rb : Ra;
    const float* sr = refA ? sa : sb; const float* si = refA ? sb : sa;
    const V3 nr = refA ? bn : v3(-bn.x, -bn.y, -bn.z);
    int ia = 0; float mn = 3.0e38f;
    for (int a = 0; a < 3; a++) { const float c = -fabsf(dot(Ri[a], nr)); if (c < mn) { mn = c; ia = a; } }
    const float sgi = dot(Ri[ia], nr) > 0.f ? -1.f : 1.f;
    const int u = (ia + 1) % 3, v = (ia + 2) % 3;
    float* poly = scr; float* out = scr + 48;
    int np = 4;
    for (int c = 0; c < 4; c++) {
        const float su = (c == 0 || c == 3) ? -1.f : 1.f, sv = c < 2 ? -1.f : 1.f;
        st3(poly + 3 * c, ipos + (sgi * si[ia]) * Ri[ia] + (su * si[u]) * Ri[u] + (sv * si[v]) * Ri[v]);
    }
    const int ru = (bi + 1) % 3, rv = (bi + 2) % 3;
    for (int side = 0; side < 4; side++) {
        const V3 ax = side < 2 ? Rr[ru] : Rr[rv];
        const float sg = (side & 1) ? -1.f : 1.f, lim = side < 2 ? sr[ru] : sr[rv];
        int no = 0;
        for (int i = 0; i < np; i++) {
            const V3 p = ld3(poly + 3 * i), q = ld3(poly + 3 * ((i + 1) % np));
            const float dp = sg * dot(p - rpos, ax) - lim, dq = sg * dot(q - rpos, ax) - lim;
            if (dp <= 0.f && no < 16) { st3(out + 3 * no, p); no++; }
            if ((dp <= 0.f) != (dq <= 0.f) && no < 16) { const float t = dp / (dp - dq); st3(out + 3 * no, p + t * (q - p)); no++; }
        }
        np = no;
        float* tmp = poly; poly = out; out = tmp;
        if (np == 0) return 0;
    }
    int cnt = 0;
    for (int i = 0; i < np && cnt < 8; i++) {
        const V3 p = ld3(poly + 3 * i);
        const float dist = dot(p - rpos, nr) - sr[bi];
        if (dist > margin) continue;
        put_rec(rec, cnt, dist, p - (0.5f * dist) * nr, bn);
        cnt++;
    }
    return cnt;
}

}  // namespace kp
