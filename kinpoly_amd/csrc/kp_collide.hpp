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

// ---- support functions (mjccd_support): farthest point along the unit world direction dir, inflated by margin along dir
struct GeomSupport {                 // box (type 0) / z-axis cylinder (type 1); record = type, size[3], pos[3], mat[9]
    double type, size[3], pos[3], R[9];
    __device__ __forceinline__ explicit GeomSupport(const float* g) {
        type = g[0];
        for (int k = 0; k < 3; k++) { size[k] = g[1 + k]; pos[k] = g[4 + k]; }
        for (int k = 0; k < 9; k++) R[k] = g[7 + k];
    }
    __device__ __forceinline__ D3 center() const { return d3(pos[0], pos[1], pos[2]); }
    __device__ __forceinline__ D3 operator()(D3 dir, double margin) const {
        const D3 l = mulmat_t(R, dir);
        D3 p;
        if (type == 0.0) p = d3(l.x > 0.0 ? size[0] : -size[0], l.y > 0.0 ? size[1] : -size[1], l.z > 0.0 ? size[2] : -size[2]);
        else {
            const double n = sqrt(l.x * l.x + l.y * l.y);
            p = n > 1e-15 ? d3(l.x / n * size[0], l.y / n * size[0], 0.0) : d3(0.0, 0.0, 0.0);
            p.z = l.z > 0.0 ? size[1] : (l.z < 0.0 ? -size[1] : 0.0);
        }
        return center() + mulmat(R, p) + margin * dir;
    }
};
struct HullSupport {                 // lane v < nv holds body-frame vertex v of the hull (fp32 model data); exhaustive arg-max in fp64, first maximum
    D3 xb, com; V3 vert; double R[9]; bool has;
    __device__ __forceinline__ HullSupport(V3 xb_, const float* R_, V3 com_, V3 vert_, bool has_) : xb(d3(xb_)), com(d3(com_)), vert(vert_), has(has_) {
        for (int k = 0; k < 9; k++) R[k] = R_[k];
    }
    __device__ __forceinline__ D3 center() const { return com; }
    __device__ __forceinline__ D3 operator()(D3 dir, double margin) const {
        const D3 l = mulmat_t(R, dir);
        const double d = has ? dot(l, d3(vert)) : -1.0e300;
        const double dmax = wave_max_d(d);
        const int idx = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(d == dmax)) - 1);
        const D3 p = d3((double)bcast_lane(vert.x, idx), (double)bcast_lane(vert.y, idx), (double)bcast_lane(vert.z, idx));
        return xb + mulmat(R, p) + margin * dir;
    }
};

// ---- libccd MPR (ccdMPRPenetration).  Every lane runs the same scalar program; the two support functors may use wave collectives.
struct Sup { D3 v, v1, v2; };
template <class SA, class SB>
__device__ __forceinline__ Sup mink(const SA& a, const SB& b, D3 dir, double margin) {
    Sup s; s.v1 = a(dir, margin); s.v2 = b(d3(-dir.x, -dir.y, -dir.z), margin); s.v = s.v1 - s.v2; return s;
}
__device__ __forceinline__ D3 portal_dir(const Sup& p1, const Sup& p2, const Sup& p3) { return c_normalize(cross(p2.v - p1.v, p3.v - p1.v)); }
__device__ __forceinline__ bool reach_tol(const Sup& p1, const Sup& p2, const Sup& p3, const Sup& v4, D3 dir) {
    const double dv4 = dot(v4.v, dir);
    const double m = fmin(dv4 - dot(p1.v, dir), fmin(dv4 - dot(p2.v, dir), dv4 - dot(p3.v, dir)));
    return c_eq(m, C_MPR_TOL) || m < C_MPR_TOL;
}
__device__ __forceinline__ void expand_portal(const Sup& p0, Sup& p1, Sup& p2, Sup& p3, const Sup& v4) {
    const D3 v4v0 = cross(v4.v, p0.v);
    if (dot(p1.v, v4v0) > 0.0) { if (dot(p2.v, v4v0) > 0.0) p1 = v4; else p3 = v4; }
    else { if (dot(p3.v, v4v0) > 0.0) p2 = v4; else p1 = v4; }
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
template <class SA, class SB>
__device__ __forceinline__ int mpr(const SA& A, const SB& B, double margin, double& depth, D3& dir_out, D3& pos) {
    Sup p0, p1, p2, p3, v4;
    p0.v1 = A.center(); p0.v2 = B.center(); p0.v = p0.v1 - p0.v2;
    if (c_eq(p0.v.x, 0.0) && c_eq(p0.v.y, 0.0) && c_eq(p0.v.z, 0.0)) p0.v.x += C_EPS * 10.0;
    D3 dir = c_normalize(d3(-p0.v.x, -p0.v.y, -p0.v.z));
    p1 = mink(A, B, dir, margin);
    double dt = dot(p1.v, dir);
    if (c_is_zero(dt) || dt < 0.0) return -1;
    dir = cross(p0.v, p1.v);
    if (c_is_zero(dot(dir, dir))) {
        pos = 0.5 * (p1.v1 + p1.v2);
        if (c_eq(p1.v.x, 0.0) && c_eq(p1.v.y, 0.0) && c_eq(p1.v.z, 0.0)) { depth = 0.0; dir_out = d3(0.0, 0.0, 0.0); return 0; }
        depth = sqrt(dot(p1.v, p1.v)); dir_out = c_normalize(p1.v);
        return 0;
    }
    dir = c_normalize(dir);
    p2 = mink(A, B, dir, margin);
    dt = dot(p2.v, dir);
    if (c_is_zero(dt) || dt < 0.0) return -1;
    dir = c_normalize(cross(p1.v - p0.v, p2.v - p0.v));
    if (dot(dir, p0.v) > 0.0) { const Sup t = p1; p1 = p2; p2 = t; dir = d3(-dir.x, -dir.y, -dir.z); }
    for (int guard = 0; guard < 64; guard++) {
        p3 = mink(A, B, dir, margin);
        dt = dot(p3.v, dir);
        if (c_is_zero(dt) || dt < 0.0) return -1;
        bool cont = false;
        dt = dot(cross(p1.v, p3.v), p0.v);
        if (dt < 0.0 && !c_is_zero(dt)) { p2 = p3; cont = true; }
        if (!cont) { dt = dot(cross(p3.v, p2.v), p0.v); if (dt < 0.0 && !c_is_zero(dt)) { p1 = p3; cont = true; } }
        if (!cont) break;
        dir = c_normalize(cross(p1.v - p0.v, p2.v - p0.v));
    }
    for (int guard = 0; guard < 64; guard++) {                           // refinePortal
        dir = portal_dir(p1, p2, p3);
        dt = dot(dir, p1.v);
        if (c_is_zero(dt) || dt > 0.0) break;
        v4 = mink(A, B, dir, margin);
        dt = dot(v4.v, dir);
        if (!(c_is_zero(dt) || dt > 0.0) || reach_tol(p1, p2, p3, v4, dir)) return -1;
        expand_portal(p0, p1, p2, p3, v4);
    }
    for (int it = 0;; it++) {                                            // findPenetr
        dir = portal_dir(p1, p2, p3);
        v4 = mink(A, B, dir, margin);
        if (reach_tol(p1, p2, p3, v4, dir) || it > C_MPR_ITER) {
            D3 w;
            depth = sqrt(pt_tri_dist2(p1.v, p2.v, p3.v, w));
            if (c_is_zero(depth)) w = dir;
            dir_out = c_normalize(w);
            // findPos: barycentric coordinates of the origin in the tetrahedron (v0, v1, v2, v3)
            double b0 = dot(cross(p1.v, p2.v), p3.v), b1 = dot(cross(p3.v, p2.v), p0.v), b2 = dot(cross(p0.v, p1.v), p3.v), b3 = dot(cross(p2.v, p1.v), p0.v);
            double sum = b0 + b1 + b2 + b3;
            if (c_is_zero(sum) || sum < 0.0) {
                b0 = 0.0; b1 = dot(cross(p2.v, p3.v), dir); b2 = dot(cross(p3.v, p1.v), dir); b3 = dot(cross(p1.v, p2.v), dir);
                sum = b1 + b2 + b3;
            }
            const D3 q1 = b0 * p0.v1 + b1 * p1.v1 + b2 * p2.v1 + b3 * p3.v1, q2 = b0 * p0.v2 + b1 * p1.v2 + b2 * p2.v2 + b3 * p3.v2;
            pos = (0.5 / sum) * (q1 + q2);
            return 0;
        }
        expand_portal(p0, p1, p2, p3, v4);
    }
}

struct Contact { float dist; V3 pos, n; };

// mjc_Convex: shapes inflated by margin / 2, dist = margin - depth, normal = libccd's direction (geom 1 -> geom 2)
template <class SA, class SB>
__device__ __forceinline__ int convex_pair(const SA& g1, const SB& g2, float margin, Contact& c) {
    double depth; D3 dir, pos;
    if (mpr(g1, g2, 0.5 * (double)margin, depth, dir, pos) != 0) return 0;
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
    V3 vec = prjaxis * axis - normal;
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
        const float b = dot(ua, ub), dd = dot(ua, w), e = dot(ub, w), den = 1.f - b * b;
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
