/* kp_collide.h -- narrow phases of the oracle: fp64 restatements of the MuJoCo 2.1.0 collision functions the reference's scenes
 * reach through sim.step() (uhc/envs/humanoid_im.py:527).  TEST INFRASTRUCTURE ONLY (see kp_oracle.c).
 *
 * Geom pairs of assets/mujoco_models/humanoid_smpl_neutral_mesh_all(_step).xml and the MuJoCo routine each one dispatches to
 * (mjCOLLISIONFUNC table, engine_collision_driver.c; geom types ordered plane < cylinder < box < mesh, g1 = the lower type) [MJ-ext]:
 *
 *   floor  - hull mesh        mjc_PlaneConvex   engine_collision_convex.c      kpo_plane_mesh
 *   floor  - box              mjc_PlaneBox      engine_collision_primitive.c   kpo_plane_box
 *   floor  - cylinder         mjc_PlaneCylinder engine_collision_primitive.c   kpo_plane_cylinder
 *   box    - box              mjc_BoxBox        engine_collision_box.c         kpo_box_box        (see the note there)
 *   cylinder/box - hull mesh  mjc_Convex        engine_collision_convex.c      kpo_convex  (libccd MPR, ONE contact per pair)
 *   cylinder - box, cylinder - cylinder         mjc_Convex                     kpo_convex
 *
 * All of it is recalled from the MuJoCo / libccd sources ("MuJoCo: Computation: Collision detection"; libccd src/mpr.c,
 * src/vec3.c) and cannot be executed against MuJoCo here: PARITY UNPINNED.  What is knowingly not reproducible is listed in
 * DESIGN.md section 2 (vertex numbering of MuJoCo's un-welded STL import, its hill-climbing support search on exact ties).
 */
#ifndef KP_COLLIDE_H
#define KP_COLLIDE_H

#define KPC_EPS 2.220446049250313e-16   /* CCD_EPS in a double build of libccd */
#define KPC_MPR_TOL 1e-6                /* mjOption.mpr_tolerance default */
#define KPC_MPR_ITER 50                 /* mjOption.mpr_iterations default */
#define KPC_MAXPAIR 8                   /* most contacts one geom pair can produce here (box-box) */

/* a convex shape for the support-function code: box (0), z-axis cylinder (1), hull mesh (2) */
typedef struct {
    int type;
    double size[3];        /* box half sizes | cylinder radius, half height */
    double pos[3];         /* frame origin (hull: body origin) */
    double mat[9];         /* world rotation of that frame */
    double center[3];      /* mjccd_center: geom_xpos (hull: the mesh's COM, where MuJoCo puts the mesh geom frame) */
    const double *verts;   /* hull: [nvert][3] in the body frame */
    int nvert;
} kpc_shape;

typedef struct { double dist, pos[3], normal[3]; } kpc_contact;   /* normal points from geom 1 to geom 2 */

static int kpc_is_zero(double x) { return fabs(x) < KPC_EPS; }
static int kpc_eq(double a, double b) {                              /* ccdEq */
    double ab = fabs(a - b);
    if (ab < KPC_EPS) return 1;
    a = fabs(a); b = fabs(b);
    return b > a ? ab < KPC_EPS * b : ab < KPC_EPS * a;
}
static void kpc_sub(double *r, const double *a, const double *b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static void kpc_cross(double *r, const double *a, const double *b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
static double kpc_dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void kpc_normalize(double *a) { double n = sqrt(kpc_dot(a, a)); a[0] /= n; a[1] /= n; a[2] /= n; }   /* ccdVec3Normalize */

/* mjccd_support: farthest point of the shape in world direction dir (unit), inflated by `margin` along dir  [MJ-ext] */
static void kpc_support(const kpc_shape *s, const double *dir, double margin, double *out) {
    double l[3], p[3];
    FL(15 + 21 + (s->type == 2 ? 5 * s->nvert : 6));
    for (int k = 0; k < 3; k++) l[k] = s->mat[k] * dir[0] + s->mat[3 + k] * dir[1] + s->mat[6 + k] * dir[2];   /* R^T dir */
    if (s->type == 0) { for (int k = 0; k < 3; k++) p[k] = (l[k] > 0 ? 1.0 : -1.0) * s->size[k]; }
    else if (s->type == 1) {
        double n = sqrt(l[0] * l[0] + l[1] * l[1]);
        if (n > 1e-15) { p[0] = l[0] / n * s->size[0]; p[1] = l[1] / n * s->size[0]; } else p[0] = p[1] = 0;
        p[2] = (l[2] > 0 ? 1.0 : (l[2] < 0 ? -1.0 : 0.0)) * s->size[1];
    } else {                                                  /* exhaustive search, first maximum (the result of MuJoCo's
                                                                 hill climbing on the hull graph wherever the maximum is unique) */
        int best = 0; double bd = -1e300;
        for (int v = 0; v < s->nvert; v++) { double d = kpc_dot(l, s->verts + 3 * v); if (d > bd) { bd = d; best = v; } }
        memcpy(p, s->verts + 3 * best, 24);
    }
    for (int k = 0; k < 3; k++) out[k] = s->pos[k] + s->mat[3 * k] * p[0] + s->mat[3 * k + 1] * p[1] + s->mat[3 * k + 2] * p[2] + margin * dir[k];
}

/* ---------------------------------------------------------------- libccd MPR (src/mpr.c)  [MJ-ext] */
typedef struct { double v[3], v1[3], v2[3]; } kpc_sup;    /* Minkowski-difference point v = v1 - v2 and its two witnesses */
static void kpc_mink(const kpc_shape *a, const kpc_shape *b, double margin, const double *dir, kpc_sup *o) {   /* __ccdSupport */
    double nd[3] = {-dir[0], -dir[1], -dir[2]};
    kpc_support(a, dir, margin, o->v1); kpc_support(b, nd, margin, o->v2); kpc_sub(o->v, o->v1, o->v2);
}
static void kpc_portal_dir(const kpc_sup *p, double *dir) {
    double a[3], b[3]; kpc_sub(a, p[2].v, p[1].v); kpc_sub(b, p[3].v, p[1].v); kpc_cross(dir, a, b); kpc_normalize(dir);
}
static int kpc_reach_tol(const kpc_sup *p, const kpc_sup *v4, const double *dir) {   /* portalReachTolerance */
    double dv4 = kpc_dot(v4->v, dir), d1 = dv4 - kpc_dot(p[1].v, dir), d2 = dv4 - kpc_dot(p[2].v, dir), d3 = dv4 - kpc_dot(p[3].v, dir);
    double m = fmin(d1, fmin(d2, d3));
    return kpc_eq(m, KPC_MPR_TOL) || m < KPC_MPR_TOL;
}
static void kpc_expand(kpc_sup *p, const kpc_sup *v4) {                                /* expandPortal */
    double v4v0[3]; kpc_cross(v4v0, v4->v, p[0].v);
    if (kpc_dot(p[1].v, v4v0) > 0) { if (kpc_dot(p[2].v, v4v0) > 0) p[1] = *v4; else p[3] = *v4; }
    else { if (kpc_dot(p[3].v, v4v0) > 0) p[2] = *v4; else p[1] = *v4; }
}
static double kpc_pt_seg_dist2(const double *x0, const double *b, double *w) {         /* __ccdVec3PointSegmentDist2 with P = origin */
    double d[3]; kpc_sub(d, b, x0);
    double t = -kpc_dot(x0, d) / kpc_dot(d, d);
    if (t < 0 || kpc_is_zero(t)) memcpy(w, x0, 24);
    else if (t > 1 || kpc_eq(t, 1)) memcpy(w, b, 24);
    else for (int k = 0; k < 3; k++) w[k] = x0[k] + t * d[k];
    return kpc_dot(w, w);
}
static double kpc_pt_tri_dist2(const double *x0, const double *B, const double *C, double *w) {   /* ccdVec3PointTriDist2, P = origin */
    double d1[3], d2[3]; kpc_sub(d1, B, x0); kpc_sub(d2, C, x0);
    double v = kpc_dot(d1, d1), ww = kpc_dot(d2, d2), p = kpc_dot(x0, d1), q = kpc_dot(x0, d2), r = kpc_dot(d1, d2);
    double s = (q * r - ww * p) / (ww * v - r * r), t = (-s * r - q) / ww;
    if ((kpc_is_zero(s) || s > 0) && (kpc_eq(s, 1) || s < 1) && (kpc_is_zero(t) || t > 0) && (kpc_eq(t, 1) || t < 1) && (kpc_eq(t + s, 1) || t + s < 1)) {
        for (int k = 0; k < 3; k++) w[k] = x0[k] + s * d1[k] + t * d2[k];
        return kpc_dot(w, w);
    }
    double w2[3], dist = kpc_pt_seg_dist2(x0, B, w), d;
    d = kpc_pt_seg_dist2(x0, C, w2); if (d < dist) { dist = d; memcpy(w, w2, 24); }
    d = kpc_pt_seg_dist2(B, C, w2); if (d < dist) { dist = d; memcpy(w, w2, 24); }
    return dist;
}
static void kpc_find_pos(const kpc_sup *p, double *pos) {                              /* findPos */
    double dir[3], b[4], t[3], sum;
    kpc_portal_dir(p, dir);
    kpc_cross(t, p[1].v, p[2].v); b[0] = kpc_dot(t, p[3].v);
    kpc_cross(t, p[3].v, p[2].v); b[1] = kpc_dot(t, p[0].v);
    kpc_cross(t, p[0].v, p[1].v); b[2] = kpc_dot(t, p[3].v);
    kpc_cross(t, p[2].v, p[1].v); b[3] = kpc_dot(t, p[0].v);
    sum = b[0] + b[1] + b[2] + b[3];
    if (kpc_is_zero(sum) || sum < 0) {
        b[0] = 0;
        kpc_cross(t, p[2].v, p[3].v); b[1] = kpc_dot(t, dir);
        kpc_cross(t, p[3].v, p[1].v); b[2] = kpc_dot(t, dir);
        kpc_cross(t, p[1].v, p[2].v); b[3] = kpc_dot(t, dir);
        sum = b[1] + b[2] + b[3];
    }
    for (int k = 0; k < 3; k++) {
        double p1 = 0, p2 = 0;
        for (int i = 0; i < 4; i++) { p1 += b[i] * p[i].v1[k]; p2 += b[i] * p[i].v2[k]; }
        pos[k] = 0.5 * (p1 + p2) / sum;
    }
}
static int kpc_stat_discover, kpc_stat_refine, kpc_stat_penetr;     /* iteration counts of the last kpc_mpr call (diagnostics) */
/* ccdMPRPenetration: 0 = the (inflated) shapes intersect, depth / dir (from shape a towards shape b... see kpo_convex) / pos filled */
static int kpc_mpr(const kpc_shape *A, const kpc_shape *B, double margin, double *depth, double *dir_out, double *pos) {
    kpc_sup p[4], v4;
    double dir[3], va[3], vb[3], dot;
    /* discoverPortal */
    memcpy(p[0].v1, A->center, 24); memcpy(p[0].v2, B->center, 24); kpc_sub(p[0].v, p[0].v1, p[0].v2);
    if (kpc_eq(p[0].v[0], 0) && kpc_eq(p[0].v[1], 0) && kpc_eq(p[0].v[2], 0)) p[0].v[0] += KPC_EPS * 10;
    for (int k = 0; k < 3; k++) dir[k] = -p[0].v[k];
    kpc_normalize(dir);
    kpc_mink(A, B, margin, dir, &p[1]);
    dot = kpc_dot(p[1].v, dir);
    if (kpc_is_zero(dot) || dot < 0) return -1;
    kpc_cross(dir, p[0].v, p[1].v);
    if (kpc_is_zero(kpc_dot(dir, dir))) {
        if (kpc_eq(p[1].v[0], 0) && kpc_eq(p[1].v[1], 0) && kpc_eq(p[1].v[2], 0)) {      /* touching contact (findPenetrTouch) */
            *depth = 0; dir_out[0] = dir_out[1] = dir_out[2] = 0;
            for (int k = 0; k < 3; k++) pos[k] = 0.5 * (p[1].v1[k] + p[1].v2[k]);
            return 0;
        }
        *depth = sqrt(kpc_dot(p[1].v, p[1].v));                                           /* origin on the segment v0-v1 (findPenetrSegment) */
        memcpy(dir_out, p[1].v, 24); kpc_normalize(dir_out);
        for (int k = 0; k < 3; k++) pos[k] = 0.5 * (p[1].v1[k] + p[1].v2[k]);
        return 0;
    }
    kpc_normalize(dir);
    kpc_mink(A, B, margin, dir, &p[2]);
    dot = kpc_dot(p[2].v, dir);
    if (kpc_is_zero(dot) || dot < 0) return -1;
    kpc_sub(va, p[1].v, p[0].v); kpc_sub(vb, p[2].v, p[0].v); kpc_cross(dir, va, vb); kpc_normalize(dir);
    if (kpc_dot(dir, p[0].v) > 0) { kpc_sup t = p[1]; p[1] = p[2]; p[2] = t; for (int k = 0; k < 3; k++) dir[k] = -dir[k]; }
    kpc_stat_discover = kpc_stat_refine = kpc_stat_penetr = 0;
    for (int guard = 0; guard < 1000; guard++) {
        kpc_stat_discover++;
        kpc_mink(A, B, margin, dir, &p[3]);
        dot = kpc_dot(p[3].v, dir);
        if (kpc_is_zero(dot) || dot < 0) return -1;
        int cont = 0;
        kpc_cross(va, p[1].v, p[3].v); dot = kpc_dot(va, p[0].v);
        if (dot < 0 && !kpc_is_zero(dot)) { p[2] = p[3]; cont = 1; }
        if (!cont) { kpc_cross(va, p[3].v, p[2].v); dot = kpc_dot(va, p[0].v); if (dot < 0 && !kpc_is_zero(dot)) { p[1] = p[3]; cont = 1; } }
        if (!cont) break;
        kpc_sub(va, p[1].v, p[0].v); kpc_sub(vb, p[2].v, p[0].v); kpc_cross(dir, va, vb); kpc_normalize(dir);
    }
    /* refinePortal */
    for (int guard = 0; guard < 1000; guard++) {
        kpc_stat_refine++;
        kpc_portal_dir(p, dir);
        dot = kpc_dot(dir, p[1].v);
        if (kpc_is_zero(dot) || dot > 0) break;                                           /* portalEncapsulesOrigin */
        kpc_mink(A, B, margin, dir, &v4);
        dot = kpc_dot(v4.v, dir);
        if (!(kpc_is_zero(dot) || dot > 0) || kpc_reach_tol(p, &v4, dir)) return -1;     /* cannot enclose the origin: no intersection */
        kpc_expand(p, &v4);
    }
    /* findPenetr */
    for (int it = 0;; it++) {
        kpc_stat_penetr = it;
        kpc_portal_dir(p, dir);
        kpc_mink(A, B, margin, dir, &v4);
        if (kpc_reach_tol(p, &v4, dir) || it > KPC_MPR_ITER) {
            double w[3];
            *depth = sqrt(kpc_pt_tri_dist2(p[1].v, p[2].v, p[3].v, w));
            if (kpc_is_zero(*depth)) memcpy(w, dir, 24);                                   /* origin on the portal: its normal */
            memcpy(dir_out, w, 24); kpc_normalize(dir_out);
            kpc_find_pos(p, pos);
            return 0;
        }
        kpc_expand(p, &v4);
    }
}

/* mjc_Convex  [MJ-ext]: both shapes inflated by margin / 2 (mjccd_support adds the margin along the query direction), one
 * ccdMPRPenetration(obj1 = geom 1, obj2 = geom 2) query, dist = margin - depth, contact position = libccd's `pos`, ONE contact.
 * libccd's Minkowski difference is v = support1(dir) - support2(-dir) and v0 = center1 - center2, the portal is searched from v0
 * through the origin, so the returned direction points from object 1 towards object 2 ("the direction in which obj2 has to be
 * translated by depth to separate"): it is MuJoCo's contact normal (geom 1 -> geom 2) as is. */
static int kpo_convex(const kpc_shape *g1, const kpc_shape *g2, double margin, kpc_contact *con) {
    double depth, dir[3], pos[3];
    if (kpc_mpr(g1, g2, 0.5 * margin, &depth, dir, pos) != 0) return 0;
    if (dir[0] == 0 && dir[1] == 0 && dir[2] == 0) return 0;                               /* touching: normal undefined, no contact */
    con->dist = margin - depth;
    for (int k = 0; k < 3; k++) { con->normal[k] = dir[k]; con->pos[k] = pos[k]; }
    return 1;
}

/* mjc_PlaneConvex for a mesh geom  [MJ-ext]: the support vertex in direction -normal (deepest vertex) makes the first contact; then
 * ITS NEIGHBOURS on the hull graph are visited in graph order while fewer than `maxcon` (maxplanemesh = 3) contacts exist, and a
 * neighbour within the margin is added (addplanemesh) unless it lies closer than tol_rbound = tolplanemesh * geom_rbound (0.3 x the
 * mesh geom's bounding radius) to the FIRST contact's position.  plane = world z = 0 (the floor geom of the XML).
 * nbr / nbr_adr: hull graph (model compiler).  Contact position = vertex - normal * dist / 2. */
static int kpo_plane_mesh(const kpc_shape *h, const int *nbr_adr, const int *nbr, double margin, double tol_rbound, int maxcon, kpc_contact *con) {
    FL(6 * h->nvert);
    int best = -1; double bd = 1e300;
    for (int v = 0; v < h->nvert; v++) {
        const double *p = h->verts + 3 * v;
        double z = h->pos[2] + h->mat[6] * p[0] + h->mat[7] * p[1] + h->mat[8] * p[2];
        if (z < bd) { bd = z; best = v; }                                                  /* support point of -normal, first extremum */
    }
    if (best < 0 || bd > margin) return 0;
    int cnt = 0;
    for (int k = -1; k < nbr_adr[best + 1] - nbr_adr[best] && (k < 0 || cnt < maxcon); k++) {
        int v = k < 0 ? best : nbr[nbr_adr[best] + k];
        const double *p = h->verts + 3 * v;
        double w[3];
        for (int a = 0; a < 3; a++) w[a] = h->pos[a] + h->mat[3 * a] * p[0] + h->mat[3 * a + 1] * p[1] + h->mat[3 * a + 2] * p[2];
        if (k >= 0) {
            if (w[2] > margin) continue;
            double d[3]; kpc_sub(d, w, con[0].pos);                                        /* "skip if too close to first contact" */
            if (kpc_dot(d, d) < tol_rbound * tol_rbound) continue;
        }
        con[cnt].dist = w[2]; con[cnt].normal[0] = con[cnt].normal[1] = 0; con[cnt].normal[2] = 1;
        con[cnt].pos[0] = w[0]; con[cnt].pos[1] = w[1]; con[cnt].pos[2] = w[2] - 0.5 * w[2];
        cnt++;
    }
    return cnt;
}

/* mjc_PlaneBox  [MJ-ext]: the 8 corners in index order (bit 0 / 1 / 2 = +x / +y / +z), skipping those that point away from the
 * plane (ldist > 0) or lie beyond the margin; at most 4. */
static int kpo_plane_box(const kpc_shape *b, double margin, kpc_contact *con) {
    int cnt = 0;
    double dist = b->pos[2];
    for (int i = 0; i < 8; i++) {
        double vec[3] = {(i & 1) ? b->size[0] : -b->size[0], (i & 2) ? b->size[1] : -b->size[1], (i & 4) ? b->size[2] : -b->size[2]}, corner[3];
        for (int a = 0; a < 3; a++) corner[a] = b->mat[3 * a] * vec[0] + b->mat[3 * a + 1] * vec[1] + b->mat[3 * a + 2] * vec[2];
        double ldist = corner[2];
        if (dist + ldist > margin || ldist > 0) continue;
        con[cnt].dist = dist + ldist; con[cnt].normal[0] = con[cnt].normal[1] = 0; con[cnt].normal[2] = 1;
        for (int a = 0; a < 3; a++) con[cnt].pos[a] = corner[a] + b->pos[a];
        con[cnt].pos[2] -= 0.5 * con[cnt].dist;
        if (++cnt >= 4) return 4;
    }
    return cnt;
}

/* mjc_PlaneCylinder  [MJ-ext]: nearest rim point of the lower cap, the matching point of the other cap, and two more points of
 * the lower cap at +-120 degrees ("triangle") when they are within the margin. */
static int kpo_plane_cylinder(const kpc_shape *c, double margin, kpc_contact *con) {
    double normal[3] = {0, 0, 1}, axis[3] = {c->mat[2], c->mat[5], c->mat[8]}, vec[3], vec1[3];
    int cnt = 0;
    double prjaxis = kpc_dot(normal, axis);
    if (prjaxis > 0) { for (int k = 0; k < 3; k++) axis[k] = -axis[k]; prjaxis = -prjaxis; }
    double dist0 = c->pos[2];
    for (int k = 0; k < 3; k++) vec[k] = axis[k] * prjaxis - normal[k];                   /* -normal with its axial component removed */
    double len2 = kpc_dot(vec, vec);
    if (len2 >= 1e-12) { double sc = c->size[0] / sqrt(len2); for (int k = 0; k < 3; k++) vec[k] *= sc; }
    else { vec[0] = c->mat[0] * c->size[0]; vec[1] = c->mat[3] * c->size[0]; vec[2] = c->mat[6] * c->size[0]; }   /* disk parallel to the plane: the cylinder's x axis */
    double prjvec = kpc_dot(vec, normal);
    for (int k = 0; k < 3; k++) axis[k] *= c->size[1];
    prjaxis *= c->size[1];
    if (dist0 + prjaxis + prjvec > margin) return 0;
    con[cnt].dist = dist0 + prjaxis + prjvec;
    for (int k = 0; k < 3; k++) con[cnt].pos[k] = c->pos[k] + vec[k] + axis[k] - normal[k] * con[cnt].dist * 0.5;
    cnt++;
    if (dist0 - prjaxis + prjvec <= margin) {
        con[cnt].dist = dist0 - prjaxis + prjvec;
        for (int k = 0; k < 3; k++) con[cnt].pos[k] = c->pos[k] + vec[k] - axis[k] - normal[k] * con[cnt].dist * 0.5;
        cnt++;
    }
    double prjvec1 = -prjvec * 0.5;
    if (dist0 + prjaxis + prjvec1 <= margin) {
        kpc_cross(vec1, vec, axis); kpc_normalize(vec1);
        for (int k = 0; k < 3; k++) vec1[k] *= c->size[0] * sqrt(3.0) * 0.5;
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
            con[cnt].dist = dist0 + prjaxis + prjvec1;
            for (int k = 0; k < 3; k++) con[cnt].pos[k] = c->pos[k] + sgn * vec1[k] + axis[k] - vec[k] * 0.5 - normal[k] * con[cnt].dist * 0.5;
            cnt++;
        }
    }
    for (int i = 0; i < cnt; i++) { con[i].normal[0] = con[i].normal[1] = 0; con[i].normal[2] = 1; }
    return cnt;
}

/* box - box.  MuJoCo's mjc_BoxBox (engine_collision_box.c, ~700 lines) is a separating-axis test over the 15 candidate axes
 * followed by face clipping (face contact: the incident face clipped against the reference face, up to 8 points) or one
 * closest-point contact (edge-edge).  Its exact candidate ordering and tolerances cannot be recalled line by line, so this is a
 * restatement of the ALGORITHM, not of the code  [MJ-ext, structure only]: same axes, same two cases, the contact points of the
 * face case are the clipped polygon's vertices with dist = signed distance to the reference face, position midway.  Normal from
 * box 1 to box 2.  In the reference's data the only box-box pair is the pushed box resting on the table top (face contact with
 * four clipped corners), where any correct clipping routine returns the same four points. */
static int kpo_box_box(const kpc_shape *A, const kpc_shape *B, double margin, kpc_contact *con) {
    double d[3]; kpc_sub(d, B->pos, A->pos);
    double Ra[3][3], Rb[3][3];                                      /* columns = world axes of each box */
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { Ra[i][k] = A->mat[3 * k + i]; Rb[i][k] = B->mat[3 * k + i]; }
    double best = -1e300; int bt = -1, bi = 0, bj = 0; double bn[3] = {0, 0, 0};
    /* face axes of A (type 0), of B (type 1), edge x edge (type 2); separation s = |d.n| - ra - rb, the largest (least negative) wins;
       an edge axis must beat the best face axis by 5 % (the usual guard against jitter between near-parallel axes) */
    for (int t = 0; t < 2; t++) for (int i = 0; i < 3; i++) {
        const double *n = t == 0 ? Ra[i] : Rb[i];
        double ra = 0, rb = 0;
        for (int k = 0; k < 3; k++) { ra += A->size[k] * fabs(kpc_dot(Ra[k], n)); rb += B->size[k] * fabs(kpc_dot(Rb[k], n)); }
        double s = fabs(kpc_dot(d, n)) - ra - rb;
        if (s > margin) return 0;
        if (s > best) { best = s; bt = t; bi = i; double sg = kpc_dot(d, n) < 0 ? -1.0 : 1.0; for (int k = 0; k < 3; k++) bn[k] = sg * n[k]; }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double n[3]; kpc_cross(n, Ra[i], Rb[j]);
        double l = sqrt(kpc_dot(n, n));
        if (l < 1e-6) continue;
        for (int k = 0; k < 3; k++) n[k] /= l;
        double ra = 0, rb = 0;
        for (int k = 0; k < 3; k++) { ra += A->size[k] * fabs(kpc_dot(Ra[k], n)); rb += B->size[k] * fabs(kpc_dot(Rb[k], n)); }
        double s = fabs(kpc_dot(d, n)) - ra - rb;
        if (s > margin) return 0;
        if (s > best + 0.05 * fabs(best) + 1e-9) { best = s; bt = 2; bi = i; bj = j; double sg = kpc_dot(d, n) < 0 ? -1.0 : 1.0; for (int k = 0; k < 3; k++) bn[k] = sg * n[k]; }
    }
    if (bt == 2) {                                                  /* edge - edge: closest points of the two supporting edges */
        double pa[3], pb[3];
        for (int k = 0; k < 3; k++) { pa[k] = A->pos[k]; pb[k] = B->pos[k]; }
        for (int a = 0; a < 3; a++) if (a != bi) { double sg = kpc_dot(bn, Ra[a]) > 0 ? 1.0 : -1.0; for (int k = 0; k < 3; k++) pa[k] += sg * A->size[a] * Ra[a][k]; }
        for (int a = 0; a < 3; a++) if (a != bj) { double sg = kpc_dot(bn, Rb[a]) > 0 ? -1.0 : 1.0; for (int k = 0; k < 3; k++) pb[k] += sg * B->size[a] * Rb[a][k]; }
        const double *ua = Ra[bi], *ub = Rb[bj];
        double w[3]; kpc_sub(w, pa, pb);
        double b = kpc_dot(ua, ub), dd = kpc_dot(ua, w), e = kpc_dot(ub, w), den = 1 - b * b;
        double sa = den > 1e-12 ? (b * e - dd) / den : 0, sb = den > 1e-12 ? (e - b * dd) / den : 0;
        sa = fmax(-A->size[bi], fmin(A->size[bi], sa)); sb = fmax(-B->size[bj], fmin(B->size[bj], sb));
        for (int k = 0; k < 3; k++) { pa[k] += sa * ua[k]; pb[k] += sb * ub[k]; con[0].pos[k] = 0.5 * (pa[k] + pb[k]); con[0].normal[k] = bn[k]; }
        con[0].dist = best;
        return 1;
    }
    /* face contact: reference box R (its face with outward normal nr), incident box I */
    const kpc_shape *Rf = bt == 0 ? A : B, *If = bt == 0 ? B : A;
    double (*Rr)[3] = bt == 0 ? Ra : Rb, (*Ri)[3] = bt == 0 ? Rb : Ra;
    double nr[3]; for (int k = 0; k < 3; k++) nr[k] = bt == 0 ? bn[k] : -bn[k];    /* outward normal of the reference face (points at the incident box) */
    int ia = 0; double mn = 1e300;                                   /* incident face: the one most anti-parallel to nr */
    for (int a = 0; a < 3; a++) { double c = kpc_dot(Ri[a], nr); if (-fabs(c) < mn) { mn = -fabs(c); ia = a; } }
    double sgi = kpc_dot(Ri[ia], nr) > 0 ? -1.0 : 1.0;
    int u = (ia + 1) % 3, v = (ia + 2) % 3;
    double poly[16][3]; int np = 4;
    for (int c = 0; c < 4; c++) {
        double su = (c == 0 || c == 3) ? -1.0 : 1.0, sv = c < 2 ? -1.0 : 1.0;
        for (int k = 0; k < 3; k++) poly[c][k] = If->pos[k] + sgi * If->size[ia] * Ri[ia][k] + su * If->size[u] * Ri[u][k] + sv * If->size[v] * Ri[v][k];
    }
    int ru = (bi + 1) % 3, rv = (bi + 2) % 3;
    for (int side = 0; side < 4; side++) {                          /* Sutherland-Hodgman against the four side planes of the reference face */
        const double *ax = side < 2 ? Rr[ru] : Rr[rv];
        double sg = (side & 1) ? -1.0 : 1.0, lim = side < 2 ? Rf->size[ru] : Rf->size[rv];
        double out[16][3]; int no = 0;
        for (int i = 0; i < np; i++) {
            const double *p = poly[i], *q = poly[(i + 1) % np];
            double rp[3], rq[3]; kpc_sub(rp, p, Rf->pos); kpc_sub(rq, q, Rf->pos);
            double dp = sg * kpc_dot(rp, ax) - lim, dq = sg * kpc_dot(rq, ax) - lim;
            if (dp <= 0) { memcpy(out[no++], p, 24); }
            if ((dp <= 0) != (dq <= 0)) { double t = dp / (dp - dq); for (int k = 0; k < 3; k++) out[no][k] = p[k] + t * (q[k] - p[k]); no++; }
        }
        np = no; memcpy(poly, out, sizeof(double) * 3 * no);
        if (np == 0) return 0;
    }
    int cnt = 0;
    for (int i = 0; i < np && cnt < KPC_MAXPAIR; i++) {
        double r[3]; kpc_sub(r, poly[i], Rf->pos);
        double dist = kpc_dot(r, nr) - Rf->size[bi];                /* signed distance of the clipped incident vertex to the reference face */
        if (dist > margin) continue;
        con[cnt].dist = dist;
        for (int k = 0; k < 3; k++) { con[cnt].pos[k] = poly[i][k] - 0.5 * dist * nr[k]; con[cnt].normal[k] = bn[k]; }
        cnt++;
    }
    return cnt;
}

#endif
