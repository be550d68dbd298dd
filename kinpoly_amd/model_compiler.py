"""Model compiler: MuJoCo-XML (+ binary STL hulls) -> compiled KPM blob.

The reference hands its humanoid to MuJoCo as
``assets/mujoco_models/humanoid_smpl_neutral_mesh*.xml`` (reference
``uhc/khrylib/rl/envs/common/mujoco_env.py:16-44``) and lets ``libmujoco`` compile it
(``mujoco_py.load_model_from_path``).  MuJoCo is not part of this engine, so the
compile step is done here: this module parses the same XML (``coordinate="global"``,
``angle="degree"``, ``inertiafromgeom="true"``; XML ``:2,11-16``) and the binary STL
convex hulls under ``geom/``, and emits every constant the HIP simulator needs:

* kinematic tree (parents, local offsets), per-body mass / COM / inertia from the
  mesh at density 1000 kg/m^3 (exact polyhedral integration),
* dof tables in MuJoCo's ``qM`` sparse layout (dof_parent, dof_madr, armature),
* hull vertices per body, the vertex adjacency graph of each convex hull (qhull, as MuJoCo's mesh compiler builds
  `mesh_graph` for its plane-mesh / support-function code) and bounding radii,
* stable-PD gains / torque limits / RFC parameters from ``config/uhc/uhc.yml:81-156``,
* constraint-model constants evaluated at ``qpos0`` (body/dof ``invweight0``,
  ``meaninertia``) the soft-contact model needs,
* free objects (chair, box, table, Can, step) geoms, recorded for later rounds.

Blob layout ("KPM1", little endian):
    u32 magic 'KPM1' | u32 version | u32 n_entries | entries[n] | payload
    entry = char name[32] | u32 dtype (0=f64, 1=i32) | u32 pad | u64 count | u64 byte offset
Both the product loader (``csrc/kp_model.hpp``) and the oracle loader
(``oracle/kp_oracle.c``) parse this format independently.

Usage:  python -m kinpoly_amd.model_compiler <model.xml> <uhc.yml> <out.kpm>
"""
from __future__ import annotations

import math
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np

KPM_MAGIC = 0x314D504B  # 'KPM1'
KPM_VERSION = 7

# MuJoCo 2.1.0 defaults that the reference never overrides [MJ-ext]
MJ_DEFAULTS = dict(
    gravity=(0.0, 0.0, -9.81),
    density=1000.0,
    solref=(0.02, 1.0),
    solimp=(0.9, 0.95, 0.001, 0.5, 2.0),
    geom_friction=(1.0, 0.005, 0.0001),
    impratio=1.0,
    solver_iterations=100,
    solver_tolerance=1e-8,
    # engine_collision_convex.c, mjc_PlaneConvex [MJ-ext]: at most `maxplanemesh` contacts per plane-mesh pair (the support vertex +
    # hull-graph neighbours), a neighbour closer than `tolplanemesh` * geom_rbound to the first contact is skipped.  Recalled from the
    # MuJoCo source by two independent readers; both are model options of the simulators ("planemesh_max", "planemesh_tol")
    maxplanemesh=3,
    tolplanemesh=0.3,
)


# --------------------------------------------------------------------------- STL / inertia
def read_binary_stl(path: str) -> np.ndarray:
    """Return triangles [n,3,3] (float64) of a binary STL file."""
    with open(path, "rb") as f:
        buf = f.read()
    (ntri,) = struct.unpack_from("<I", buf, 80)
    assert len(buf) >= 84 + 50 * ntri, f"truncated STL {path}"
    rec = np.frombuffer(buf, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                        count=ntri, offset=84)
    return rec["v"].astype(np.float64)


def polyhedron_mass_props(tris: np.ndarray, density: float):
    """Exact mass, COM and inertia-about-COM of a closed triangle mesh.

    Signed tetrahedra against the vertex centroid; |volume| per tetra so that the triangle
    winding does not matter for a convex hull (what MuJoCo 2.1's mesh compiler does for
    meshes, [MJ-ext]).  Inertia returned as full 3x3 in the mesh frame about the COM.
    Plain loops in the order kinpoly_amd/csrc/kp_compile.hpp::mass_props runs them (the two compilers write the same bytes)."""
    ref = [0.0, 0.0, 0.0]
    for t in tris:
        for vtx in t:
            ref = [ref[0] + float(vtx[0]), ref[1] + float(vtx[1]), ref[2] + float(vtx[2])]
    n3 = 3.0 * float(len(tris))
    ref = [ref[0] / n3, ref[1] / n3, ref[2] / n3]
    V = 0.0
    cr = [0.0, 0.0, 0.0]
    C = [0.0] * 9
    for t in tris:
        a = [float(t[0][k]) - ref[k] for k in range(3)]; b = [float(t[1][k]) - ref[k] for k in range(3)]; c = [float(t[2][k]) - ref[k] for k in range(3)]
        bc = [b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]]
        vol = abs(a[0] * bc[0] + a[1] * bc[1] + a[2] * bc[2]) / 6.0
        V += vol
        for k in range(3):
            cr[k] += vol * ((a[k] + b[k] + c[k]) / 4.0)
        sv = [a[0] + b[0] + c[0], a[1] + b[1] + c[1], a[2] + b[2] + c[2]]
        for q in (a, b, c, sv):
            for i in range(3):
                for j in range(3):
                    C[3 * i + j] += (vol / 20.0) * q[i] * q[j]
    cm = [cr[0] / V, cr[1] / V, cr[2] / V]
    for i in range(3):
        for j in range(3):
            C[3 * i + j] -= V * cm[i] * cm[j]
    tr = C[0] + C[4] + C[8]
    inertia = np.array([[((tr if i == j else 0.0) - C[3 * i + j]) * density for j in range(3)] for i in range(3)])
    return V * density, np.array([ref[0] + cm[0], ref[1] + cm[1], ref[2] + cm[2]]), inertia


def eigh3(A):
    """eigenvectors (columns) of a symmetric 3 x 3 by cyclic Jacobi rotations: the same sweeps as kp_compile.hpp::eigh3"""
    a = [[float(A[i][j]) for j in range(3)] for i in range(3)]
    v = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    for _ in range(64):
        off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2]
        if off < 1e-40:
            break
        for p in range(2):
            for q in range(p + 1, 3):
                if abs(a[p][q]) < 1e-300:
                    continue
                th = (a[q][q] - a[p][p]) / (2.0 * a[p][q])
                tt = (1.0 if th >= 0 else -1.0) / (abs(th) + math.sqrt(th * th + 1.0))
                c = 1.0 / math.sqrt(tt * tt + 1.0); s_ = tt * c
                for k in range(3):
                    akp, akq = a[k][p], a[k][q]; a[k][p] = c * akp - s_ * akq; a[k][q] = s_ * akp + c * akq
                for k in range(3):
                    apk, aqk = a[p][k], a[q][k]; a[p][k] = c * apk - s_ * aqk; a[q][k] = s_ * apk + c * aqk
                for k in range(3):
                    vkp, vkq = v[k][p], v[k][q]; v[k][p] = c * vkp - s_ * vkq; v[k][q] = s_ * vkp + c * vkq
    return np.array(v)


def seq_sum(x, axis=-1):
    """left-to-right sum (np.sum adds pairwise): the order the C++ compiler's loops accumulate in"""
    return np.take(np.cumsum(x, axis=axis), -1, axis=axis)


def gauss_jordan_inverse(A):
    """Gauss-Jordan with partial pivoting, row operation by row operation as kp_compile.hpp::invert (LAPACK would round differently)"""
    A = np.array(A, float); n = A.shape[0]
    inv = np.eye(n)
    for c in range(n):
        p = c + int(np.argmax(np.abs(A[c:, c])))
        if p != c:
            A[[p, c]] = A[[c, p]]; inv[[p, c]] = inv[[c, p]]
        d = 1.0 / A[c, c]
        A[c] *= d; inv[c] *= d
        for r in range(n):
            if r != c and A[r, c] != 0.0:
                f = A[r, c]
                A[r] -= f * A[c]; inv[r] -= f * inv[c]
    return inv


def hull_graph_qhull(v: np.ndarray):
    """Vertex adjacency of the convex hull of `v` [n,3] through qhull ("Qt": triangulated facets, the option MuJoCo's mesh compiler uses),
    neighbour lists in qhull's facet order.  Kept as a cross-check of hull_graph's edge set (tests/test_host_cpu.py): the two agree except for
    WHICH diagonal triangulates a coplanar face and for the order of a vertex's neighbours -- both are artefacts of qhull's run on ITS input order,
    and MuJoCo's own input order (its STL import) cannot be reproduced anyway [MJ-ext]."""
    from scipy.spatial import ConvexHull
    h = ConvexHull(v, qhull_options="Qt")
    assert len(h.vertices) == len(v), "mesh has vertices inside its convex hull"
    lists = [[] for _ in range(len(v))]
    for tri in h.simplices:
        for a in range(3):
            for c in range(3):
                if c != a and int(tri[c]) not in lists[int(tri[a])]:
                    lists[int(tri[a])].append(int(tri[c]))
    return lists


def hull_graph(v: np.ndarray):
    """Vertex adjacency of the convex hull of `v` [n,3] (every row a hull vertex, n <= 64), built by a rule that needs no library and that the
    C++ compiler (kinpoly_amd/csrc/kp_compile.hpp) restates operation for operation, so both write the same graph:

      * a vertex triple (i < j < k) spans a FACE when every other vertex lies on one side of its plane (|signed distance| <= 1e-9 m counts as
        on the plane); the face is the set of all vertices on that plane;
      * a face's boundary edges are hull edges; a face with more than three vertices (coplanar STL quads) is triangulated as a fan from its
        lowest-numbered vertex, the fan's diagonals are edges too (MuJoCo's mesh graph comes from qhull's TRIANGULATED facets, so coplanar
        faces contribute diagonals there as well -- which ones is qhull's business, see hull_graph_qhull);
      * the neighbours of a vertex are listed in ascending vertex number (vertices are numbered by np.unique's lexicographic row order).

    mjc_PlaneConvex takes the first maxplanemesh - 1 qualifying neighbours in list order; with MuJoCo's own vertex numbering unknowable, any
    fixed order is as (un)verifiable as qhull's [MJ-ext]."""
    n = len(v)
    assert 4 <= n <= 64
    tol = 1e-9
    adj = np.zeros((n, n), bool)
    seen = set()
    for i in range(n):
        for j in range(i + 1, n):
            eij = v[j] - v[i]
            for k in range(j + 1, n):
                eik = v[k] - v[i]
                nx = eij[1] * eik[2] - eij[2] * eik[1]; ny = eij[2] * eik[0] - eij[0] * eik[2]; nz = eij[0] * eik[1] - eij[1] * eik[0]
                ln = math.sqrt(nx * nx + ny * ny + nz * nz)
                if ln < 1e-14:
                    continue
                nx /= ln; ny /= ln; nz /= ln
                pos = neg = False
                face = []
                for m in range(n):
                    d = nx * (v[m][0] - v[i][0]) + ny * (v[m][1] - v[i][1]) + nz * (v[m][2] - v[i][2])
                    if d > tol:
                        pos = True
                    elif d < -tol:
                        neg = True
                    else:
                        face.append(m)
                    if pos and neg:
                        break
                if pos and neg:
                    continue
                key = tuple(face)
                if key in seen:
                    continue
                seen.add(key)
                if pos:                                    # outward normal: every other vertex behind the plane
                    nx, ny, nz = -nx, -ny, -nz
                # order the face's vertices counter-clockwise about the outward normal, starting from its lowest-numbered vertex
                cx = sum(v[m][0] for m in face) / len(face); cy = sum(v[m][1] for m in face) / len(face); cz = sum(v[m][2] for m in face) / len(face)
                f0 = face[0]
                ux, uy, uz = v[f0][0] - cx, v[f0][1] - cy, v[f0][2] - cz
                wx, wy, wz = ny * uz - nz * uy, nz * ux - nx * uz, nx * uy - ny * ux      # n x u
                ang = []
                for m in face:
                    px, py, pz = v[m][0] - cx, v[m][1] - cy, v[m][2] - cz
                    ang.append((math.atan2(px * wx + py * wy + pz * wz, px * ux + py * uy + pz * uz), m))
                ring = [f0] + [m for a_, m in sorted(x for x in ang if x[1] != f0)]
                # angles of the others are measured from f0's direction: wrap negatives so that the ring runs 0 .. 2 pi
                ring = [f0] + [m for a_, m in sorted(((a_ if a_ > 0 else a_ + 2 * math.pi), m) for a_, m in ang if m != f0)]
                L = len(ring)
                for t in range(L):
                    a_, b_ = ring[t], ring[(t + 1) % L]
                    adj[a_, b_] = adj[b_, a_] = True
                for t in range(2, L - 1):                  # fan diagonals from the lowest-numbered vertex
                    adj[f0, ring[t]] = adj[ring[t], f0] = True
    lists = [[int(m) for m in np.nonzero(adj[i])[0]] for i in range(n)]
    assert all(len(x) >= 3 for x in lists), "mesh has vertices inside its convex hull"
    return lists


# --------------------------------------------------------------------------- XML parsing
def _floats(s, n=None):
    v = [float(x) for x in s.replace(",", " ").split()]
    if n is not None:
        assert len(v) == n, (s, n)
    return v


def euler_deg_to_mat(e):
    """MuJoCo default eulerseq 'xyz' (intrinsic): R = Rx * Ry * Rz."""
    ax, ay, az = [math.radians(x) for x in e]
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def parse_xml(xml_path: str):
    root = ET.parse(xml_path).getroot()
    comp = root.find("compiler").attrib
    assert comp.get("coordinate") == "global" and comp.get("angle") == "degree"
    assert comp.get("inertiafromgeom") == "true"
    dflt = root.find("default")
    jd = dflt.find("joint").attrib
    gd = dflt.find("geom").attrib
    opt = root.find("option").attrib
    base = os.path.dirname(os.path.abspath(xml_path))
    meshes = {}
    for m in root.find("asset").findall("mesh"):
        f = m.attrib["file"]
        name = m.attrib.get("name", os.path.splitext(os.path.basename(f))[0])
        meshes[name] = os.path.join(base, f)

    wb = root.find("worldbody")
    floor = None
    for g in wb.findall("geom"):
        if g.attrib.get("type") == "plane":
            floor = dict(gd)
            floor.update(g.attrib)
    assert floor is not None

    bodies = []  # humanoid bodies in DFS order
    objects = []

    def walk(elem, parent):
        idx = len(bodies)
        gpos = np.array(_floats(elem.attrib["pos"], 3))
        joints = []
        for j in elem.findall("joint"):
            a = dict(jd)
            a.update(j.attrib)
            joints.append(a)
        geoms = []
        for g in elem.findall("geom"):
            a = dict(gd)
            a.update(g.attrib)
            geoms.append(a)
        bodies.append(dict(name=elem.attrib["name"], parent=parent, gpos=gpos, joints=joints, geoms=geoms))
        for ch in elem.findall("body"):
            walk(ch, idx)

    for top in wb.findall("body"):
        jn = top.findall("joint")
        is_humanoid = any(g.attrib.get("type") == "mesh" for g in top.findall("geom"))
        if is_humanoid:
            walk(top, -1)
        else:
            geoms = []
            for g in top.findall("geom"):
                a = dict(gd)
                a.update(g.attrib)
                geoms.append(a)
            objects.append(dict(name=top.attrib["name"], geoms=geoms, free=len(jn) == 1 and jn[0].attrib.get("type") == "free"))
    return dict(bodies=bodies, objects=objects, meshes=meshes, floor=floor, timestep=float(opt["timestep"]),
                joint_default=jd, geom_default=gd)


# --------------------------------------------------------------------------- dynamics at qpos0 (numpy, host)
def _mass_matrix_qpos0(parent, gpos, com_g, mass, inertia_w, dof_body, dof_axis, dof_is_trans, armature):
    """Dense M(qpos0) = sum_b Jv^T m Jv + Jw^T I Jw  (kinetic-energy form; host-only, runs once), accumulated in the order of
    kp_compile.hpp (per body, per row d1: s = sum_k m Jv[k, d1] Jv[k, :] + Jw[k, :] (I Jw[:, d1])[k])."""
    nb, nv = len(parent), len(dof_body)
    anc = np.zeros((nb, nb), bool)
    for b in range(nb):
        k = b
        while k >= 0:
            anc[b, k] = True
            k = parent[k]
    M = np.zeros((nv, nv))
    Js = []
    for b in range(nb):
        Jv = np.zeros((3, nv))
        Jw = np.zeros((3, nv))
        for d in range(nv):
            if not anc[b, dof_body[d]]:
                continue
            ax = dof_axis[d]
            if dof_is_trans[d]:
                Jv[:, d] = ax
            else:
                r = com_g[b] - gpos[dof_body[d]]
                Jw[:, d] = ax
                Jv[:, d] = [ax[1] * r[2] - ax[2] * r[1], ax[2] * r[0] - ax[0] * r[2], ax[0] * r[1] - ax[1] * r[0]]
        I = inertia_w[b]
        for d1 in range(nv):
            iw = [(I[k, 0] * Jw[0, d1] + I[k, 1] * Jw[1, d1]) + I[k, 2] * Jw[2, d1] for k in range(3)]
            srow = np.zeros(nv)
            for k in range(3):
                srow = srow + ((mass[b] * Jv[k, d1]) * Jv[k, :] + Jw[k, :] * iw[k])
            M[d1, :] += srow
        Js.append((Jv, Jw))
    for d in range(nv):
        M[d, d] += armature[d]
    return M, Js


def compile_model(xml_path: str, uhc_yml: str | None = None) -> dict:
    import yaml

    px = parse_xml(xml_path)
    bodies = px["bodies"]
    nb = len(bodies)
    density = MJ_DEFAULTS["density"]

    parent = np.array([b["parent"] for b in bodies], np.int32)
    gpos = np.stack([b["gpos"] for b in bodies])
    body_pos = gpos.copy()
    for i in range(nb):
        if parent[i] >= 0:
            body_pos[i] = gpos[i] - gpos[parent[i]]  # all rest quats are identity (XML quat="1 0 0 0")

    mass = np.zeros(nb)
    ipos = np.zeros((nb, 3))
    inertia = np.zeros((nb, 3, 3))
    verts_all, vert_adr, rbound, mesh_rbound = [], [0], np.zeros(nb), np.zeros(nb)
    nbr_adr, nbr = [0], []
    for i, b in enumerate(bodies):
        assert len(b["geoms"]) == 1 and b["geoms"][0]["type"] == "mesh"
        tris = read_binary_stl(px["meshes"][b["geoms"][0]["mesh"]])
        m, com, I = polyhedron_mass_props(tris, density)
        mass[i], ipos[i], inertia[i] = m, com - gpos[i], I
        v = np.unique(tris.reshape(-1, 3), axis=0) - gpos[i]  # hull vertices in the body frame
        verts_all.append(v)
        vert_adr.append(vert_adr[-1] + len(v))
        rbound[i] = float(np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]).max())      # bounding sphere about the BODY origin: the kernels' broad phase
        # mjModel.geom_rbound of the mesh geom [MJ-ext]: MuJoCo's mesh compiler centres the mesh at its COM, rotates it to its
        # principal axes of inertia and keeps the half-sizes max |coordinate| of that frame as the geom's `size`; rbound of a mesh
        # (as of a box) is the norm of `size`.  mjc_PlaneConvex's "too close to the first contact" test is tolplanemesh * rbound.
        axes = eigh3(I)
        d = (v + gpos[i]) - com
        half = [float(np.abs((d[:, 0] * axes[0, c] + d[:, 1] * axes[1, c]) + d[:, 2] * axes[2, c]).max()) for c in range(3)]
        mesh_rbound[i] = math.sqrt(half[0] * half[0] + half[1] * half[1] + half[2] * half[2])
        lists = hull_graph(v)
        for lst in lists:
            nbr.extend(lst)
            nbr_adr.append(len(nbr))
    verts = np.concatenate(verts_all)

    # ---- dofs: free root (3 trans world axes + 3 rot body axes) then 3 hinges (z,y,x) per body
    dof_body, dof_axis, dof_trans, arm, jrange, jlimited = [], [], [], [], [], []
    for i, b in enumerate(bodies):
        for j in b["joints"]:
            if j["type"] == "free":
                for k in range(3):
                    dof_body.append(i); dof_axis.append(np.eye(3)[k]); dof_trans.append(1); arm.append(float(j.get("armature", 0)))
                for k in range(3):
                    dof_body.append(i); dof_axis.append(np.eye(3)[k]); dof_trans.append(0); arm.append(float(j.get("armature", 0)))
            else:
                assert j["type"] == "hinge"
                assert np.allclose(_floats(j["pos"], 3), b["gpos"])  # hinge anchored at the body origin
                dof_body.append(i); dof_axis.append(np.array(_floats(j["axis"], 3))); dof_trans.append(0)
                arm.append(float(j["armature"]))
                r = _floats(j["range"], 2)
                jrange.append([math.radians(r[0]), math.radians(r[1])])
                jlimited.append(1 if j.get("limited", "false") == "true" else 0)
    nv = len(dof_body)
    dof_body = np.array(dof_body, np.int32)
    dof_axis = np.stack(dof_axis)
    # hinge order inside each body must be z, y, x
    for i in range(1, nb):
        ax = dof_axis[6 + 3 * (i - 1): 9 + 3 * (i - 1)]
        assert np.allclose(ax, np.eye(3)[[2, 1, 0]]), f"unexpected hinge order in body {i}"
    # dof tree in MuJoCo layout: dof_parent = previous dof in the chain
    last_dof_of_body = {}
    dof_parent = np.full(nv, -1, np.int32)
    for d in range(nv):
        b = dof_body[d]
        if d > 0 and dof_body[d - 1] == b:
            dof_parent[d] = d - 1
        elif parent[b] >= 0:
            dof_parent[d] = last_dof_of_body[parent[b]]
        last_dof_of_body[b] = d
    dof_depth = np.zeros(nv, np.int32)
    for d in range(nv):
        dof_depth[d] = 0 if dof_parent[d] < 0 else dof_depth[dof_parent[d]] + 1
    dof_madr = np.zeros(nv + 1, np.int32)
    for d in range(nv):
        dof_madr[d + 1] = dof_madr[d] + dof_depth[d] + 1
    nM = int(dof_madr[nv])

    # subtree sizes (bodies are in DFS order => subtree(b) = [b, b+size))
    subtree = np.ones(nb, np.int32)
    for i in range(nb - 1, 0, -1):
        subtree[parent[i]] += subtree[i]
    body_depth = np.zeros(nb, np.int32)
    for i in range(1, nb):
        body_depth[i] = body_depth[parent[i]] + 1

    # ---- qpos0 constants for the constraint model [MJ-ext: engine_setconst.c set0]
    com_g = gpos + ipos
    M0, Js = _mass_matrix_qpos0(parent, gpos, com_g, mass, inertia, dof_body, dof_axis, dof_trans, np.array(arm))
    Minv = gauss_jordan_inverse(M0)
    body_invw = np.zeros((nb, 2))
    for b in range(nb):
        for part, J in enumerate(Js[b]):                      # trace of J Minv J^T over the three translational / rotational rows
            tr = 0.0
            for k in range(3):
                t = seq_sum(Minv * J[k][None, :], axis=1)     # t[d1] = sum_d2 Minv[d1, d2] J[k, d2], left to right
                tr += float(seq_sum(J[k] * t))
            body_invw[b, part] = tr / 3.0
    dinv = np.diag(Minv).copy()
    dof_invw = dinv.copy()
    dof_invw[0:3] = ((dinv[0] + dinv[1]) + dinv[2]) / 3.0
    dof_invw[3:6] = ((dinv[3] + dinv[4]) + dinv[5]) / 3.0
    # (meaninertia: see the free-object section below -- it spans all dofs of the scene)

    # ---- floor
    fl = px["floor"]
    floor_friction = _floats(fl["friction"]) if "friction" in fl else list(MJ_DEFAULTS["geom_friction"])
    geom_margin = float(px["geom_default"].get("margin", 0.0))
    # contact friction = elementwise max of the pair, margin = max of the pair [MJ-ext]
    fric = np.maximum(np.array(floor_friction), np.array(MJ_DEFAULTS["geom_friction"]))
    condim = max(int(fl.get("condim", 3)), int(px["geom_default"].get("condim", 3)))

    # ---- controller gains (uhc.yml joint_params; reference copycat_config.py:133-146)
    nu = nv - 6
    kp = np.zeros(nu); kd = np.zeros(nu); tlim = np.zeros(nu); a_scale = np.ones(nu)
    rfc_scale, rfc_lim = 100.0, 100.0
    base_rot = [0.7071, 0.7071, 0.0, 0.0]
    if uhc_yml is not None:
        cfg = yaml.safe_load(open(uhc_yml))
        jp = cfg["joint_params"]
        names = [f"{b['name']}_{a}" for b in bodies[1:] for a in "zyx"]
        assert [r[0] for r in jp] == names, "uhc.yml joint order differs from the XML dof order"
        kp = np.array([r[1] for r in jp], float)
        kd = np.array([r[2] for r in jp], float)
        a_scale = np.array([r[4] for r in jp], float)
        tlim = np.array([r[5] for r in jp], float)
        rfc_scale = float(cfg.get("residual_force_scale", 200.0))
        rfc_lim = float(cfg.get("residual_force_lim", 100.0))
        base_rot = cfg.get("data_specs", {}).get("base_rot", base_rot)
    # env.jpos_diffw (calc_body_diff) defaults to ones in both envs: no config carries reward_weights['jpos_diffw']
    # (uhc/envs/humanoid_im.py:28, kin_poly/envs/humanoid_ar_v1.py:59).  uhc.yml body_params (toes / hands 0) only weigh the pose
    # term of the UHC reward, cfg.b_diffw (copycat_config.py:139-143, uhc/core/reward_function.py:31).
    diffw = np.ones(nb)
    uhc_b_diffw = np.ones(nb)
    if uhc_yml is not None and "body_params" in cfg:
        bp = cfg["body_params"]
        assert [r[0] for r in bp] == [b["name"] for b in bodies[1:]], "uhc.yml body order differs from the XML body order"
        uhc_b_diffw = np.concatenate([[1.0], np.array([r[1] for r in bp], float)])

    # ---- free objects: collision geoms (body frame) + inertial properties from the geoms' explicit `mass=`
    # (inertiafromgeom) [MJ-ext].  obj_inertial[o] = mass, com[3], inertia about com in body axes (xx yy zz xy xz yz),
    # invweight0 (translational, rotational), free-joint armature (the <default><joint armature> applies to them).
    obj_geoms = []
    for oi, ob in enumerate(px["objects"]):
        for g in ob["geoms"]:
            typ = {"box": 0, "cylinder": 1}[g["type"]]
            size = _floats(g["size"]) + [0.0]
            R = euler_deg_to_mat(_floats(g.get("euler", "0 0 0"), 3))
            obj_geoms.append([oi, typ, *size[:3], *_floats(g.get("pos", "0 0 0"), 3), *R.reshape(-1), float(g["mass"])])
    obj_geoms = np.array(obj_geoms, float).reshape(-1, 18)
    nobj = len(px["objects"])
    obj_geom_adr = np.zeros(nobj + 1, np.int32)
    obj_mass = np.zeros(nobj)
    for g in obj_geoms:
        obj_geom_adr[int(g[0]) + 1:] += 1
        obj_mass[int(g[0])] += g[17]
    obj_arm = float(px["joint_default"].get("armature", 0.0))
    obj_inertial = np.zeros((nobj, 13))
    obj_trace = 0.0
    for oi in range(nobj):                                   # plain loops in kp_compile.hpp's order (same bytes from both compilers)
        gs = [[float(x) for x in g] for g in obj_geoms if int(g[0]) == oi]
        mo = 0.0
        for g in gs:
            mo += g[17]
        com = [0.0, 0.0, 0.0]
        for g in gs:
            for k in range(3):
                com[k] += g[17] * g[5 + k]
        com = [com[0] / mo, com[1] / mo, com[2] / mo]
        Io = [0.0] * 9
        for g in gs:
            mg, sz, Rg = g[17], g[2:5], g[8:17]
            if int(g[1]) == 0:      # box, half sizes
                Il = [mg / 3.0 * (sz[1] * sz[1] + sz[2] * sz[2]), mg / 3.0 * (sz[0] * sz[0] + sz[2] * sz[2]), mg / 3.0 * (sz[0] * sz[0] + sz[1] * sz[1])]
            else:                   # cylinder along local z: radius, half height
                ixx = mg * (3.0 * sz[0] * sz[0] + (2.0 * sz[1]) * (2.0 * sz[1])) / 12.0
                Il = [ixx, ixx, 0.5 * mg * sz[0] * sz[0]]
            dd = [g[5] - com[0], g[6] - com[1], g[7] - com[2]]
            d2 = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]
            for i in range(3):
                for j in range(3):
                    s_ = 0.0
                    for k in range(3):
                        s_ += Rg[3 * i + k] * Il[k] * Rg[3 * j + k]
                    Io[3 * i + j] += s_ + mg * ((d2 if i == j else 0.0) - dd[i] * dd[j])
        # generalized mass matrix of the free joint at the identity pose: dofs = [lin (world); ang (body axes, about the body origin)]
        rx = [0.0, -com[2], com[1], com[2], 0.0, -com[0], -com[1], com[0], 0.0]
        Jv = [[0.0] * 6 for _ in range(3)]; Jw = [[0.0] * 6 for _ in range(3)]
        for i in range(3):
            for j in range(3):
                Jv[i][j] = 1.0 if i == j else 0.0; Jv[i][3 + j] = -rx[3 * i + j]; Jw[i][3 + j] = 1.0 if i == j else 0.0
        Mo = np.zeros((6, 6))
        for a_ in range(6):
            for b_ in range(6):
                s_ = 0.0
                for k in range(3):
                    s_ += mo * Jv[k][a_] * Jv[k][b_]
                for k in range(3):
                    for l_ in range(3):
                        s_ += Jw[k][a_] * Io[3 * k + l_] * Jw[l_][b_]
                Mo[a_, b_] = s_ + (obj_arm if a_ == b_ else 0.0)
        Moi = gauss_jordan_inverse(Mo)
        trv = trw = 0.0
        for k in range(3):
            for a_ in range(6):
                for b_ in range(6):
                    trv += Jv[k][a_] * Moi[a_, b_] * Jv[k][b_]; trw += Jw[k][a_] * Moi[a_, b_] * Jw[k][b_]
        obj_inertial[oi] = [mo, *com, Io[0], Io[4], Io[8], Io[1], Io[2], Io[5], trv / 3.0, trw / 3.0, obj_arm]
        for a_ in range(6):
            obj_trace += float(Mo[a_, a_])
    # mjModel.stat.meaninertia is the mean diagonal of qM at qpos0 over ALL dofs of the scene, objects included, and the
    # solver's termination scale is 1 / (meaninertia * nv) with the scene's nv [MJ-ext]: both are kept as the reference has them.
    nv_full = nv + 6 * nobj
    meaninertia = float((float(seq_sum(np.diag(M0))) + obj_trace) / nv_full)

    model = dict(
        dims=np.array([nb, nv, nv + 1, nu, nM, len(verts), len(px["objects"]), len(obj_geoms), condim], np.int32),
        body_parent=parent, body_depth=body_depth, body_subtree=subtree,
        body_pos=body_pos, body_ipos=ipos, body_mass=mass,
        body_inertia=np.stack([inertia[:, 0, 0], inertia[:, 1, 1], inertia[:, 2, 2],
                               inertia[:, 0, 1], inertia[:, 0, 2], inertia[:, 1, 2]], axis=1),
        body_gpos0=gpos, body_rbound=rbound, mesh_rbound=mesh_rbound, body_diffw=diffw, uhc_b_diffw=uhc_b_diffw,
        planemesh=np.array([MJ_DEFAULTS["maxplanemesh"], MJ_DEFAULTS["tolplanemesh"]], float),
        body_invweight0=body_invw, dof_invweight0=dof_invw,
        dof_body=dof_body, dof_parent=dof_parent, dof_depth=dof_depth, dof_madr=dof_madr,
        dof_armature=np.array(arm), jnt_range=np.array(jrange), jnt_limited=np.array(jlimited, np.int32),
        vert_adr=np.array(vert_adr, np.int32), verts=verts,
        vert_nbr_adr=np.array(nbr_adr, np.int32), vert_nbr=np.array(nbr, np.int32),   # hull graph: neighbours of vertex v (hull-local ids)
        kp=kp, kd=kd, torque_lim=tlim, a_scale=a_scale,
        opt=np.array([px["timestep"], *MJ_DEFAULTS["gravity"], *MJ_DEFAULTS["solref"], *MJ_DEFAULTS["solimp"],
                      *fric, geom_margin, MJ_DEFAULTS["impratio"], meaninertia,
                      rfc_scale, rfc_lim, *base_rot,
                      MJ_DEFAULTS["solver_iterations"], MJ_DEFAULTS["solver_tolerance"], nv_full], float),
        obj_geoms=obj_geoms, obj_geom_adr=obj_geom_adr, obj_mass=obj_mass, obj_inertial=obj_inertial,
        M0=M0,
    )
    model["_names"] = [b["name"] for b in bodies]
    return model


# opt[] index map (shared with the C side: kp_model.hpp / kp_oracle.c)
OPT_FIELDS = ["timestep", "gx", "gy", "gz", "solref_tc", "solref_dr", "solimp_d0", "solimp_dw", "solimp_w",
              "solimp_mid", "solimp_pow", "fric_slide", "fric_spin", "fric_roll", "margin", "impratio",
              "meaninertia", "rfc_scale", "rfc_lim", "base_rot_w", "base_rot_x", "base_rot_y", "base_rot_z",
              "solver_iter", "solver_tol", "nv_full"]


def write_kpm(model: dict, path: str):
    entries = [(k, np.ascontiguousarray(v)) for k, v in model.items() if not k.startswith("_")]
    hdr = 12 + 56 * len(entries)
    off = (hdr + 7) // 8 * 8
    table, blobs = [], []
    for name, arr in entries:
        if arr.dtype.kind == "f":
            arr = arr.astype("<f8"); dt = 0
        else:
            arr = arr.astype("<i4"); dt = 1
        raw = arr.tobytes()
        table.append(struct.pack("<32sIIQQ", name.encode(), dt, 0, arr.size, off))
        blobs.append((off, raw))
        off = (off + len(raw) + 7) // 8 * 8
    with open(path, "wb") as f:
        f.write(struct.pack("<III", KPM_MAGIC, KPM_VERSION, len(entries)))
        for t in table:
            f.write(t)
        for o, raw in blobs:
            f.seek(o)
            f.write(raw)
        f.truncate(off)


def read_kpm(path: str) -> dict:
    buf = open(path, "rb").read()
    magic, ver, n = struct.unpack_from("<III", buf, 0)
    assert magic == KPM_MAGIC, "not a KPM blob"
    out = {"_version": ver}
    for i in range(n):
        name, dt, _, cnt, off = struct.unpack_from("<32sIIQQ", buf, 12 + 56 * i)
        name = name.split(b"\0")[0].decode()
        out[name] = np.frombuffer(buf, dtype="<f8" if dt == 0 else "<i4", count=cnt, offset=off).copy()
    return out


DEFAULT_KPM = os.path.join(os.path.dirname(__file__), "assets", "smpl_humanoid.kpm")            # humanoid_smpl_neutral_mesh_all.xml
STEP_KPM = os.path.join(os.path.dirname(__file__), "assets", "smpl_humanoid_step.kpm")          # ..._all_step.xml (mocap training, agent_ar.py:168)


def main(argv):
    xml, yml, out = argv[1], argv[2], argv[3]
    m = compile_model(xml, yml if yml != "-" else None)
    write_kpm(m, out)
    print(f"wrote {out}: nbody={m['dims'][0]} nv={m['dims'][1]} nM={m['dims'][4]} nvert={m['dims'][5]} "
          f"mass={m['body_mass'].sum():.3f} kg")


if __name__ == "__main__":
    main(sys.argv)
