"""CPU tests of the oracle's narrow phases (oracle/kp_collide.h: restatements of mjc_PlaneConvex, mjc_PlaneBox, mjc_PlaneCylinder,
mjc_Convex / libccd MPR, box-box) on configurations whose contacts are known in closed form.  MuJoCo itself cannot run here, so
these pin the restatement to geometry, not to MuJoCo's binary (DESIGN.md section 2 lists what stays unverifiable)."""
import numpy as np
import pytest

from kinpoly_amd.model_compiler import DEFAULT_KPM, hull_graph, read_kpm
from oracle.kpo import narrowphase, shape_record

KPM = read_kpm(DEFAULT_KPM)


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


CUBE = np.array([[x, y, z] for z in (-1, 1) for y in (-1, 1) for x in (-1, 1)], float)


def test_plane_box_returns_the_bottom_corners():
    # flat box, 1 mm into the floor: four bottom corners in index order, dist = -0.001, positions midway
    c = narrowphase("plane_box", a=shape_record("box", (0.3, 0.2, 0.1), (1.0, 2.0, 0.099)))
    assert c.shape[0] == 4
    np.testing.assert_allclose(c[:, 0], -0.001, atol=1e-12)
    np.testing.assert_allclose(c[:, 1:3], [[0.7, 1.8], [1.3, 1.8], [0.7, 2.2], [1.3, 2.2]], atol=1e-12)
    np.testing.assert_allclose(c[:, 3], -0.0005, atol=1e-12)
    np.testing.assert_allclose(c[:, 4:], np.tile([0, 0, 1.0], (4, 1)))
    # inside the margin but not touching: still contacts (dist > 0); beyond the margin: none
    assert narrowphase("plane_box", a=shape_record("box", (0.3, 0.2, 0.1), (0, 0, 0.1005))).shape[0] == 4
    assert narrowphase("plane_box", a=shape_record("box", (0.3, 0.2, 0.1), (0, 0, 0.1015))).shape[0] == 0
    # tilted: only the lowest edge's corners are within the margin
    R = rot([1, 0, 0], 0.3)
    z = -(R @ np.array([0.3, -0.2, -0.1]))[2]
    c = narrowphase("plane_box", a=shape_record("box", (0.3, 0.2, 0.1), (0, 0, z), R))
    assert c.shape[0] == 2 and np.allclose(c[:, 0], 0.0, atol=1e-12)
    # a box deep in the floor: corners that point up (ldist > 0) are skipped even if they are below the plane; at most 4
    c = narrowphase("plane_box", a=shape_record("box", (0.1, 0.1, 0.1), (0, 0, -0.5)))
    assert c.shape[0] == 4 and np.allclose(c[:, 0], -0.6)


def test_plane_cylinder_upright_and_lying():
    # upright can standing on the floor: lower-cap point along the cylinder's x axis + two more at +-120 degrees (the "triangle")
    r, h = 0.279, 0.345
    c = narrowphase("plane_cylinder", a=shape_record("cylinder", (r, h), (0, 0, h - 0.0004)))
    assert c.shape[0] == 3
    np.testing.assert_allclose(c[:, 0], -0.0004, atol=1e-12)
    rim = c[:, 1:3]
    np.testing.assert_allclose(np.linalg.norm(rim, axis=1), r, atol=1e-12)
    ang = np.sort(np.mod(np.arctan2(rim[:, 1], rim[:, 0]), 2 * np.pi))
    np.testing.assert_allclose(np.diff(ang), 2 * np.pi / 3, atol=1e-9)
    # lying on its side: the two end points of the lowest generator line
    R = rot([0, 1, 0], np.pi / 2)                       # cylinder axis along world x
    c = narrowphase("plane_cylinder", a=shape_record("cylinder", (0.03, 0.3), (0, 0, 0.0295), R))
    assert c.shape[0] == 2
    np.testing.assert_allclose(c[:, 0], -0.0005, atol=1e-12)
    np.testing.assert_allclose(sorted(c[:, 1]), [-0.3, 0.3], atol=1e-12)
    np.testing.assert_allclose(c[:, 2], 0.0, atol=1e-12)
    # out of reach
    assert narrowphase("plane_cylinder", a=shape_record("cylinder", (0.03, 0.3), (0, 0, 0.5), R)).shape[0] == 0


def test_plane_mesh_support_vertex_then_graph_neighbours():
    """mjc_PlaneConvex [MJ-ext]: support vertex, then its hull-graph neighbours in graph order while fewer than maxplanemesh = 3 contacts
    exist; a neighbour within tolplanemesh * rbound = 0.3 rbound of the FIRST contact is skipped (addplanemesh)."""
    v = 0.1 * CUBE
    graph = hull_graph(v)
    assert all(3 <= len(g) <= 6 for g in graph)
    rb = 0.1 * np.sqrt(3.0)                              # geom_rbound of the cube: norm of its half-sizes
    # cube hull resting flat, 0.5 mm deep: the support vertex (first of the four lowest) + its lower neighbours in graph order, 3 in all
    b = shape_record("hull", pos=(0.5, 0.5, 0.0995), center=(0.5, 0.5, 0.0995))
    c = narrowphase("plane_mesh", b=b, verts_b=v, graph=graph, tol_rbound=0.3 * rb, maxcon=3)
    low = [i for i in range(8) if v[i, 2] < 0]
    first = low[0]
    expect = ([first] + [j for j in graph[first] if j in low])[:3]       # neighbours are 0.2 or 0.283 apart > 0.3 * 0.173
    assert c.shape[0] == len(expect) == 3
    np.testing.assert_allclose(c[:, 1:3], v[expect, :2] + 0.5, atol=1e-12)
    np.testing.assert_allclose(c[:, 0], -0.0005, atol=1e-12)
    np.testing.assert_allclose(c[:, 3], -0.00025, atol=1e-12)
    # the cap is the option: 4 keeps the third lower neighbour as well
    c4 = narrowphase("plane_mesh", b=b, verts_b=v, graph=graph, tol_rbound=0.3 * rb, maxcon=4)
    assert c4.shape[0] == len([first] + [j for j in graph[first] if j in low]) and (c4[:3] == c).all()
    # the tolerance: with 0.3 * rbound above the edge length (0.2) only the face-diagonal neighbour (0.283 away) survives
    ct = narrowphase("plane_mesh", b=b, verts_b=v, graph=graph, tol_rbound=0.25, maxcon=3)
    far = [j for j in graph[first] if j in low and np.linalg.norm(v[j] - v[first]) > 0.25]
    assert ct.shape[0] == 1 + len(far)
    np.testing.assert_allclose(ct[1:, 1:3], v[far, :2] + 0.5, atol=1e-12)
    # standing on one corner: exactly one contact (the neighbours are far above the margin)
    R = rot([1, -1, 0], np.arccos(1 / np.sqrt(3)))      # the body diagonal (1,1,1) to +z: corner (-1,-1,-1) becomes the lowest point
    zmin = (v @ R.T)[:, 2].min()
    c = narrowphase("plane_mesh", b=shape_record("hull", pos=(0, 0, -zmin - 0.0002), mat=R, center=(0, 0, 0)), verts_b=v, graph=graph, tol_rbound=0.3 * rb)
    assert c.shape[0] == 1 and abs(c[0, 0] + 0.0002) < 1e-12
    # a real foot hull of the model flat on the floor: at most 3 contacts, graph neighbours of the deepest vertex that are not within
    # 0.3 * mesh_rbound of it, in graph order
    adr = KPM["vert_adr"]; vb = KPM["verts"].reshape(-1, 3)[adr[4]:adr[5]]
    g4 = [list(KPM["vert_nbr"][KPM["vert_nbr_adr"][adr[4] + i]:KPM["vert_nbr_adr"][adr[4] + i + 1]]) for i in range(len(vb))]
    z0 = -vb[:, 2].min() - 0.002
    tol = float(KPM["planemesh"][1] * KPM["mesh_rbound"][4])
    assert int(KPM["planemesh"][0]) == 3 and abs(KPM["planemesh"][1] - 0.3) < 1e-12
    c = narrowphase("plane_mesh", b=shape_record("hull", pos=(0, 0, z0), center=(0, 0, z0)), verts_b=vb, graph=g4, tol_rbound=tol, maxcon=3)
    deepest = int(np.argmin(vb[:, 2]))
    assert 1 <= c.shape[0] <= 3 and abs(c[0, 0] + 0.002) < 1e-9
    ids = [int(np.argmin(np.linalg.norm(vb[:, :2] - p[1:3], axis=1))) for p in c]
    first_pos = np.array([vb[deepest, 0], vb[deepest, 1], 0.5 * (vb[deepest, 2] + z0)])
    want = [j for j in g4[deepest] if vb[j, 2] + z0 <= 0.001 and np.linalg.norm(vb[j] + [0, 0, z0] - first_pos) >= tol][:2]
    assert ids[0] == deepest and ids[1:] == want


def _cube_hull(pos, half=0.1, R=None):
    return shape_record("hull", pos=pos, mat=R, center=pos), half * CUBE


def test_mpr_contact_matches_closed_form_penetration():
    # cube hull (half 0.1) pressed 5 mm into the top face of a big box: one contact, normal box -> hull = +z, dist = -0.005
    box = shape_record("box", (0.4, 0.4, 0.1), (0, 0, 0))
    hull, v = _cube_hull((0.05, -0.02, 0.195))
    c = narrowphase("convex", a=box, b=hull, verts_b=v)
    assert c.shape[0] == 1
    assert abs(c[0, 0] + 0.005) < 1e-6
    np.testing.assert_allclose(c[0, 4:], [0, 0, 1], atol=1e-6)
    assert abs(c[0, 3] - 0.0975) < 1e-6                   # midway between the two surfaces (z = 0.1 and 0.095)
    assert abs(c[0, 1] - 0.05) < 0.11 and abs(c[0, 2] + 0.02) < 0.11     # somewhere under the cube's face
    # a gap smaller than the margin is still a contact with positive distance; a larger one is none
    hull, v = _cube_hull((0, 0, 0.2004))
    c = narrowphase("convex", a=box, b=hull, verts_b=v)
    assert c.shape[0] == 1 and abs(c[0, 0] - 0.0004) < 1e-6 and c[0, 6] > 0.999999
    hull, v = _cube_hull((0, 0, 0.2012))
    assert narrowphase("convex", a=box, b=hull, verts_b=v).shape[0] == 0
    # sideways: the hull touches the +x face
    hull, v = _cube_hull((0.497, 0, 0.0))
    c = narrowphase("convex", a=box, b=hull, verts_b=v)
    assert c.shape[0] == 1 and abs(c[0, 0] + 0.003) < 1e-6
    np.testing.assert_allclose(c[0, 4:], [1, 0, 0], atol=1e-6)
    # rotated hull, corner down into the box top: dist = lowest vertex z - 0.1, normal +z
    R = rot([1, -1, 0], np.arccos(1 / np.sqrt(3)))
    zmin = (v @ R.T)[:, 2].min()
    hull, v = _cube_hull((0, 0, 0.1 - zmin - 0.002), R=R)
    c = narrowphase("convex", a=box, b=hull, verts_b=v)
    assert c.shape[0] == 1 and abs(c[0, 0] + 0.002) < 1e-5 and c[0, 6] > 0.9999
    np.testing.assert_allclose(c[0, 1:3], 0.0, atol=2e-3)


def test_mpr_cylinder_pairs():
    # table leg (cylinder r = 0.03, half height 0.3) against the side of a box: axis distance 0.028 + 0.15 -> 2 mm deep
    cyl = shape_record("cylinder", (0.03, 0.3), (0.178, 0, 0))
    box = shape_record("box", (0.15, 0.19, 0.12), (0, 0, 0))
    c = narrowphase("convex", a=cyl, b=box)                 # geom 1 = cylinder (lower type): normal cylinder -> box = -x
    assert c.shape[0] == 1 and abs(c[0, 0] + 0.002) < 1e-5
    np.testing.assert_allclose(c[0, 4:], [-1, 0, 0], atol=1e-4)
    # cube hull standing on the Can's top cap
    can = shape_record("cylinder", (0.279, 0.345), (0, 0, 0.345))
    hull, v = _cube_hull((0.05, 0.05, 0.69 + 0.1 - 0.001))
    c = narrowphase("convex", a=can, b=hull, verts_b=v)
    assert c.shape[0] == 1 and abs(c[0, 0] + 0.001) < 1e-5 and c[0, 6] > 0.99999


def test_box_box_face_contact_is_the_clipped_footprint():
    # the pushed box (0.15 x 0.19 x 0.12) resting 0.4 mm deep on the table top (0.499 x 0.294 x 0.01): its four bottom corners
    top = shape_record("box", (0.499, 0.294, 0.01), (0, 0, 0.70))
    Rz = rot([0, 0, 1], 0.4)
    bx = shape_record("box", (0.15, 0.19, 0.12), (0.1, -0.05, 0.71 + 0.12 - 0.0004), Rz)
    c = narrowphase("box_box", a=bx, b=top)                  # geom 1 = the box: normal box -> table = -z
    assert c.shape[0] == 4
    np.testing.assert_allclose(c[:, 0], -0.0004, atol=1e-9)
    np.testing.assert_allclose(c[:, 4:], np.tile([0, 0, -1.0], (4, 1)), atol=1e-9)
    corners = (np.array([[sx * 0.15, sy * 0.19, 0] for sx in (-1, 1) for sy in (-1, 1)]) @ Rz.T)[:, :2] + [0.1, -0.05]
    got = c[:, 1:3]
    assert all(np.min(np.linalg.norm(corners - g, axis=1)) < 1e-9 for g in got) and len({tuple(np.round(g, 6)) for g in got}) == 4
    np.testing.assert_allclose(c[:, 3], 0.71 - 0.0002, atol=1e-9)
    # overhanging the table edge: the footprint is clipped (points on the table's boundary appear)
    bx = shape_record("box", (0.15, 0.19, 0.12), (0.45, 0.0, 0.71 + 0.12 - 0.0004))
    c = narrowphase("box_box", a=bx, b=top)
    assert c.shape[0] == 4 and np.isclose(c[:, 1].max(), 0.499) and np.isclose(c[:, 1].min(), 0.30)
    # far apart
    assert narrowphase("box_box", a=shape_record("box", (0.1, 0.1, 0.1), (0, 0, 2.0)), b=top).shape[0] == 0
    # edge - edge: two boxes crossed at 45 degrees touching along their edges -> a single contact
    a = shape_record("box", (0.5, 0.05, 0.05), (0, 0, 0), rot([1, 0, 0], np.pi / 4))
    b = shape_record("box", (0.05, 0.5, 0.05), (0, 0, 2 * 0.05 * np.sqrt(2) - 0.002), rot([0, 1, 0], np.pi / 4))
    c = narrowphase("box_box", a=a, b=b)
    assert c.shape[0] == 1 and abs(c[0, 0] + 0.002) < 1e-9 and c[0, 6] > 0.999999
    np.testing.assert_allclose(c[0, 1:3], 0.0, atol=1e-9)


def test_hull_graph_of_the_model_is_symmetric_and_complete():
    adr, nadr, nbr = KPM["vert_adr"], KPM["vert_nbr_adr"], KPM["vert_nbr"]
    assert nadr.shape[0] == adr[-1] + 1 and nadr[-1] == nbr.shape[0]
    for b in range(24):
        n = adr[b + 1] - adr[b]
        lists = [list(nbr[nadr[adr[b] + i]:nadr[adr[b] + i + 1]]) for i in range(n)]
        assert all(0 <= j < n for g in lists for j in g) and all(len(g) >= 3 and len(set(g)) == len(g) for g in lists)
        assert all(i in lists[j] for i, g in enumerate(lists) for j in g)               # undirected
        assert sum(len(g) for g in lists) // 2 == 3 * n - 6                               # Euler: a triangulated convex polyhedron
